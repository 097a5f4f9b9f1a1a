set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r02_gpu_tests.log 2>&1 || { tail -60 gpurun_out/r02_gpu_tests.log; exit 1; }
tail -3 gpurun_out/r02_gpu_tests.log
