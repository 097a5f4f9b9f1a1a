set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x -k "exp or EXP or golden or tutorial or c1" 2>&1 | tail -3
timeout -k 10 300 python tools/probe_exp.py 2 2>&1 | tee gpurun_out/r02_probe_exp_C2.log
timeout -k 10 300 python tools/probe_exp.py 1 2>&1 | tee gpurun_out/r02_probe_exp_C1.log
