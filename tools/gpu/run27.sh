set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x -k "wide or ks_sweep or c4 or c5 or golden or capacity or random or many_tiles" > gpurun_out/r02_wt_test.log 2>&1 || { tail -60 gpurun_out/r02_wt_test.log; exit 1; }
tail -2 gpurun_out/r02_wt_test.log
for S in 64 256 1024 2048 4096 6144; do python tools/probe.py $S 10 1 4 tiles | tail -1 | cut -c1-95; done
for S in 64 256 1024 2048 4096; do python tools/probe.py $S 10 1 5 tiles | tail -1 | cut -c1-95; done
