cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x -k "wide or ks_sweep or c4 or c5 or golden or capacity or many_tiles or pruning_kernels or random" > gpurun_out/r02_wt_test.log 2>&1 || { tail -60 gpurun_out/r02_wt_test.log; exit 1; }
tail -2 gpurun_out/r02_wt_test.log
python tools/probe_phases.py 4 65536
python tools/probe_phases.py 5 16384
python tools/probe_phases.py 4 4096
python tools/probe_phases.py 5 4096
python tools/probe_phases.py 4 64
timeout -k 10 300 python tools/fuzz_mappings.py 41 300 2>&1 | tail -2
