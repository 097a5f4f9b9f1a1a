cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x -k "wide or ks_sweep or c4 or c5 or golden or capacity or random or many_tiles" > gpurun_out/r02_wt_test.log 2>&1 || { tail -60 gpurun_out/r02_wt_test.log; exit 1; }
tail -2 gpurun_out/r02_wt_test.log
timeout -k 10 300 python tools/probe_small_S.py 4 > gpurun_out/r02_probe_small_S_c4.log 2>&1 && timeout -k 10 300 python tools/probe_small_S.py 5 > gpurun_out/r02_probe_small_S_c5.log 2>&1
cat gpurun_out/r02_probe_small_S_c4.log gpurun_out/r02_probe_small_S_c5.log
for S in 16384 32768 65536; do echo -n "C4 S=$S: "; python tools/probe.py $S 8 1 4 tiles | tail -1 | cut -c1-95; done
for S in 2048 4096 8192; do echo -n "C5 S=$S: "; python tools/probe.py $S 8 1 5 tiles | tail -1 | cut -c1-95; done
