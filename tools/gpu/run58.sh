cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x -k "wide or ks_sweep or c4 or c5 or golden or capacity or random or many_tiles or pruning_kernels" > gpurun_out/r02_wt_test.log 2>&1 || { tail -60 gpurun_out/r02_wt_test.log; exit 1; }
tail -2 gpurun_out/r02_wt_test.log
echo "3 waves (168 regs), n-row LDS"; python tools/probe_phases.py 5 16384; python tools/probe_phases.py 4 65536
echo "2 waves (216 regs), n-row LDS"; PHM_LIB=$PWD/phylomap_amd/libvariant_w2.so python tools/probe_phases.py 5 16384
echo "4 waves (128 regs), n-row LDS"; PHM_LIB=$PWD/phylomap_amd/libvariant_w4.so python tools/probe_phases.py 5 16384
