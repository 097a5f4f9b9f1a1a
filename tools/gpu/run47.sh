cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x -k "wide or ks_sweep or c4 or c5 or golden or capacity or random or many_tiles" > gpurun_out/r02_wt_test.log 2>&1 || { tail -60 gpurun_out/r02_wt_test.log; exit 1; }
tail -2 gpurun_out/r02_wt_test.log
echo "512-thread workgroups, 32-bit offsets"
python tools/probe_phases.py 5 16384
python tools/probe_phases.py 4 65536
echo "640-thread workgroups"
PHM_LIB=$PWD/phylomap_amd/libvariant_b640.so python tools/probe_phases.py 5 16384
PHM_LIB=$PWD/phylomap_amd/libvariant_b640.so python tools/probe_phases.py 5 4096
