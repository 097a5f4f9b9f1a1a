cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x -k "tiles or c1 or c2 or c3 or golden or mapping" > gpurun_out/r02_t_test.log 2>&1 || { tail -40 gpurun_out/r02_t_test.log; exit 1; }
tail -2 gpurun_out/r02_t_test.log
echo "persistent 2048"; python tools/probe_phases.py 3 16384; python tools/probe_phases.py 2 65536
echo head; PHM_LIB=$PWD/phylomap_amd/libvariant_head.so python tools/probe_phases.py 3 16384; PHM_LIB=$PWD/phylomap_amd/libvariant_head.so python tools/probe_phases.py 2 65536
echo "persistent 4096"; PHM_LIB=$PWD/phylomap_amd/libvariant_p4096.so python tools/probe_phases.py 3 16384
echo "persistent 1024"; PHM_LIB=$PWD/phylomap_amd/libvariant_p1024.so python tools/probe_phases.py 3 16384
