cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x -k "tiles or c1 or c2 or c3 or golden or mapping" > gpurun_out/r02_t_test.log 2>&1 || { tail -40 gpurun_out/r02_t_test.log; exit 1; }
tail -2 gpurun_out/r02_t_test.log
python tools/probe_phases.py 3 16384; python tools/probe_phases.py 2 65536
echo "main wide"; python tools/probe_phases.py 4 65536; python tools/probe_phases.py 5 16384
echo "persistent wide node draws"; export PHM_LIB=$PWD/phylomap_amd/libvariant_pd.so
python tools/probe_phases.py 4 65536; python tools/probe_phases.py 5 16384
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x -k "wide or ks_sweep or c4 or c5 or many_tiles" 2>&1 | tail -2
