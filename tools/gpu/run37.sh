cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
echo "main (3 waves per SIMD at 61 states)"
python tools/probe_phases.py 4 65536
python tools/probe_phases.py 5 16384
echo "variant: two passes over the row, 8 states at a time, 55 registers"
PHM_LIB=$PWD/phylomap_amd/libvariant_down2.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x -k "wide or ks_sweep or c4 or c5 or many_tiles" 2>&1 | tail -2
PHM_LIB=$PWD/phylomap_amd/libvariant_down2.so python tools/probe_phases.py 4 65536
PHM_LIB=$PWD/phylomap_amd/libvariant_down2.so python tools/probe_phases.py 5 16384
PHM_LIB=$PWD/phylomap_amd/libvariant_down2.so python tools/probe_phases.py 4 4096
