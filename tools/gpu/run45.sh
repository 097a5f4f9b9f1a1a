cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for KT in 0 2 4 8; do echo "KT=$KT"; PHM_WT_KT=$KT python tools/probe_phases.py 5 16384; done
PHM_WT_KT=4 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x -k "wide or ks_sweep or c5 or golden or many_tiles" 2>&1 | tail -2
