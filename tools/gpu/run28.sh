cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x -k "wide or ks_sweep or c4 or c5 or golden or many_tiles" 2>&1 | tail -2
for S in 1024 4096; do echo -n "C4 default S=$S: "; python tools/probe.py $S 8 1 4 tiles | tail -1 | cut -c1-95; done
for S in 8192 16384 65536; do for lib in libphylomap_hip.so libvariant_msplit.so; do echo -n "C4 $lib S=$S: "; PHM_LIB=$PWD/phylomap_amd/$lib python tools/probe.py $S 8 1 4 tiles | tail -1 | cut -c1-95; done; done
for S in 4096 8192 16384; do for lib in libphylomap_hip.so libvariant_msplit.so; do echo -n "C5 $lib S=$S: "; PHM_LIB=$PWD/phylomap_amd/$lib python tools/probe.py $S 8 1 5 tiles | tail -1 | cut -c1-95; done; done
