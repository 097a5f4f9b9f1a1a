cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
echo "main (chunk 16 at 61 states)"
python tools/probe_phases.py 4 65536
python tools/probe_phases.py 5 16384
echo "chunk 32 at 61 states"
PHM_LIB=$PWD/phylomap_amd/libvariant_down2.so python tools/probe_phases.py 4 65536
for W in 6 8; do echo "branch kernel at $W waves per SIMD"
PHM_LIB=$PWD/phylomap_amd/libvariant_b$W.so python tools/probe_phases.py 4 65536
PHM_LIB=$PWD/phylomap_amd/libvariant_b$W.so python tools/probe_phases.py 5 16384
done
