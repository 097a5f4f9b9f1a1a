cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
echo main; python tools/probe_phases.py 3 16384; python tools/probe_phases.py 2 65536
for G in 32 64; do echo "group $G"
PHM_LIB=$PWD/phylomap_amd/libvariant_g$G.so python tools/probe_phases.py 3 16384
PHM_LIB=$PWD/phylomap_amd/libvariant_g$G.so python tools/probe_phases.py 2 65536
done
