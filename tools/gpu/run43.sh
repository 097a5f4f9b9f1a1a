cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
echo main; python tools/probe_phases.py 5 16384; python tools/probe_phases.py 4 65536
for G in 16 32; do echo "group $G"
PHM_LIB=$PWD/phylomap_amd/libvariant_g$G.so python tools/probe_phases.py 5 16384
PHM_LIB=$PWD/phylomap_amd/libvariant_g$G.so python tools/probe_phases.py 4 65536
done
