cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for i in 1 2 3; do
echo -n "new  "; python tools/probe_phases.py 4 65536 | cut -d';' -f2
echo -n "head "; PHM_LIB=$PWD/phylomap_amd/libvariant_head.so python tools/probe_phases.py 4 65536 | cut -d';' -f2
done
