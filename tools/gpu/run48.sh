cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for i in 1 2; do
echo "new"; python tools/probe_phases.py 4 65536; python tools/probe_phases.py 5 16384
echo "head"; PHM_LIB=$PWD/phylomap_amd/libvariant_head.so python tools/probe_phases.py 4 65536; PHM_LIB=$PWD/phylomap_amd/libvariant_head.so python tools/probe_phases.py 5 16384
done
