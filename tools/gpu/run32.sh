cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for i in 1 2; do
python tools/probe_phases.py 3 16384
PHM_LIB=$PWD/phylomap_amd/libvariant_old.so python tools/probe_phases.py 3 16384
done
python tools/probe_phases.py 2 65536
PHM_LIB=$PWD/phylomap_amd/libvariant_old.so python tools/probe_phases.py 2 65536
python tools/probe_phases.py 4 65536
PHM_LIB=$PWD/phylomap_amd/libvariant_old.so python tools/probe_phases.py 4 65536
python tools/probe_phases.py 5 16384
PHM_LIB=$PWD/phylomap_amd/libvariant_old.so python tools/probe_phases.py 5 16384
