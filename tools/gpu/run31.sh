cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python tools/probe_phases.py 3 16384
PHM_LIB=$PWD/phylomap_amd/libvariant_uniform.so python tools/probe_phases.py 3 16384
python tools/probe_phases.py 2 65536
PHM_LIB=$PWD/phylomap_amd/libvariant_uniform.so python tools/probe_phases.py 2 65536
