cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
PHM_LIB=$PWD/phylomap_amd/libvariant_padeph.so timeout -k 5 120 python tools/probe_pade_phases.py 61 8192
PHM_LIB=$PWD/phylomap_amd/libvariant_padeph.so timeout -k 5 120 python tools/probe_pade_phases.py 32 8192
