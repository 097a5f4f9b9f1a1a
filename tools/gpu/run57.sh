cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export PHM_LIB=$PWD/phylomap_amd/libvariant_pad.so
for PAD in 0 40; do echo "pad $PAD KB"; PHM_WT_PAD_LDS=$PAD python tools/probe_phases.py 5 16384; done
