cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for S in 2048 4096 8192 16384 32768; do for thr in 1 100000; do echo -n "C4 S=$S few_thr=$thr: "; PHM_WT_FEW_TILES=$thr python tools/probe.py $S 8 1 4 tiles | tail -1 | cut -c1-90; done; done
for S in 1024 2048 4096 8192; do for thr in 1 100000; do echo -n "C5 S=$S few_thr=$thr: "; PHM_WT_FEW_TILES=$thr python tools/probe.py $S 8 1 5 tiles | tail -1 | cut -c1-90; done; done
