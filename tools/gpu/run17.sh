cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 300 python tools/probe_small_S.py 4 2>&1 | tee gpurun_out/r02_probe_small_S_C4.log
timeout -k 10 300 python tools/probe_small_S.py 5 2>&1 | tee gpurun_out/r02_probe_small_S_C5.log
