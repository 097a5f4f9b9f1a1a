set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_wt_c4 -o c4 -- python3 tools/probe.py 16384 4 1 4 tiles > gpurun_out/r02_prof_wt_c4.log 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_wt_c5 -o c5 -- python3 tools/probe.py 4096 4 1 5 tiles > gpurun_out/r02_prof_wt_c5.log 2>&1
python tools/rocpd_summary.py gpurun_out/prof_wt_c4/c4_results.db | head -12
python tools/rocpd_summary.py gpurun_out/prof_wt_c5/c5_results.db | head -12
