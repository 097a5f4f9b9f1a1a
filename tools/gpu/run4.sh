set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 420 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x -k "wide or ks_sweep or c4 or c5 or golden or capacity_overflow" > gpurun_out/r02_wt_test.log 2>&1 || { tail -60 gpurun_out/r02_wt_test.log; exit 1; }
tail -3 gpurun_out/r02_wt_test.log
timeout -k 10 300 python tools/fuzz_mappings.py 8 60 > gpurun_out/r02_wt_fuzz.log 2>&1 || { tail -30 gpurun_out/r02_wt_fuzz.log; exit 1; }
tail -2 gpurun_out/r02_wt_fuzz.log
for S in 4096 16384; do timeout -k 10 120 python tools/probe.py $S 6 1 4 tiles; done 2>&1 | tee gpurun_out/r02_wt_C4.log
for S in 1024 4096; do timeout -k 10 120 python tools/probe.py $S 6 1 5 tiles; done 2>&1 | tee gpurun_out/r02_wt_C5.log
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_wt_c4 -o c4 -- python3 tools/probe.py 16384 4 1 4 tiles > gpurun_out/r02_prof_wt_c4.log 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_wt_c5 -o c5 -- python3 tools/probe.py 4096 4 1 5 tiles > gpurun_out/r02_prof_wt_c5.log 2>&1
python tools/rocpd_summary.py gpurun_out/prof_wt_c4/c4_results.db | head -6
python tools/rocpd_summary.py gpurun_out/prof_wt_c5/c5_results.db | head -6
