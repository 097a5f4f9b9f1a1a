set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
(time python -m pytest tests/test_gpu_configs.py -m gpu -q -x --durations=15) > gpurun_out/r02_configs_test.log 2>&1 || { tail -40 gpurun_out/r02_configs_test.log; exit 1; }
tail -25 gpurun_out/r02_configs_test.log
python tools/probe.py 4096 6 1 4 branches > gpurun_out/r02_base_C4.log 2>&1; cat gpurun_out/r02_base_C4.log
python tools/probe.py 1024 6 1 5 branches > gpurun_out/r02_base_C5.log 2>&1; cat gpurun_out/r02_base_C5.log
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_c4 -o c4 -- python3 tools/probe.py 4096 4 1 4 branches > gpurun_out/r02_prof_c4.log 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_c5 -o c5 -- python3 tools/probe.py 1024 4 1 5 branches > gpurun_out/r02_prof_c5.log 2>&1
ls gpurun_out/prof_c4 gpurun_out/prof_c5
