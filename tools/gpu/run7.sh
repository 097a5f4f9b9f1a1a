set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "capacity" > gpurun_out/r02_cap_test.log 2>&1 || { tail -60 gpurun_out/r02_cap_test.log; exit 1; }
tail -3 gpurun_out/r02_cap_test.log
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r02_gpu_tests.log 2>&1 || { tail -60 gpurun_out/r02_gpu_tests.log; exit 1; }
tail -3 gpurun_out/r02_gpu_tests.log
python bench.py --no-cpu --no-extras --steps 20 --warmup 5 | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['sweep']['frac'], j['phases_ms_per_sweep'])"
