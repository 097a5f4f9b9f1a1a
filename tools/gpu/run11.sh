set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_00_gpu_multirank.py -m gpu -q -x -k "not wide" > gpurun_out/r02_t.log 2>&1 || { tail -60 gpurun_out/r02_t.log; exit 1; }
tail -2 gpurun_out/r02_t.log
for c in 3; do python bench.py --config $c --no-cpu --no-extras --steps 20 --warmup 5 | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print(j['config']['workload'][:3], j['config']['replicas_per_gpu'], '%.4g'%j['value'], '%.3f ms'%j['ms_per_step'], j['roofline'].get('sweep',{}).get('frac'), j.get('phases_ms_per_sweep'))"; done
