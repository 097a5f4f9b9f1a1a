set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x -k "wide or ks_sweep or c4 or c5 or golden or capacity or random" > gpurun_out/r02_wt_test.log 2>&1 || { tail -60 gpurun_out/r02_wt_test.log; exit 1; }
tail -2 gpurun_out/r02_wt_test.log
for c in 4 5; do python bench.py --config $c --no-cpu --no-extras --steps 12 --warmup 6 | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print(j['config']['workload'][:3], j['config']['replicas_per_gpu'], '%.4g'%j['value'], '%.3f ms'%j['ms_per_step'], j.get('phases_ms_per_sweep'))"; done
