set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
(time timeout -k 10 900 python -m pytest tests -m gpu -q -x --durations=8) > gpurun_out/r02_gpu_tests.log 2>&1 || { tail -60 gpurun_out/r02_gpu_tests.log; exit 1; }
tail -16 gpurun_out/r02_gpu_tests.log
(time timeout -k 10 600 python bench.py --steps 20 --warmup 5) > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err || { tail -30 gpurun_out/r02_bench.err; exit 1; }
tail -5 gpurun_out/r02_bench.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r02_bench.json').readline())
print(json.dumps({k:v for k,v in j.items() if k not in ('configs',)}, indent=1)[:3500])
for k,v in j.get('configs',{}).items():
    print(k, json.dumps(v)[:900])
PY
