cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r02_gpu_tests.log 2>&1 || { tail -40 gpurun_out/r02_gpu_tests.log; exit 1; }
tail -2 gpurun_out/r02_gpu_tests.log
(time timeout -k 10 900 python bench.py --steps 20 --warmup 5) > gpurun_out/r02_bench_full.json 2> gpurun_out/r02_bench_full.err || { tail -20 gpurun_out/r02_bench_full.err; exit 1; }
tail -4 gpurun_out/r02_bench_full.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r02_bench_full.json').readline())
print('C3 value %.4g ms %.3f frac %.3f sweep %.3f'%(j['value'],j['ms_per_step'],j['roofline']['frac'],j['roofline']['sweep']['frac']), j['phases_ms_per_sweep'])
for k,v in j['configs'].items():
    r=v.get('roofline',{})
    print(k,'%.4g'%v['realisations_per_s'], v.get('replicas'), 'frac',r.get('frac'), 'sweep',r.get('sweep',{}).get('frac'), v.get('whole_call_ms_at_N_1000'), (v.get('cpu_baseline') or {}).get('value'))
print(j['cpu_baseline']['value'], j['cpu_baseline']['faithful_value'], j['cpu_baseline']['all_cores']['value'], j['expm_per_s'])
PY
