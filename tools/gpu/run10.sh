set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
(time python bench.py --steps 20 --warmup 5) > gpurun_out/r02_bench_full.json 2> gpurun_out/r02_bench_full.err || { tail -20 gpurun_out/r02_bench_full.err; exit 1; }
tail -4 gpurun_out/r02_bench_full.err
python -c "
import json
j=json.loads(open('gpurun_out/r02_bench_full.json').readline())
print('value %.4g'%j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['sweep']['frac'], j['roofline']['traffic'], j['pruning_sweep'])
for k,v in j['configs'].items(): print(k, '%.4g'%v['realisations_per_s'], v.get('roofline',{}).get('frac'), v.get('roofline',{}).get('sweep',{}).get('frac'), v.get('pruning_sweep',{}).get('mfma'))
print(j['cpu_baseline']); print(j['expm_per_s'])
"
