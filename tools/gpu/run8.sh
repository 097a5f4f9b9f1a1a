set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r02_gpu_tests.log 2>&1 || { tail -60 gpurun_out/r02_gpu_tests.log; exit 1; }
tail -3 gpurun_out/r02_gpu_tests.log
timeout -k 10 300 python tools/fuzz_mappings.py 11 80 > gpurun_out/r02_fuzz.log 2>&1 || { tail -30 gpurun_out/r02_fuzz.log; exit 1; }
tail -1 gpurun_out/r02_fuzz.log
for c in 3 2 4 5; do python bench.py --config $c --no-cpu --no-extras --steps 12 --warmup 6 | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print(j['config']['workload'][:3], j['config']['replicas_per_gpu'], '%.4g'%j['value'], '%.3f ms'%j['ms_per_step'], 'frac', round(j['roofline']['frac'],4), j['roofline'].get('sweep',{}).get('frac'), j.get('phases_ms_per_sweep'))"; done
: > gpurun_out/r02_squamate_dic.jsonl
for seed in 101 1; do for model in 2 4; do timeout -k 10 300 python tools/squamate_dic/run_dic.py --model $model --engine hip --N 10000 --seed $seed >> gpurun_out/r02_squamate_dic.jsonl 2>gpurun_out/dic.err || tail -3 gpurun_out/dic.err; done; done
cat gpurun_out/r02_squamate_dic.jsonl
