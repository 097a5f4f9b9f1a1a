set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
PHM_ALL_RANKS_ON_DEVICE0=1 PHM_DIST_BACKEND=gloo timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 6 --warmup 3 --replicas 4096 > gpurun_out/r02_bench_2rank.json 2> gpurun_out/r02_bench_2rank.err || { tail -30 gpurun_out/r02_bench_2rank.err; exit 1; }
python -c "
import json
j=json.loads(open('gpurun_out/r02_bench_2rank.json').readline()); print(j['n_gpus'], '%.4g'%j['value'], j['ms_per_step'], j['config'])"
python bench.py --gpus 2 2>&1 | tail -2 || true
