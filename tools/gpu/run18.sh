cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
(time timeout -k 10 900 python -m pytest tests -m gpu -q -x) > gpurun_out/r02_gpu_tests.log 2>&1 || { tail -40 gpurun_out/r02_gpu_tests.log; exit 1; }
tail -6 gpurun_out/r02_gpu_tests.log
(time timeout -k 10 900 python bench.py --steps 20 --warmup 5) > gpurun_out/r02_bench_full.json 2> gpurun_out/r02_bench_full.err || { tail -20 gpurun_out/r02_bench_full.err; exit 1; }
tail -4 gpurun_out/r02_bench_full.err
for C in 3 4 5; do
  for P in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES"; do
    tag=$(echo $P | cut -d' ' -f1)
    echo "pmc C$C $tag"
    timeout -k 10 400 rocprofv3 --pmc $P --output-format csv -d gpurun_out/pmc_c${C}_${tag} -o x -- python3 bench.py --config $C --steps 3 --warmup 3 --no-cpu --no-extras > gpurun_out/pmc_c${C}_${tag}.log 2>&1 || { echo failed; tail -3 gpurun_out/pmc_c${C}_${tag}.log; }
  done
  echo "trace C$C"
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_c${C} -o x -- python3 bench.py --config $C --steps 6 --warmup 4 --no-cpu --no-extras > gpurun_out/kt_c${C}.log 2>&1 || tail -3 gpurun_out/kt_c${C}.log
  grep -o '"avg_launch_ms": [0-9.]*' gpurun_out/kt_c${C}.log | head -2
done
