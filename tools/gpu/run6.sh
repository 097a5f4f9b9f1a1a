set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rocprofv3 -L > gpurun_out/r02_counters.txt 2>&1 || true
for C in 3 4 5; do
  for P in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES"; do
    tag=$(echo $P | cut -d' ' -f1)
    rocprofv3 --pmc $P --output-format csv -d gpurun_out/pmc_c${C}_${tag} -o x -- python3 bench.py --config $C --steps 3 --warmup 3 --no-cpu --no-extras > gpurun_out/pmc_c${C}_${tag}.log 2>&1 || { tail -5 gpurun_out/pmc_c${C}_${tag}.log; }
  done
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_c${C} -o x -- python3 bench.py --config $C --steps 6 --warmup 4 --no-cpu --no-extras > gpurun_out/kt_c${C}.log 2>&1 || tail -5 gpurun_out/kt_c${C}.log
  tail -c 600 gpurun_out/kt_c${C}.log
done
find gpurun_out -name "*.csv" | head -40
