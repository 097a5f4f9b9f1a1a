cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for C in 5; do
  for P in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES"; do
    tag=$(echo $P | cut -d' ' -f1)
    timeout -k 10 300 rocprofv3 --pmc $P --output-format csv -d gpurun_out/pmc3_c${C}_${tag} -o x -- python3 bench.py --config $C --steps 3 --warmup 3 --no-cpu --no-extras > gpurun_out/pmc3_c${C}_${tag}.log 2>&1 || { echo "pmc C$C $tag failed"; tail -3 gpurun_out/pmc3_c${C}_${tag}.log; }
  done
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt3_c${C} -o x -- python3 bench.py --config $C --steps 6 --warmup 4 --no-cpu --no-extras > gpurun_out/kt3_c${C}.log 2>&1 || tail -5 gpurun_out/kt3_c${C}.log
  python tools/summarise_pmc.py $C 3 3 gpurun_out/pmc3_c${C}_FETCH_SIZE/x_counter_collection.csv gpurun_out/pmc3_c${C}_WRITE_SIZE/x_counter_collection.csv gpurun_out/pmc3_c${C}_SQ_WAVES/x_counter_collection.csv > gpurun_out/r02_pmc_C${C}_summary_v3.json 2> gpurun_out/summarise_c$C.err || tail -3 gpurun_out/summarise_c$C.err
  cp gpurun_out/kt3_c${C}/x_kernel_stats.csv gpurun_out/r02_kernel_stats_C${C}_v3.csv
  echo "C$C done"; head -6 gpurun_out/r02_kernel_stats_C${C}_v3.csv | cut -c1-160
done
