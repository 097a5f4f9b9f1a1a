cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
i=0
for P in "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD" "TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  echo "pass $i: $P"
  timeout -k 5 150 rocprofv3 --pmc $P --output-format csv -d gpurun_out/pmcx_$i -o x -- python3 tools/probe.py 16384 3 1 4 tiles > gpurun_out/pmcx_$i.log 2>&1 || { echo "pass $i failed"; grep -m2 -i "error\|exceeds" gpurun_out/pmcx_$i.log; }
done
python - <<'PY'
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob('gpurun_out/pmcx_*/x_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        for key in ('wt_branch','wt_down','wt_up'):
            if key in k:
                agg[key][r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in agg.items():
    print(k, {a:'%.3g'%b for a,b in sorted(v.items())})
PY
