cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "TCC_EA0_[A-Z0-9_]*\|TCC_[A-Z_]*ATOMIC[A-Z_]*\|TCC_WRITE[A-Z_]*\|TCC_EA_[A-Z0-9_]*" | sort -u | head -60 > gpurun_out/tcc_counters.txt
cat gpurun_out/tcc_counters.txt | tr '\n' ' '
i=0
for P in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_ATOMIC_sum TCC_ATOMIC_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  i=$((i+1))
  echo "pass $i: $P"
  timeout -k 5 150 rocprofv3 --pmc $P --output-format csv -d gpurun_out/pmcw_$i -o x -- python3 tools/probe.py 4096 3 1 5 tiles > gpurun_out/pmcw_$i.log 2>&1 || { echo "pass $i failed"; grep -m2 -i "error\|exceeds" gpurun_out/pmcw_$i.log; }
done
python - <<'PY'
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float))
cnt=collections.defaultdict(int)
for f in glob.glob('gpurun_out/pmcw_*/x_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        for key in ('wt_branch','wt_down','wt_up_kernel','wt_stats'):
            if key in k:
                agg[key][r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in agg.items():
    print(k, {a:'%.4g'%b for a,b in sorted(v.items())})
PY
