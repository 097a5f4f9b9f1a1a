cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for CFG in "4 64" "4 1024" "3 64" "3 1024"; do
set -- $CFG
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt_gap_$1_$2 -o x -- python3 tools/probe.py $2 30 1 $1 tiles > gpurun_out/kt_gap_$1_$2.log 2>&1
tail -1 gpurun_out/kt_gap_$1_$2.log | cut -c1-100
python - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/kt_gap_$1_$2/x_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
rows=[r for r in rows if 'stats_reduce' not in r['Kernel_Name'] and 'rocclr' not in r['Kernel_Name']]
# last 20 sweeps: find stats kernels
idx=[i for i,r in enumerate(rows) if 'stats_kernel' in r['Kernel_Name']]
a,b=idx[-21],idx[-1]
seg=rows[a+1:b+1]
busy=sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in seg)
span=int(seg[-1]['End_Timestamp'])-int(rows[a]['End_Timestamp'])
print("C$1 S=$2: 20 sweeps: %d launches per sweep, span %.1f us per sweep, kernels busy %.1f us per sweep, gaps %.1f %%"%(len(seg)/20, span/20e3, busy/20e3, 100*(span-busy)/span))
PY
done
