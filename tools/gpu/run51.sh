cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python tools/probe_single_chain.py 2000
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_single -o x -- python3 tools/probe_single_chain.py 400 > gpurun_out/kt_single.log 2>&1
tail -1 gpurun_out/kt_single.log
cut -d, -f1-4 gpurun_out/kt_single/x_kernel_stats.csv | head -12
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/kt_single/x_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# one steady-state sweep: find the last 'narrow_emit' and print the preceding launches with gaps
idx=[i for i,r in enumerate(rows) if 'narrow_emit' in r['Kernel_Name']]
a,b=idx[-3],idx[-2]
prev=int(rows[a]['End_Timestamp'])
tot=0
for r in rows[a+1:b+1]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    print('%-28s run %6.1f us  gap before %5.1f us'%(r['Kernel_Name'].split('::')[-1][:28],(e-s)/1e3,(s-prev)/1e3))
    prev=e
print('sweep span %.1f us'%((int(rows[b]['End_Timestamp'])-int(rows[a]['End_Timestamp']))/1e3))
PY
