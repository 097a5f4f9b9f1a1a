cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for C in 4 5; do
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_small_c$C -o x -- python3 tools/probe.py 64 20 1 $C tiles > gpurun_out/kt_small_c$C.log 2>&1
tail -2 gpurun_out/kt_small_c$C.log
head -8 gpurun_out/kt_small_c$C/x_kernel_stats.csv | cut -c1-200
done
