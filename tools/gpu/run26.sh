cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_c4_1024 -o x -- python3 tools/probe.py 1024 20 1 4 tiles > gpurun_out/kt_c4_1024.log 2>&1
tail -1 gpurun_out/kt_c4_1024.log | cut -c1-100
head -7 gpurun_out/kt_c4_1024/x_kernel_stats.csv | cut -c1-160
