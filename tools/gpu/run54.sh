cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x -k "branches or narrow or mapping or dic or squamate or golden or single or qupdate or bf or ks" > gpurun_out/r02_n_test.log 2>&1 || { tail -40 gpurun_out/r02_n_test.log; exit 1; }
tail -2 gpurun_out/r02_n_test.log
python tools/probe_single_chain.py 2000
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_single2 -o x -- python3 tools/probe_single_chain.py 400 > gpurun_out/kt_single2.log 2>&1
cut -d, -f1-4 gpurun_out/kt_single2/x_kernel_stats.csv | head -7
