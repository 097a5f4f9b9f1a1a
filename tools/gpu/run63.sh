cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_00_gpu_multirank.py -m gpu -q -x -k "not wide and not c4 and not c5 and not expm and not pade" > gpurun_out/r02_t_test.log 2>&1 || { tail -40 gpurun_out/r02_t_test.log; exit 1; }
tail -2 gpurun_out/r02_t_test.log
for S in 64 1024 4096 16384; do
echo -n "new  S=$S: "; python tools/probe_phases.py 3 $S | cut -d: -f2-
echo -n "head S=$S: "; PHM_LIB=$PWD/phylomap_amd/libvariant_head.so python tools/probe_phases.py 3 $S | cut -d: -f2-
done
echo -n "new  C2 4096: "; python tools/probe_phases.py 2 4096 | cut -d: -f2-
echo -n "head C2 4096: "; PHM_LIB=$PWD/phylomap_amd/libvariant_head.so python tools/probe_phases.py 2 4096 | cut -d: -f2-
