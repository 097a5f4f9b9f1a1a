cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 5 60 tools/ubench/blocks | tee gpurun_out/r02_ubench_blocks_after.log
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r02_gpu_tests.log 2>&1 || { tail -40 gpurun_out/r02_gpu_tests.log; exit 1; }
tail -2 gpurun_out/r02_gpu_tests.log
for S in 4096 16384; do echo -n "C3 S=$S: "; python tools/probe.py $S 8 1 3 tiles | tail -1 | cut -c1-110; done
echo -n "C2 S=65536 tiles: "; python tools/probe.py 65536 8 1 2 tiles | tail -1 | cut -c1-110
echo -n "C2 S=262144 replicas: "; python tools/probe.py 262144 8 1 2 replicas | tail -1 | cut -c1-110
echo -n "C4 S=65536: "; python tools/probe.py 65536 8 1 4 tiles | tail -1 | cut -c1-110
echo -n "C5 S=16384: "; python tools/probe.py 16384 8 1 5 tiles | tail -1 | cut -c1-110
