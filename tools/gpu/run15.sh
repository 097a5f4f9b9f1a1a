set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for n in 20 32 48 61 64; do timeout -k 10 120 python tools/probe_expm.py $n 65536 2>&1 | grep pade; done | tee gpurun_out/r02_probe_expm.log
