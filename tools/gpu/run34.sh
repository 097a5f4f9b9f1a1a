cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "expm or pade or dic" 2>&1 | tail -5
timeout -k 5 120 python tools/debug_pade.py | tail -10
for n in 20 32 48 61 64; do timeout -k 10 120 python tools/probe_expm.py $n 65536 2>&1 | grep "pade mfma"; done
