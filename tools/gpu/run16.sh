set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_configs.py -m gpu -q -x -k "many_tiles" --durations=3 2>&1 | tail -6
