cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 300 python tools/probe_two_streams.py 3 16384
timeout -k 10 300 python tools/probe_two_streams.py 4 32768
timeout -k 10 300 python tools/probe_two_streams.py 5 8192
