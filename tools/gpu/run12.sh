set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for lib in libphylomap_hip.so libvariant_k16.so libvariant_k1.so; do PHM_LIB=$PWD/phylomap_amd/$lib python bench.py --config 3 --no-cpu --no-extras --steps 20 --warmup 5 | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('$lib', '%.4g'%j['value'], '%.3f ms'%j['ms_per_step'], j.get('phases_ms_per_sweep'))"; done
