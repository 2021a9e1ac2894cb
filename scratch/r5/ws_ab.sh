cd $GRAFT_REPO_ROOT
O=gpurun_out/r5i
mkdir -p $O
cat > /tmp/ws.py <<'PY'
import os, sys, runpy
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
from dynmm_amd import ops
ops.WGRAD_STREAMS = int(sys.argv[1])
ops.WGRAD_GROUP = int(sys.argv[2])
sys.argv = [os.path.join(os.environ['GRAFT_REPO_ROOT'], 'bench.py')] + sys.argv[3:]
runpy.run_path(sys.argv[0], run_name='__main__')
PY
B="--steps 12 --warmup 3 --no-cpu-baseline --no-extra --no-kernel-timing"
for rep in 1 2 3; do
for cfg in "2 4" "3 4" "1 4" "2 2" "2 6"; do
  set -- $cfg
  v=$(timeout 300 python /tmp/ws.py $1 $2 $B 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null)
  echo "streams=$1 group=$2 : $v ms" | tee -a $O/ws_ab.log
done
done
