cd $GRAFT_REPO_ROOT
O=gpurun_out/r5i; mkdir -p $O
cat > /tmp/g.py <<'PY'
import os, sys, runpy
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
from dynmm_amd.nn import net
net.GATE_BESIDE_STAGE1 = sys.argv[1] == 'beside'
sys.argv = [os.path.join(os.environ['GRAFT_REPO_ROOT'], 'bench.py')] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name='__main__')
PY
B="--steps 12 --warmup 3 --no-cpu-baseline --no-extra --no-kernel-timing"
for rep in 1 2 3 4; do for c in main beside; do
  v=$(timeout 300 python /tmp/g.py $c $B 2>$O/g_err.log | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null)
  echo "gate $c : $v ms" | tee -a $O/gate2_ab.log
done; done
tail -2 $O/g_err.log
