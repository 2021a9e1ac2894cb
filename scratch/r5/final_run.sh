# full validation of the tree: GPU tests, the default bench line, the profiling recipe, an A/B of the deferred stem BatchNorm backward
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5d
mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gputests.log 2>&1
tail -3 $O/gputests.log
timeout 900 python bench.py > $O/bench_full.log 2> $O/bench_full.err
tail -1 $O/bench_full.log | cut -c1-400
sh profiles/r05_recipe.sh > $O/recipe.log 2>&1
tail -32 $O/recipe.log
cat > /tmp/nodefer.py <<'PY'
import os, sys, runpy
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
from dynmm_amd import ops
if sys.argv[1] == 'nodefer':
    ops.stem_bn_defer = lambda x, bn: x
sys.argv = [os.path.join(os.environ['GRAFT_REPO_ROOT'], 'bench.py')] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name='__main__')
PY
B="--steps 12 --warmup 3 --no-cpu-baseline --no-extra --no-kernel-timing"
for rep in 1 2 3; do
for cfg in nodefer defer; do
  v=$(timeout 300 python /tmp/nodefer.py $cfg $B 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null)
  echo "$cfg : $v ms" | tee -a $O/defer_ab.log
done
done
