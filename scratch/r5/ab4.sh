cd $GRAFT_REPO_ROOT
O=gpurun_out/r5e
mkdir -p $O
cat > /tmp/noahead.py <<'PY'
import os, sys, runpy
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
from dynmm_amd.nn import net
if sys.argv[1] == 'lockstep':
    net.DEPTH_RUNS_AHEAD = False
sys.argv = [os.path.join(os.environ['GRAFT_REPO_ROOT'], 'bench.py')] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name='__main__')
PY
B="--steps 12 --warmup 3 --no-cpu-baseline --no-extra --no-kernel-timing"
for rep in 1 2 3 4; do
for cfg in lockstep ahead; do
  v=$(timeout 300 python /tmp/noahead.py $cfg $B 2>$O/ab4_err.log | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null)
  echo "$cfg : $v ms" | tee -a $O/ahead_ab.log
done
done
timeout 1500 python -m pytest tests/test_engine.py tests/test_skip_esanet.py tests/test_esanet.py -x -q -m gpu > $O/ahead_tests.log 2>&1
tail -3 $O/ahead_tests.log
timeout 900 python -m pytest tests/test_hip_model.py -x -q -m gpu -k "goldens or equal_decisions or reproducible" > $O/ahead_tests2.log 2>&1
tail -3 $O/ahead_tests2.log
