cd $GRAFT_REPO_ROOT
O=gpurun_out/r5i
mkdir -p $O
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_blocks.py -x -q -m gpu -k "normalise or non_bottleneck or encoder_stage or strided_block" > $O/bnin_tests.log 2>&1
tail -6 $O/bnin_tests.log
cat > /tmp/noload.py <<'PY'
import os, sys, runpy
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
from dynmm_amd import ops
if sys.argv[1] == 'twopass':
    ops.BN_ON_LOAD = False
sys.argv = [os.path.join(os.environ['GRAFT_REPO_ROOT'], 'bench.py')] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name='__main__')
PY
B="--steps 12 --warmup 3 --no-cpu-baseline --no-extra --no-kernel-timing"
for rep in 1 2 3 4; do
for cfg in twopass onload; do
  v=$(timeout 300 python /tmp/noload.py $cfg $B 2>$O/ab_err.log | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null)
  echo "$cfg : $v ms" | tee -a $O/bnin2_ab.log
done
done
tail -3 $O/ab_err.log
timeout 1500 python -m pytest tests/test_engine.py tests/test_hip_model.py -x -q -m gpu -k "train or step or grad or golden" > $O/model_tests.log 2>&1
tail -3 $O/model_tests.log
