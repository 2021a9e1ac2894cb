#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6w; mkdir -p $O
timeout 900 python -m pytest tests/test_engine.py tests/test_hip_ops.py -x -q -m gpu -k "infer or eval_run or evaluate or gate_head or stream" > $O/pytest.log 2>&1
tail -n 5 $O/pytest.log
timeout 600 python bench.py --no-cpu-baseline --steps 8 > $O/bench.log 2>$O/bench_err.log
tail -n 3 $O/bench_err.log
python - <<PY
import json
d = json.loads(open('$O/bench.log').read().strip().splitlines()[-1])
print('headline', d['value'], d['ms_per_step'])
e = d['extra']
print('fwd_only', {k: e['fwd_only'].get(k) for k in ('value', 'ms_per_step', 'launch', 'launch_policy', 'auto_timing', 'eager')})
print('fwd_hard_uniform', {k: (v if not isinstance(v, dict) else {kk: v.get(kk) for kk in ('value', 'ms_per_step', 'launch', 'auto_timing', 'eager')}) for k, v in e['fwd_hard_uniform'].items() if k != 'workload'})
PY
