#!/bin/bash
cd $GRAFT_REPO_ROOT
bash scratch/r6/ab_lib.sh r6g scratch/r6/libdynmm_vocc3.so
timeout 600 python -m pytest tests/test_engine.py tests/test_hip_ops.py tests/test_hip_blocks.py -x -q -m gpu -k "infer_step or wino or conv2d or block or encoder" > gpurun_out/r6g/pytest.log 2>&1
tail -n 5 gpurun_out/r6g/pytest.log
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r6g/bench.log 2>gpurun_out/r6g/bench_err.log
tail -n 5 gpurun_out/r6g/bench_err.log
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r6g/bench.log').read().strip().splitlines()[-1])
print('headline', d['value'], d['ms_per_step'], d['config'].get('stream_census'))
print('roofline', {k: d['roofline'][k] for k in ('kernel', 'achieved', 'attainable', 'frac', 'algorithmic_frac', 'avg_launch_us')})
e = d['extra']
print('fwd_only', {k: e['fwd_only'][k] for k in ('value', 'ms_per_step', 'launch', 'eager')})
print('fwd_hard_uniform', {k: (v if not isinstance(v, dict) else {kk: v[kk] for kk in ('value', 'ms_per_step', 'launch', 'eager') if kk in v}) for k, v in e['fwd_hard_uniform'].items() if k != 'workload'})
print('train_hard', {k: (v if not isinstance(v, dict) else v.get('ms_per_step')) for k, v in e['train_hard'].items() if k != 'workload'})
print('config_S', e['config_S']['ms_per_step'], e['config_S']['dominant_kernel'])
print('affect', e['affect_mosei']['train_step'])
PY
