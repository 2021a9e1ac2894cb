#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1
tail -n 5 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -n 2 $O/smoke.log
timeout 1200 python bench.py > $O/bench_full.log 2>$O/bench_err.log
tail -n 3 $O/bench_err.log
python - <<PY
import json
d = json.loads(open('$O/bench_full.log').read().strip().splitlines()[-1])
print('headline', d['value'], d['ms_per_step'], d['config'].get('stream_census'), d['config'].get('stream_plan'), d['config'].get('dependent_kernel_interval_us'))
print('whole_step', d['whole_step'])
print('roofline', {k: d['roofline'][k] for k in ('kernel', 'achieved', 'attainable', 'frac', 'algorithmic_frac', 'avg_launch_us', 'traffic', 'algorithmic_bytes_per_launch', 'launches_per_step')})
print('cpu', {k: d['cpu_baseline'][k] for k in ('value', 'cores', 'kind')}, 'parity', d['parity'])
e = d['extra']
print('fwd_only', {k: e['fwd_only'][k] for k in ('value', 'ms_per_step', 'launch', 'eager')})
print('fwd_hard_uniform', {k: (v if not isinstance(v, dict) else {kk: v[kk] for kk in ('value', 'ms_per_step', 'launch', 'eager') if kk in v}) for k, v in e['fwd_hard_uniform'].items() if k != 'workload'})
print('train_hard', {k: (v if not isinstance(v, dict) else v.get('ms_per_step')) for k, v in e['train_hard'].items() if k != 'workload'})
print('config_S', e['config_S']['value'], e['config_S']['ms_per_step'], e['config_S']['frac_of_fp32_mfma_peak'], e['config_S']['dominant_kernel'])
print('affect', e['affect_mosei'].get('train_step'), e['affect_mosei'].get('dynmmnet_3expert'))
PY
