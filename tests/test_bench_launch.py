"""bench.py's own launcher: `python bench.py --gpus N` with no torchrun around it must start N
ranks; under a launcher (WORLD_SIZE set) it must not spawn again and must refuse a --gpus / WORLD_SIZE mismatch.
Runs on CPU: DYNMM_BENCH_LAUNCH_PROBE makes every rank join a gloo group, all-reduce a 1 and exit."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(argv, extra_env=None, drop=('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(DYNMM_BENCH_LAUNCH_PROBE='1', OMP_NUM_THREADS='1')
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(REPO, 'bench.py')] + argv, env=env, capture_output=True,
                          text=True, timeout=600)


def _line(out):
    rows = [json.loads(ln) for ln in out.splitlines() if ln.startswith('{')]
    assert len(rows) == 1, out                     # ONE JSON line, printed by rank 0
    return rows[0]


def test_gpus_flag_spawns_that_many_ranks():
    r = _run(['--gpus', '2', '--steps', '4'])
    assert r.returncode == 0, r.stderr[-2000:]
    line = _line(r.stdout)
    assert line['n_gpus'] == 2 and line['ranks_seen'] == 2 and line['steps'] == 4


def test_eight_ranks_like_the_scaling_run():
    """the driver's largest point (8 x MI355X): launcher, rendezvous on 127.0.0.1 with a free port, the host-thread split
    (OMP_NUM_THREADS = cpus / ranks, so 8 ranks do not oversubscribe the host 8-fold), dmabuf IPC for RCCL, and still
    exactly ONE JSON line."""
    env = {k: v for k, v in os.environ.items()
           if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'OMP_NUM_THREADS')}
    env.update(DYNMM_BENCH_LAUNCH_PROBE='1')
    r = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '8', '--steps', '3', '--warmup', '1'],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _line(r.stdout)
    assert line['n_gpus'] == 8 and line['ranks_seen'] == 8 and line['steps'] == 3
    assert int(line['omp_num_threads']) == max(1, (os.cpu_count() or 8) // 8)
    assert line['master'].startswith('127.0.0.1:') and line['ipc_mode_legacy'] == '0'


def test_single_rank_does_not_spawn():
    r = _run(['--gpus', '1'])
    assert r.returncode == 0, r.stderr[-2000:]
    line = _line(r.stdout)
    assert line['n_gpus'] == 1 and line['ranks_seen'] == 1


def test_world_size_mismatch_is_refused():
    r = _run(['--gpus', '4'], extra_env={'WORLD_SIZE': '1', 'RANK': '0', 'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': '29533'})
    assert r.returncode != 0 and 'WORLD_SIZE=1' in (r.stderr + r.stdout)


def test_roofline_kernel_names_match_the_pmc_summary():
    """bench.py quotes the roofline per compiled kernel instance (kernel_instance) and looks its HBM traffic up under that name in
    profiles/pmc_dominant_kernel.json, which profiles/summarize_pmc.py writes: the two name tables must agree, or `traffic`
    silently becomes null."""
    import importlib.util
    import json
    import os
    import re
    import bench
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, 'profiles', 'summarize_pmc.py')).read()
    keys = set(re.findall(r"'(conv_[a-z0-9_]+<[^']*>)':\s*\(\[?'", src))
    labels = ['conv_wino43_dgrad<co128,1x3>', 'conv_wino43_dgrad<co64,1x3>', 'conv_wino2d_dgrad<co128,3x3>',
              'conv_wino_fwd<co128,1x3>', 'conv_wino_fwd<co64,3x1>', 'conv_wino2d_fwd<co64,3x3>', 'conv_wino_dgrad<co128,3x1>',
              'conv_wino_dgrad<co128,1x3s2>', 'conv_wino_dgrad<co64,3x1s2>', 'conv_wgrad_v6<co128,1x3>', 'conv_wgrad_v6<co64,3x1>',
              'conv_wgrad_v6<co64,3x3>', 'conv_wgrad_s2<co128,3x1>', 'conv_wgrad_s2<co128,1x3>', 'conv_igemm_v5_fwd<128x64,kw1>']
    for lb in labels:
        assert bench.kernel_instance(lb) in keys, (lb, bench.kernel_instance(lb))
    assert bench.executed_fraction('conv_wino43_dgrad<co128,1x3>') == 0.5
    assert abs(bench.executed_fraction('conv_wino_fwd<co64,1x3>') - 2 / 3) < 1e-12
    assert abs(bench.executed_fraction('conv_wino2d_fwd<co64,3x3>') - 4 / 9) < 1e-12
    assert bench.executed_fraction('conv_wino_dgrad<co128,1x3s2>') == 1.0 and bench.executed_fraction('conv_igemm_fwd<128x64>') == 1.0
    assert bench.executed_fraction('conv_wgrad_s2<co128,3x1>') == 1.0          # stride 2: the direct form
    # the committed record carries the kernel the last bench line named
    rec = json.load(open(os.path.join(root, 'profiles', 'pmc_dominant_kernel.json')))
    assert 'conv_wino43_dgrad<horizontal>' in rec and rec['conv_wino43_dgrad<horizontal>']['launches_per_step'] == 76
    assert 'conv_wino_dgrad<vertical>' in rec and rec['conv_wino_dgrad<vertical>']['launches_per_step'] == 76     # (BNRED 0 | 1 | 2 instances)
    assert rec['conv_wgrad_s2<co128,3x1>']['launches_per_step'] == 6


def test_roofline_frac_is_a_fraction():
    """`achieved` stays algorithmic FLOP / time, but `frac` is quoted against what the kernel's algorithm
    can attain (peak / executed share) and never exceeds 1 — the round-5 line carried 1.13 for config S's F(2x2,3x3) kernel."""
    import bench
    # a recorded case: conv_wino2d_dgrad<3x3> at 180.09 TF/s algorithmic (BENCH r05/r06 extra.config_S), + two others
    for label, tf in (('conv_wino2d_dgrad<co128,3x3>', 180.09), ('conv_wino43_dgrad<co128,1x3>', 165.0),
                      ('conv_wino_dgrad<co128,3x1>', 122.5), ('conv_igemm_fwd<128x64>', 100.0)):
        ms = 1.0
        agg = {label: [1, ms, tf * 1e12 * ms * 1e-3, 1e8]}
        r = bench.roofline_of(agg)
        share = bench.executed_fraction(label)
        assert abs(r['achieved'] - tf) < 0.01 and r['peak'] == bench.FP32_MFMA_PEAK_TFLOPS
        assert abs(r['attainable'] - r['peak'] / share) < 0.06
        assert 0 < r['frac'] <= 1 and abs(r['frac'] - tf * share / r['peak']) < 1e-3 and r['frac'] == r['executed_frac']
        assert abs(r['algorithmic_frac'] - tf / r['peak']) < 1e-3
