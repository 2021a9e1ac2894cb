"""Diagnostic (not collected by pytest): the tie-aware block comparison of tests/test_hip_blocks.py over many input draws and both
training-forward kernels.  python tests/tie_sweep.py [draws]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from dynmm_amd import ops  # noqa: E402
from dynmm_amd.nn.blocks import NonBottleneck1D  # noqa: E402
from dynmm_amd.nn.decoder import DecoderModule  # noqa: E402
from oracle import dynmm_oracle as O  # noqa: E402
from tests import test_hip_blocks as T  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
fails = 0
for fwd in ('all', 'dgrad'):
    ops.WINO = fwd
    for seed in range(n):
        for name, mk, ref, ins in (
            ('decoder', lambda: DecoderModule(128, 128, 3, 40), lambda sd, x, skip, tr: O.decoder_module(sd, 'm', x, skip, tr, 3),
             [T.rnd(3, 128, 12, 16, seed=seed), T.rnd(3, 128, 24, 32, seed=5 + seed)]),
            ('nb1d', lambda: NonBottleneck1D(128, 128), lambda sd, x, tr: O.non_bottleneck_1d(sd, 'm', x, tr),
             [T.rnd(3, 128, 24, 32, seed=seed)])):
            try:
                T.run_pair(mk(), ref, ins)
            except AssertionError as e:
                fails += 1
                print(f'FAIL {fwd} {name} seed {seed}: {str(e)[:200]}', flush=True)
print(f'{2 * 2 * n - fails} of {2 * 2 * n} comparisons pass at the strict bars (outputs {T.TOL}, gradients {T.GTOL})')
