"""A static gate on the compiled kernels (no GPU): loads must not be serialised behind per-element predicates.

`v = ok ? p[i] : 0.f` inside an unrolled loop compiles to a divergent branch per element whose load is followed by its own
`s_waitcnt vmcnt(0)` — N dependent round trips instead of N loads in flight.  Round 6 found the pattern in the ModalityDynMM
LayerNorm / attention kernels (8 / 24 per workgroup) and in the loss head's prologue (34 forward, 110 backward) and replaced every
such load by an unconditional one on a clamped address with the value selected afterwards (DESIGN.md §4 "Round 6"); this test keeps
it that way.  scratch/r6/serial_loads.py counts, per kernel of the gfx950 ISA, the loads whose next memory event is a full
vmcnt(0) wait with no other load in between.  The budgets below are today's counts: the eight that remain in the sequence kernels
are the byte loads of the injected-keep-flag path the tests use (DropState::keep8 / row8 with `mask`), not the generator path."""
import importlib.util
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

BUDGET = {
    'seq.hip': {'ln_fwd_kernelILi1E': 10, 'ln_bwd_dx_kernelILi1E': 10, 'mha_fwd_kernelILi24ELi4E': 10, 'mha_bwd_kernelILi24ELi4E': 10,
                'mha_fwd_kernelILi12ELi4E': 10, 'mha_bwd_kernelILi12ELi4E': 10},
    'tail.hip': {'up2ce_fwd_kernel': 4, 'up2ce_bwd_kernel': 4},
}


@pytest.mark.skipif(shutil.which('hipcc') is None, reason='needs hipcc (cross-compiles gfx950 without a GPU)')
@pytest.mark.parametrize('src', sorted(BUDGET))
def test_loads_are_not_serialised_behind_predicates(src):
    spec = importlib.util.spec_from_file_location('serial_loads', os.path.join(ROOT, 'scratch', 'r6', 'serial_loads.py'))
    sl = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sl)
    rows = sl.compile_and_scan(os.path.join(ROOT, 'dynmm_amd', 'csrc', src))
    assert rows, 'no kernels found in the ISA'
    for key, budget in BUDGET[src].items():
        hits = [(n, nl, k) for n, nl, k in rows if key in k]
        assert hits, f'{key}: kernel not found (renamed? update the budget table)'
        for n, nl, k in hits:
            assert n <= budget, f'{k}: {n} of {nl} loads wait alone (budget {budget})'
