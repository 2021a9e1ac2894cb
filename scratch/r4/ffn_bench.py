"""Isolated timing of the fused feed-forward kernels: python scratch/r4/ffn_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from dynmm_amd import ops, ops_seq as S  # noqa: E402

lib = S._lib()
ops.manual_seed(1)
B, T, F = 128, 50, 2048
for D in (120, 60, 10):
    x = torch.randn(B, D, T, device='cuda')
    w1 = torch.randn(F, D, device='cuda') * 0.05
    b1 = torch.randn(F, device='cuda') * 0.05
    w2 = torch.randn(D, F, device='cuda') * 0.02
    ns = lib.dynmm_ffn_nsplit(B, D, T, F)
    hid = torch.empty(B, F, T, device='cuda')
    dhid = torch.empty_like(hid)
    parts = torch.empty(ns, B, D, T, device='cuda')
    d = S.Drop(0.1, 7, 'dropout', (B, F, T), x.device)
    st = torch.cuda.current_stream().cuda_stream

    def fwd():
        S.L.check(lib.dynmm_ffn_fwd(x.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), hid.data_ptr(), parts.data_ptr(),
                                    B, D, T, F, ns, S._drop_arg(d), st), 'fwd')

    def bwd():
        S.L.check(lib.dynmm_ffn_bwd_data(x.data_ptr(), hid.data_ptr(), w1.data_ptr(), w2.data_ptr(), dhid.data_ptr(),
                                         parts.data_ptr(), B, D, T, F, ns, 0.1, st), 'bwd')
    for name, fn, gf in (('fwd', fwd, 4), ('bwd', bwd, 4)):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / 50
        print(f'D={D} nsplit={ns} {name}: {us:.1f} us, {gf * D * F * B * T / us / 1e6:.1f} TF/s', flush=True)
