"""Isolated forward / input-gradient convolution launches through the C ABI on the hot shapes of config P (batch 32).
Run once per kernel generation:  DYNMM_IGEMM_V5=0 python scratch/igemm_bench.py ;  DYNMM_IGEMM_V5=1 python ..."""
import ctypes as C
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynmm_amd import lib as L
lib = L.load()
st = torch.cuda.current_stream().cuda_stream
N = int(os.environ.get('BENCH_N', '32'))
SHAPES = [(64, 120, 160, 3, 1), (64, 120, 160, 1, 3), (128, 60, 80, 3, 1), (128, 60, 80, 1, 3), (256, 30, 40, 3, 1),
          (256, 30, 40, 1, 3), (512, 15, 20, 3, 1), (512, 15, 20, 1, 3), (128, 60, 80, 3, 3), (128, 30, 40, 3, 3)]
print('V5 =', os.environ.get('DYNMM_IGEMM_V5', '(default)'), ' N =', N)
tot = {'fwd': 0.0, 'dgrad': 0.0}
for (Cc, H, W, KH, KW) in SHAPES:
    x = torch.randn(N, Cc, H, W, device='cuda')
    b = torch.randn(Cc, device='cuda')
    y = torch.empty_like(x)
    mask = torch.randn_like(x)
    w = torch.randn(Cc, Cc, KH, KW, device='cuda') * 0.05
    g = L.ConvGeom(N, Cc, H, W, Cc, H, W, KH, KW, 1, 1, KH // 2, KW // 2, Cc)
    wp = torch.empty(KH * KW * Cc * Cc, device='cuda')
    wd = torch.empty(KH * KW * Cc * Cc, device='cuda')
    lib.dynmm_pack_weight(w.data_ptr(), wp.data_ptr(), wd.data_ptr(), Cc, Cc, KH, KW, st)

    def fwd():
        return lib.dynmm_conv2d_fwd(x.data_ptr(), None, wp.data_ptr(), None, b.data_ptr(), None, y.data_ptr(), C.byref(g), 1, st)

    def dgrad():
        return lib.dynmm_conv2d_dgrad(x.data_ptr(), wd.data_ptr(), mask.data_ptr(), None, y.data_ptr(), None, C.byref(g), st)
    flop = 2.0 * N * H * W * Cc * Cc * KH * KW
    row = f'C={Cc:4d} {H:3d}x{W:<3d} k{KH}x{KW}:'
    for name, fn in (('fwd', fwd), ('dgrad', dgrad)):
        for _ in range(3):
            assert fn() == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1000
        tot[name] += us
        row += f'  {name} {us:7.1f} us {flop / us / 1e6:6.1f} TF/s'
    print(row)
print('sum fwd %.1f us, dgrad %.1f us' % (tot['fwd'], tot['dgrad']))
