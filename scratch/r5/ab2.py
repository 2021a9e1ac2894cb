"""A/B wrapper (scratch): python scratch/r5/ab2.py <nostem|stem> <wgrad prio x|1> [bench args]
nostem: dynmm_conv2d_stem_fwd_stats_supported answers 0 (the stems' BatchNorm statistics take their own bn_stats pass again)."""
import os, sys, runpy, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from dynmm_amd import ops, lib as L
mode, wg_p = sys.argv[1], sys.argv[2]
lib = L.load()
if mode == 'nostem':
    lib.dynmm_conv2d_stem_fwd_stats_supported = lambda g: 0
_keep = []
if wg_p != 'x':
    hip = ctypes.CDLL('libamdhip64.so')
    torch.zeros(1, device='cuda')
    pool = []
    for _ in range(max(1, ops.WGRAD_STREAMS)):
        h = ctypes.c_void_p()
        assert hip.hipStreamCreateWithPriority(ctypes.byref(h), 1, int(wg_p)) == 0
        _keep.append(h)
        pool.append(torch.cuda.ExternalStream(h.value))
    ops._WGRAD_POOL[:] = pool
sys.argv = [os.path.join(ROOT, 'bench.py')] + sys.argv[3:]
runpy.run_path(sys.argv[0], run_name='__main__')
