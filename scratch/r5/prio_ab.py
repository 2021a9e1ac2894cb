"""A/B of HIP stream priorities for the step's streams (scratch experiment, not product):
   python scratch/r5/prio_ab.py <side_prio|x> <wgrad_prio|x> [bench.py args...]
side = the depth encoder's stream (nn/net.py encoder_stage_pair), wgrad = the weight-gradient pool (ops._WGRAD_POOL);
'x' = leave as is (default priority 0).  Lower number = higher priority."""
import os, sys, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from dynmm_amd import ops
from dynmm_amd.nn import net

side_p, wg_p = sys.argv[1], sys.argv[2]
_keep = []


def mk(prio):
    """torch only reaches priorities 0 (normal) and -1 (high); 1 (low) through the runtime + ExternalStream"""
    prio = int(prio)
    if prio <= 0:
        return torch.cuda.Stream(priority=prio)
    import ctypes
    hip = ctypes.CDLL('libamdhip64.so')
    torch.cuda.init()
    torch.zeros(1, device='cuda')
    lo, hi = ctypes.c_int(), ctypes.c_int()
    hip.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi))
    print('runtime priority range: least', lo.value, 'greatest', hi.value, file=sys.stderr)
    h = ctypes.c_void_p()
    rc = hip.hipStreamCreateWithPriority(ctypes.byref(h), 1, prio)        # hipStreamNonBlocking
    assert rc == 0, rc
    got = ctypes.c_int()
    hip.hipStreamGetPriority(h, ctypes.byref(got))
    print('created stream with priority', got.value, file=sys.stderr)
    s = torch.cuda.ExternalStream(h.value)
    _keep.append(h)
    return s


try:
    print('priority range', torch.cuda.Stream.priority_range(), file=sys.stderr)
except Exception as e:
    print('no priority_range:', e, file=sys.stderr)
if wg_p != 'x':
    ops._WGRAD_POOL[:] = [mk(wg_p) for _ in range(max(1, ops.WGRAD_STREAMS))]
if side_p != 'x':
    orig = net.encoder_stage_pair

    def pair(model, j, r_in, d_in):
        if getattr(model, '_side', None) is None:
            model._side = mk(side_p)
        return orig(model, j, r_in, d_in)
    net.encoder_stage_pair = pair
sys.argv = [os.path.join(ROOT, 'bench.py')] + sys.argv[3:]
runpy.run_path(sys.argv[0], run_name='__main__')
