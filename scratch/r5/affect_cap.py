"""ModalityDynMM branch streams: all branches on streams of their own (5 busy streams) against a cap of `cap` side streams
(the rest of the branches follow on the main stream).  python scratch/r5/affect_cap.py <cap|all>"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dynmm_amd.nn import affect as A
cap = sys.argv[1]
if cap != 'all':
    cap = int(cap)
    orig = A.run_branches

    def capped(fns):
        if len(fns) - 1 <= cap:
            return orig(fns)
        # the first `cap` + 1 branches in parallel, the remaining ones after them on the main stream
        head = orig(fns[:cap + 1])
        return head + [f() for f in fns[cap + 1:]]
    A.run_branches = capped
import bench
print(cap, json.dumps(bench.measure_affect(torch.device('cuda', 0), 20))[:400])
