"""Round-5 tree (scratch/r6/r5tree, git archive of d691c1c + today's .so: no kernel changed): four models built one after the
other in ONE process, each timed on BASELINE configs[2].  Prints each model's depth-encoder stream handle."""
import os, sys, time, json
import numpy as np
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'r5tree'))
from dynmm_amd import engine, ops                    # noqa: E402
import bench                                         # noqa: E402
assert 'r5tree' in engine.__file__, engine.__file__
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device('cuda:0')
N, H, W = 32, 480, 640
rgb, depth, labels = bench.make_batch(N, H, W, dev, 1234)
cw = np.linspace(0.5, 2.0, 40)
rows = []
for i in range(4):
    m = bench.make_model('P', H, W, dev).train()
    m.temp, m.hard_gate = 1.0, False
    ts = engine.TrainStep(m, cw, lr=1e-4, momentum=0.9, weight_decay=1e-4, loss_ratio=1.0, flop_budget=0.0)
    for _ in range(3):
        ts(rgb, depth, labels)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ts(rgb, depth, labels)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    rows.append({'model': i, 'ms': round(ms, 2), 'side': hex(m._side.cuda_stream), 'wgrad': [hex(s.cuda_stream) for s in ops._WGRAD_POOL]})
    print(rows[-1], flush=True)
    ts.reducer.remove_hooks()
    del m, ts
    torch.cuda.empty_cache()
print(json.dumps({'tree': 'r5', 'rows': rows}))
