import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dynmm_amd import engine, synth, ops
from dynmm_amd.nn.net import SkipGateESANet
g = np.load('tests/golden/train_steps_P_se.npz')
h, w, n = [int(v) for v in g['meta']]
lr, wd, mom, ratio, budget, temp = [float(v) for v in g['hyper']]
res = {}
for use_graph in (False, True):
    m = SkipGateESANet(height=h, width=w, encoder_block='NonBottleneck1D', fuse_depth_in_rgb_encoder='SE-add')
    synth.fill_state_dict(m.state_dict(), 0)
    m = m.cuda().train(); m.temp = temp
    rgb, depth = synth.synth_inputs(n, h, w, seed=1234, device='cuda')
    labels = [synth.synth_labels(n, h // s, w // s, seed=300 + s, device='cuda') for s in (1, 8, 16, 32)]
    step = engine.TrainStep(m, g['cw'], lr=lr, momentum=mom, weight_decay=wd, loss_ratio=ratio, flop_budget=budget, use_graph=use_graph)
    for s in range(2):
        out = step(rgb, depth, labels)
        print(use_graph, s, out['losses'].cpu().numpy(), out['total'].item())
    res[use_graph] = {k: v.detach().double().cpu().clone() for k, v in m.state_dict().items()}
names = [str(k) for k in g['param_names']]
for k in names:
    a, b = res[False][k], res[True][k]
    d = (a - b).abs().max().item() / max(a.abs().max().item(), 1e-12)
    if d > 1e-4: print('eager vs graph', k, d, a.norm().item(), b.norm().item())
