"""Upper bound of 'pair the two encoders into grouped launches' (VERDICT r3 #1-i): per encoder stage, fwd + bwd of
  (A) two encoders at N = 32 on two streams (what the step does today) against
  (B) ONE encoder at N = 64 on one stream (the tile count / launch count a 2-problem grouped launch would have).
Same FLOPs; (B) has half the launches on the dependent chain and twice the tiles per launch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dynmm_amd import ops, engine
from dynmm_amd.nn.blocks import ResNetEncoder

torch.manual_seed(0)
dev = 'cuda'
SHAPES = {1: (64, 120, 160), 2: (64, 120, 160), 3: (128, 60, 80), 4: (256, 30, 40)}   # stage input (C, H, W)


def enc():
    e = ResNetEncoder('resnet34', 'NonBottleneck1D', 3).to(dev).train()
    for p in e.parameters():
        p.grad = torch.zeros_like(p)
    return e


er, ed = enc(), enc()
side = torch.cuda.Stream()


def run_pair(j, xr, xd):
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        yd = getattr(ed, f'forward_layer{j}')(xd)
    yr = getattr(er, f'forward_layer{j}')(xr)
    main.wait_stream(side)
    return yr, yd


def timeit(fn, n=6):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for j in (1, 2, 3, 4):
    C_, H, W = SHAPES[j]
    xr = torch.randn(32, C_, H, W, device=dev).relu_().requires_grad_(True)
    xd = torch.randn(32, C_, H, W, device=dev).relu_().requires_grad_(True)
    x64 = torch.randn(64, C_, H, W, device=dev).relu_().requires_grad_(True)
    res = {}
    for mode in ('fwd', 'fwd+bwd'):
        def a():
            with engine.direct_gradients(True):
                ops.begin_step()
                yr, yd = run_pair(j, xr, xd)
                if mode != 'fwd':
                    torch.autograd.backward([yr, yd], [torch.ones_like(yr), torch.ones_like(yd)])
                ops.join_async()

        def b():
            with engine.direct_gradients(True):
                ops.begin_step()
                y = getattr(er, f'forward_layer{j}')(x64)
                if mode != 'fwd':
                    torch.autograd.backward([y], [torch.ones_like(y)])
                ops.join_async()

        def c():   # two encoders at N = 32 one after the other on ONE stream (no overlap at all)
            with engine.direct_gradients(True):
                ops.begin_step()
                yr = getattr(er, f'forward_layer{j}')(xr)
                yd = getattr(ed, f'forward_layer{j}')(xd)
                if mode != 'fwd':
                    torch.autograd.backward([yr, yd], [torch.ones_like(yr), torch.ones_like(yd)])
                ops.join_async()
        res[mode] = (timeit(a), timeit(b), timeit(c))
    print(f'stage {j}: ' + ' | '.join(f'{m}: 2x32 two streams {v[0]:.2f} ms, 1x64 one stream {v[1]:.2f} ms, 2x32 one stream {v[2]:.2f} ms'
                                      for m, v in res.items()), flush=True)
