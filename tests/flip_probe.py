"""Is the decoder-module failure of the Winograd training forward a ReLU tie?  Run the module twice (direct / Winograd forward),
hook every submodule output, and list where (out > 0) differs and how large the values are there; both against the fp64
oracle.  A diagnostic, not a test (pytest does not collect it): python tests/flip_probe.py [draws]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from dynmm_amd import ops, synth  # noqa: E402
from dynmm_amd.nn.decoder import DecoderModule  # noqa: E402
from oracle import dynmm_oracle as O  # noqa: E402


def rnd(*shape, seed=0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return torch.randn(*shape, generator=g)


for seed_off in range(int(sys.argv[1]) if len(sys.argv) > 1 else 1):
    m = DecoderModule(128, 128, 3, 40)
    synth.fill_state_dict(m.state_dict(), seed=3)
    x, skip = rnd(3, 128, 12, 16, seed=seed_off), rnd(3, 128, 24, 32, seed=5 + seed_off)
    sd = {f'm.{k}': (v.detach().clone().double() if v.dtype.is_floating_point else v.detach().clone()) for k, v in m.state_dict().items()}
    params = {k: v.requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and 'running_' not in k}
    xr, sr = x.double().requires_grad_(True), skip.double().requires_grad_(True)
    out_ref = O.decoder_module(sd, 'm', xr, sr, True, 3)
    outs_ref = [o for o in out_ref if o is not None]
    gs = [rnd(*o.shape, seed=11 + i) for i, o in enumerate(outs_ref)]
    torch.autograd.backward(outs_ref, [g.double() for g in gs])
    m = m.cuda().train()
    res = {}
    for mode in ('dgrad', 'all'):
        ops.WINO = mode
        acts = {}
        hooks = [mod.register_forward_hook(lambda mod, i, o, n=n: acts.__setitem__(n, o.detach().clone()) if torch.is_tensor(o) else None)
                 for n, mod in m.named_modules() if n]
        for p in m.parameters():
            p.grad = None
        a, b = x.clone().cuda().requires_grad_(True), skip.clone().cuda().requires_grad_(True)
        out = m(a, b)
        outs = [o for o in out if o is not None]
        torch.autograd.backward(outs, [g.cuda() for g in gs])
        for h in hooks:
            h.remove()
        res[mode] = (acts, {n: p.grad.clone() for n, p in m.named_parameters()}, [o.detach() for o in outs])
    A, B = res['dgrad'][0], res['all'][0]
    flips = 0
    for n in A:
        da, db = A[n], B[n]
        mism = (da > 0) != (db > 0)
        if mism.any():
            flips += int(mism.sum())
            print(f'  seed {seed_off} {n}: {int(mism.sum())} sign decisions differ of {da.numel()}; values there: direct '
                  f'{da[mism].abs().max().item():.2e} wino {db[mism].abs().max().item():.2e}; max |diff| anywhere {(da - db).abs().max().item():.2e}')
    worst = {}
    for mode in ('dgrad', 'all'):
        e = 0.0
        for n, g in res[mode][1].items():
            ref = params[f'm.{n}'].grad
            if ref.abs().max() < 1e-5 * max(v.grad.abs().max().item() for v in params.values()):
                continue
            e = max(e, ((g.double().cpu() - ref).abs().max() / ref.abs().max()).item())
        o = max(((a_.double().cpu() - b_).abs().max() / b_.abs().max()).item() for a_, b_ in zip(res[mode][2], outs_ref))
        worst[mode] = (o, e)
    print(f'seed {seed_off}: sign flips between the two forwards {flips}; vs fp64 oracle (output, worst param grad): direct '
          f'{worst["dgrad"][0]:.2e} {worst["dgrad"][1]:.2e} | wino {worst["all"][0]:.2e} {worst["all"][1]:.2e}', flush=True)
