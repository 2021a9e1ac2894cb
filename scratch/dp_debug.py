"""Debug helper for tests/test_dp_gpu.py: two ranks on GPU 0 over gloo, prints how the reduced flat gradient
relates to the per-rank gradients, bucket by bucket, with and without overlap."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, port, overlap):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from dynmm_amd import dp, engine, ops, synth
    from dynmm_amd.nn.net import SkipGateESANet
    h, w, n = 96, 128, 3
    m = SkipGateESANet(height=h, width=w, encoder_block='NonBottleneck1D', fuse_depth_in_rgb_encoder='SE-add')
    synth.fill_state_dict(m.state_dict(), seed=0)
    m = m.cuda().train()
    m.temp, m.hard_gate = 1.0, False
    rgb, depth = synth.synth_inputs(n, h, w, seed=100 + rank, device='cuda')
    labels = [synth.synth_labels(n, h // s, w // s, seed=300 + s + rank, device='cuda').to(torch.uint8) for s in (1, 8, 16, 32)]
    step = engine.TrainStep(m, np.linspace(0.5, 2.0, 40), lr=0.01, loss_ratio=0.1, bucket_mb=8.0, overlap=overlap)
    red = step.reducer
    red.active = False                                           # (1) un-reduced pass: hooks ignored
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    step._body(rgb, depth, labels)
    torch.cuda.synchronize()
    local = red.flat.clone()
    step._body(rgb, depth, labels)
    torch.cuda.synchronize()
    local2 = red.flat.clone()
    m.load_state_dict(sd)
    red.active = True
    step._body(rgb, depth, labels)
    log = list(red.launch_log)
    red.finish()
    torch.cuda.synchronize()
    reduced = red.flat.clone()
    both = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(both, local)
    want = (both[0] + both[1]) * 0.5
    if rank == 0:
        print(f'overlap={overlap} buckets={len(red.buckets)} log={log}')
        print('  local repeat identical:', bool(torch.equal(local, local2)), 'n diff', int((local != local2).sum()))
        for b, (s, e) in enumerate(red.buckets):
            r, wv, l0, l1 = reduced[s:e], want[s:e], both[0][s:e], both[1][s:e]
            print(f'  bucket {b} [{s},{e}) |want| {wv.abs().max():.3e}  |red-want| {(r - wv).abs().max():.3e}  '
                  f'|red-l0| {(r - l0).abs().max():.3e} |red-0.5*l0| {(r - 0.5 * l0).abs().max():.3e} '
                  f'|red-(l0+l1)| {(r - (l0 + l1)).abs().max():.3e} |red| {r.abs().max():.3e}')
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    for overlap in (False, True):
        ctx = mp.get_context('spawn')
        port = 29500 + (1 if overlap else 0)
        ps = [ctx.Process(target=worker, args=(r, 2, port, overlap)) for r in range(2)]
        for p in ps:
            p.start()
        for p in ps:
            p.join()
