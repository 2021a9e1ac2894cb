"""CPU, world_size 2, gloo: the data-parallel gradient exchange (dynmm_amd/dp.py) that bench.py and
train.py use over RCCL on the GPUs.  Checks that bucketed all-reduce over the flat gradient buffer
equals the single-process gradient of the concatenated batch (linear loss => exact DP equivalence),
with and without backward overlap, and that parameter broadcast works."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model():
    torch.manual_seed(7)
    return torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(),
                               torch.nn.Conv2d(8, 8, 3, padding=1), torch.nn.ReLU(),
                               torch.nn.Conv2d(8, 4, 1))


def _worker(rank, world, port, overlap, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from dynmm_amd import dp
    m = _model()
    if rank == 1:                      # de-synchronise, then broadcast must repair it
        with torch.no_grad():
            for p in m.parameters():
                p.add_(1.0)
    dp.broadcast_parameters(m)
    red = dp.GradBucketReducer(m.parameters(), bucket_mb=0.001, overlap=overlap)   # tiny buckets -> several
    assert len(red.buckets) > 1
    torch.manual_seed(100)
    x = torch.randn(4, 3, 8, 8)
    lo, hi = dp.shard_batch(4, rank, world)
    for it in range(2):                # two steps: zero() must reset state
        red.zero()
        (m(x[lo:hi]).sum() / 4.0).backward()     # mean over the GLOBAL batch, per-rank share
        if it == 0:
            red.finish()
            assert red.pending_scale == 1.0
        else:
            # what engine.TrainStep does: keep the SUM, hand 1/world to the optimizer kernel's grad_scale
            red.finish(average=False)
            assert red.pending_scale == 1.0 / world
            red.flat.mul_(red.pending_scale)
    # numpy payloads: a torch tensor on an mp.Queue travels as a file descriptor the parent must fetch from a
    # still-living child; numpy arrays are pickled by value, so the worker may exit right after put()
    q.put((rank, [p.grad.numpy().copy() for p in m.parameters()], [p.detach().numpy().copy() for p in m.parameters()],
           red.launched_in_backward, len(red.buckets)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('overlap', [False, True])
def test_bucketed_allreduce_matches_single_process(overlap):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, overlap, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = _model()
    torch.manual_seed(100)
    x = torch.randn(4, 3, 8, 8)
    (ref(x).sum() / 4.0).backward()
    for rank, grads, params, in_bwd, nb in res:
        for g, p, pr in zip(grads, params, ref.parameters()):
            assert np.allclose(p, pr.detach().numpy())                  # broadcast restored rank 1
            # all_reduce(sum)/world of per-rank grads of (sum over shard)/4  ==  grad of global mean / world ... x world
            assert np.allclose(g * 2.0, pr.grad.numpy(), atol=1e-6), rank
        # with overlap every bucket's all-reduce is launched from a gradient hook DURING backward
        assert in_bwd == (nb if overlap else 0), (in_bwd, nb)
    assert all(np.array_equal(a, b) for a, b in zip(res[0][1], res[1][1]))   # replicas agree bit-for-bit


def _worker_uneven(rank, world, port, q):
    """rank 1 never uses the middle convolution (a rank whose shard skips a depth stage under gate-decision
    compaction): its buckets complete in a different order than rank 0's — launches must still pair up."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from dynmm_amd import dp
    m = _model()
    red = dp.GradBucketReducer(m.parameters(), bucket_mb=0.00001, overlap=True)      # one bucket per parameter
    assert len(red.buckets) == 6
    torch.manual_seed(100)
    x = torch.randn(4, 3, 8, 8)
    lo, hi = dp.shard_batch(4, rank, world)
    for _ in range(2):
        red.zero()
        h = torch.relu(m[0](x[lo:hi]))
        if rank == 0:
            h = torch.relu(m[2](h))
        loss = m[4](h).sum() / 4.0
        red.set_loss(loss if rank == 0 else loss * float('nan'))     # one rank sees a non-finite loss
        loss.backward()
        red.finish()
    q.put((rank, [p.grad.numpy().copy() for p in m.parameters()], list(red.launch_log),
           float(red.reduced_loss(loss.detach()).item())))
    dist.barrier()
    dist.destroy_process_group()


def test_uneven_touch_sets_keep_bucket_order_and_share_the_nan():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_uneven, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.manual_seed(100)
    x = torch.randn(4, 3, 8, 8)
    want = None
    for rank in range(2):
        ref = _model()
        h = torch.relu(ref[0](x[2 * rank:2 * rank + 2]))
        if rank == 0:
            h = torch.relu(ref[2](h))
        (ref[4](h).sum() / 4.0).backward()
        g = [(p.grad if p.grad is not None else torch.zeros_like(p)).numpy() for p in ref.parameters()]
        want = g if want is None else [a + b for a, b in zip(want, g)]
    for rank, grads, log, shared_loss in res:
        assert [b for b, _ in log] == sorted(b for b, _ in log), log        # strictly ascending bucket order on every rank
        for g, w in zip(grads, want):
            assert np.allclose(g * 2.0, w, atol=1e-6), rank
        assert np.isnan(shared_loss)                                         # BOTH ranks see the non-finite loss
    assert all(np.array_equal(a, b) for a, b in zip(res[0][1], res[1][1]))


def test_shard_batch_partitions():
    from dynmm_amd import dp
    for n, w in ((256, 8), (10, 4), (3, 8)):
        spans = [dp.shard_batch(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def _worker_buffers(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from dynmm_amd import dp
    torch.manual_seed(3)
    m = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 1), torch.nn.BatchNorm2d(4), torch.nn.BatchNorm2d(4))
    m.train()
    torch.manual_seed(50 + rank)                     # each replica updates its running statistics from its own shard
    for _ in range(3):
        m(torch.randn(2, 3, 5, 5) * (1 + rank))
    if rank == 1:
        m[2].num_batches_tracked.add_(5)
    before = [b.detach().clone() for b in m.buffers()]
    n = dp.broadcast_buffers(m, 0)
    after = [b.detach().numpy().copy() for b in m.buffers()]
    q.put((rank, n, [b.numpy().copy() for b in before], after, [p.detach().numpy().copy() for p in m.parameters()]))
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_buffers_makes_replicas_agree_on_running_statistics():
    """engine.evaluate shards the validation batches over the ranks: every rank must then evaluate rank 0's BatchNorm
    running statistics (the replicas' own differ), int64 step counters included; parameters are not touched."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_buffers, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, n0, before0, after0, _), (_, n1, before1, after1, _) = res
    assert n0 == n1 == 6                                            # 2 x (running_mean, running_var, num_batches_tracked)
    assert any(not np.array_equal(a, b) for a, b in zip(before0, before1))     # the replicas really had diverged
    for b0, a0, a1 in zip(before0, after0, after1):
        assert np.array_equal(a0, b0) and np.array_equal(a1, b0) and a1.dtype == b0.dtype      # rank 0's values, bit for bit
