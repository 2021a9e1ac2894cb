"""Pure data-parallel gradient exchange for the hot path: one process per GPU, a full
replica per rank, ONE logical all-reduce(sum)/world of all gradients per step over RCCL/xGMI.

Design for MI355X: gradients live in a single flat fp32 buffer (129.6 MB for config P) carved into a
few large buckets; every parameter's `.grad` is a view into it, so autograd accumulates in place and
no flatten/unflatten copies exist.  xGMI is point-to-point, so a ring all-reduce is bound by one link
(~153 GB/s/dir): a few ~32 MB buckets keep each collective far above the latency floor while letting
the first buckets (decoder grads, produced first) fly under the rest of backward — as stream-ordered collectives
on one of the step's two weight-gradient streams, not on a stream of their own (the step holds four busy streams; a fifth
costs it 10 ms: GradBucketReducer.__init__).
BatchNorm statistics stay per replica (DDP semantics; the reference has no SyncBN).

Works with any torch.distributed backend: `nccl` (= RCCL) on GPUs, `gloo` on CPU for the
world_size-2 tests in tests/test_dp_gloo.py.
"""
import os

import torch
import torch.distributed as dist


class GradBucketReducer:
    EXCHANGE_MODES = ('wgrad', 'depth', 'comm')

    def __init__(self, params, bucket_mb=32.0, overlap=True, process_group=None, exchange='wgrad'):
        """exchange (RCCL only; other backends always use asynchronous collectives): where the bucket all-reduces are enqueued —
          'wgrad'  stream-ordered on the LAST weight-gradient stream of ops.stream_plan() (least priority; default),
          'depth'  stream-ordered on the depth-encoder stream of the plan (normal priority; idle during the decoder's backward,
                   on the dependent chain during the encoders'),
          'comm'   asynchronous collectives from a communication stream of their own (the classic DDP arrangement: the process
                   group's internal stream does the work) — a FIFTH and sixth busy stream, which costs a single rank 10 ms per
                   step on this part (profiles/r05_ab_runs.md), but the one arrangement whose behaviour at world > 1 is
                   common knowledge.
        None of the three has run at world > 1 on hardware (no multi-GPU node in any round); the choice is a constructor
        argument (engine.TrainStep(exchange=...), bench.py --dp-exchange) so the first real run can compare them."""
        if exchange not in self.EXCHANGE_MODES:
            raise ValueError(f'exchange must be one of {self.EXCHANGE_MODES}, got {exchange!r}')
        self.exchange = exchange
        self.params = [p for p in params if p.requires_grad]
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # tests: run the collectives of a 1-rank group too (exercises the RCCL path on a single-GPU box)
        self.force = dist.is_initialized() and os.environ.get('DYNMM_DP_FORCE_COLLECTIVES') is not None
        dev = self.params[0].device
        # buckets are filled in REVERSE parameter order: backward produces the last layers' grads first
        order = list(reversed(self.params))
        total = sum(p.numel() for p in order)
        # one extra element behind the gradients: the step's total loss rides in the LAST bucket's all-reduce, so every
        # rank takes the same skip / raise decision on a non-finite loss (a NaN on one rank is a NaN in the sum)
        self._store = torch.zeros(total + 1, device=dev, dtype=torch.float32)
        self.flat = self._store[:total]
        self.loss_slot = self._store[total:]
        cap = max(1, int(bucket_mb * (1 << 20) / 4))
        self.buckets = []          # (start, end) element ranges in self.flat
        self._bucket_of = {}
        off, bstart, bidx = 0, 0, 0
        for p in order:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            self._bucket_of[p] = bidx
            off += n
            if off - bstart >= cap:
                self.buckets.append((bstart, off))
                bstart, bidx = off, bidx + 1
        if off > bstart:
            self.buckets.append((bstart, off))
        self._total = total
        self._pending = [0] * len(self.buckets)
        self._count = [0] * len(self.buckets)
        for p in order:
            self._count[self._bucket_of[p]] += 1
        self._works = []
        self.overlap = overlap and (self.world > 1 or self.force)
        self._comm_stream = None        # created on first use (never for the stream-ordered modes: a torch pool stream would be a
                                        # stream outside ops.stream_plan())
        self._dev = dev
        # RCCL: the bucket all-reduces are issued as stream-ordered collectives (async_op=False: the process group enqueues them on
        # the CURRENT stream, no stream of its own) on one of the streams the step already uses.  A stream of
        # their own — the first design: `_comm_stream` + the process group's internal stream — makes a fifth / sixth busy stream,
        # which costs the step 10 ms on this part (profiles/r05_ab_runs.md: 74.0 against 63.2 ms with ONE rank's collectives
        # forced on; the same as a third weight-gradient stream or a stream for the gate).  Other backends (gloo: tests) keep the
        # asynchronous form.
        nccl = bool(dev.type == 'cuda' and dist.is_initialized() and dist.get_backend(process_group) == 'nccl')
        self._stream_ordered = nccl and exchange != 'comm'
        if self._stream_ordered:
            # "async_op=False runs on the current stream" is the behaviour of ProcessGroupNCCL since torch 2.8 (before, a
            # synchronous collective still ran on the group's internal stream and the current stream waited for it: correct
            # results, but the fifth stream is back).  Refuse silently-different behaviour.
            ver = tuple(int(x) for x in torch.__version__.split('+')[0].split('.')[:2])
            if ver < (2, 8):
                raise RuntimeError(f"GradBucketReducer(exchange={exchange!r}) needs torch >= 2.8 (stream-ordered collectives with "
                                   f"async_op=False); this is torch {torch.__version__}: pass exchange='comm'")
            from . import ops
            ops.stream_plan(dev)            # the plan's streams exist before any capture / first backward
        self._hooks = []
        self._streams = [dict() for _ in self.buckets]      # per bucket: producer streams seen this step
        self.active = True                                   # False: gradient hooks are ignored (tests: un-reduced pass)
        self.enabled = True                                  # False: no collective at all (bench: the step without its exchange)
        self._ready = [False] * len(self.buckets)
        self._next = 0                                       # buckets are launched strictly in index order (see _arrived)
        self.launched_in_backward = 0                        # buckets whose all-reduce started before finish()
        self.launch_log = []                                 # (bucket, 'backward' | 'finish') of the last step
        self.pending_scale = 1.0                             # 1/world still owed to the flat buffer after finish(average=False)
        if self.overlap:
            # (a) plain autograd accumulation (torch modules): post-accumulate hooks;
            # (b) the HIP kernels' in-place gradient protocol (ops.DIRECT_GRAD: autograd never sees these
            #     gradients, so hooks never fire): ops reports every parameter whose gradient kernels are enqueued.
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad_ready))
            from . import ops
            ops.GRAD_READY_HOOK = self._on_grad_written

    # -- step protocol --------------------------------------------------------------------------
    def zero(self):
        """Start of a step: clear the flat buffer (grads accumulate in place into their views)."""
        self._store.zero_()
        self._pending = list(self._count)
        self._ready = [False] * len(self.buckets)
        self._next = 0
        self._works = []
        self._streams = [dict() for _ in self.buckets]
        self.launched_in_backward = 0
        self.launch_log = []

    def _launch(self, b, where='finish'):
        s, e = self.buckets[b]
        if b == len(self.buckets) - 1:
            e = self._total + 1                          # + the loss slot
        chunk = self._store[s:e]
        self.launch_log.append((b, where))
        if self._stream_ordered:
            from . import ops
            cs = self._comm_stream = ops.exchange_stream() if self.exchange == 'wgrad' else ops.side_stream()
            cs.wait_stream(torch.cuda.current_stream())
            for st in self._streams[b].values():         # side streams that wrote into this bucket (wgrad queue, depth encoder)
                if st.cuda_stream != cs.cuda_stream:
                    cs.wait_stream(st)
            with torch.cuda.stream(cs):
                dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=False)
        elif self._dev.type == 'cuda':
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream(device=self._dev)
            self._comm_stream.wait_stream(torch.cuda.current_stream())
            for st in self._streams[b].values():         # side streams that wrote into this bucket (wgrad queue, depth encoder)
                self._comm_stream.wait_stream(st)
            with torch.cuda.stream(self._comm_stream):
                self._works.append(dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            self._works.append(dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _arrived(self, p):
        b = self._bucket_of.get(p)
        if b is None or self._pending[b] <= 0 or not self.active or not self.enabled:
            return
        self._pending[b] -= 1
        if self._pending[b] == 0:
            # Collectives of one communicator must be issued in the SAME order on every rank.  Completion order can
            # differ between ranks (hard-gate compaction: a rank whose shard skips a depth stage never touches those
            # parameters), so a complete bucket waits until every bucket before it has been launched.
            self._ready[b] = True
            while self._next < len(self.buckets) and self._ready[self._next]:
                self.launched_in_backward += 1
                self._launch(self._next, 'backward')
                self._next += 1

    def _on_grad_ready(self, p):
        # autograd runs a parameter's AccumulateGrad node — and this hook — even when the backward returned None for
        # it, i.e. also under the in-place gradient protocol; there the kernels report completion themselves (with
        # the streams they were launched on), so the hook must stay out of the way
        from . import ops
        if ops.DIRECT_GRAD:
            return
        self._arrived(p)

    def _on_grad_written(self, p, streams):
        b = self._bucket_of.get(p)
        if b is None:
            return
        for st in streams:
            self._streams[b][st.cuda_stream] = st
        self._arrived(p)

    def finish(self, average=True):
        """After backward: make sure every bucket is reduced, then average.  `average=False` leaves the SUM over ranks in
        the flat buffer (and the loss slot): engine.TrainStep folds 1/world into the fused optimizer kernel's
        `grad_scale` argument instead of spending a pass over the 130 MB buffer on it; `pending_scale` says what is owed."""
        self.pending_scale = 1.0
        if (self.world == 1 and not self.force) or not self.enabled:
            return
        first = self._next if self.overlap else 0
        for b in range(first, len(self.buckets)):       # the rest (incl. buckets with parameters that got no gradient), in order
            self._pending[b] = 0
            self._launch(b)
        self._next = len(self.buckets)
        for w in self._works:
            w.wait()
        if self._comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self._comm_stream)
        if average:
            self._store.mul_(1.0 / self.world)
        else:
            self.pending_scale = 1.0 / self.world
        self._works = []

    def set_loss(self, total):
        """Record this rank's total loss (device scalar) for the shared non-finite decision; call before backward."""
        if self.world > 1 or self.force:
            self.loss_slot.copy_(total.detach().reshape(1))

    def reduced_loss(self, local):
        """The loss every rank should base its skip decision on: the mean over ranks after finish(), else `local`."""
        return self.loss_slot if ((self.world > 1 or self.force) and self.enabled) else local

    def remove_hooks(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
        from . import ops
        if ops.GRAD_READY_HOOK == self._on_grad_written:
            ops.GRAD_READY_HOOK = None


def broadcast_parameters(module, src=0, process_group=None):
    """Make every replica start from rank `src`'s parameters and buffers."""
    if not dist.is_initialized() or dist.get_world_size(process_group) == 1:
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t, src=src, group=process_group)


def broadcast_buffers(module, src=0, process_group=None):
    """Rank `src`'s buffers (BatchNorm running statistics, step counters) on every replica — DDP's `broadcast_buffers`.
    Running statistics are updated per replica from its own shard (no SyncBN, like the reference), so before a SHARDED
    evaluation the replicas must agree on them, or the all-reduced confusion matrix describes no single model
    (engine.evaluate).  One collective per dtype: the buffers are flattened, broadcast and copied back."""
    if not dist.is_initialized() or dist.get_world_size(process_group) == 1:
        return 0
    by_dtype = {}
    for b in module.buffers():
        if b.numel():
            by_dtype.setdefault((b.dtype, b.device), []).append(b)
    with torch.no_grad():
        for bufs in by_dtype.values():
            flat = torch.cat([b.reshape(-1) for b in bufs])
            dist.broadcast(flat, src=src, group=process_group)
            off = 0
            for b in bufs:
                n = b.numel()
                b.copy_(flat[off:off + n].view_as(b))
                off += n
    return sum(len(v) for v in by_dtype.values())


def shard_batch(global_batch, rank, world):
    """Contiguous per-rank slice [lo, hi) of a global batch (used by train/eval drivers)."""
    per = global_batch // world
    rem = global_batch % world
    lo = rank * per + min(rank, rem)
    return lo, lo + per + (1 if rank < rem else 0)
