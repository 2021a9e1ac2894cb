"""The callers either side of the hot path, on the HIP path:

  TrainStep  = FusionDynMM/train.py:289-335 — zero grads, forward, weighted multi-scale CE,
               total = sum(CE_s) + ratio*max(0, flop_loss - budget), backward, [DP all-reduce],
               optimizer step (SGD-Nesterov or Adam, train.py:554-579), NaN guard — as ONE
               hipGraph-capturable sequence: trainable parameters, gradients and optimizer state live in
               flat fp32 buffers (every nn.Parameter / .grad is a view); learning rate, momentum / beta1
               and the step counter are DEVICE scalars, so a captured step replays with no host work.
  evaluate   = FusionDynMM/eval.py:104-146 — forward(test=True), bilinear resize to the label size,
               arg-max, void mask, confusion matrix, mIoU*100.
"""
import contextlib
import ctypes as C

import torch

from . import dp, ops
from . import lib as L


class FlatParameters:
    """Re-home the given parameters into one contiguous fp32 buffer (views keep state_dict,
    load_state_dict and autograd working unchanged).  Layout = reverse parameter order; with align = 1 (the default) it is
    element for element the layout of dp.GradBucketReducer's flat gradient buffer.  align > 1 (elements): every parameter
    starts on a multiple of `align` (zero padding between them; kernels that stage weights with 16-byte loads —
    csrc/seq_ffn.hip — need align = 4): such a buffer does NOT match the reducer's packed layout and needs a gradient buffer
    laid out with the same spans (AffectTrainStep allocates its own; _FlatOptimizer asserts equal sizes)."""

    def __init__(self, params, align=1):
        self.params = list(params)
        dev = self.params[0].device
        align = max(1, int(align))
        total = sum(-(-p.numel() // align) * align for p in self.params)
        self.flat = torch.zeros(total, device=dev, dtype=torch.float32)
        self.span = {}                      # id(param) -> (lo, hi) element range in the flat buffers
        self.pspan = {}                     # the same with hi rounded up to the next parameter's start (what an optimizer walks)
        off = 0
        with torch.no_grad():
            for p in reversed(self.params):
                n = p.numel()
                self.flat[off:off + n].copy_(p.detach().reshape(-1))
                p.data = self.flat[off:off + n].view_as(p)
                self.span[id(p)] = (off, off + n)
                off += -(-n // align) * align
                self.pspan[id(p)] = (self.span[id(p)][0], off)


    def refresh(self):
        """Assert that every parameter still is a view of the flat buffer (load_state_dict copies into the views; code
        that REPLACES `.data` would silently detach a parameter from the fused optimizer)."""
        base = self.flat.data_ptr()
        for q in self.params:
            lo, _ = self.span[id(q)]
            if q.data_ptr() != base + 4 * lo:
                raise RuntimeError('a parameter was re-homed outside the flat buffer; rebuild TrainStep after replacing parameters')


def _merge(spans):
    out = []
    for lo, hi in sorted(spans):
        if out and lo <= out[-1][1]:
            out[-1][1] = max(out[-1][1], hi)
        else:
            out.append([lo, hi])
    return [(a, b) for a, b in out]


class _FlatOptimizer:
    """Common part of the fused flat optimizers.

    * torch.optim semantics for parameters that took no part in a step (`.grad is None` there): they are
      skipped — no weight decay, no momentum update.  The kernels that write gradients report the
      parameters they touched (ops.touched()); `step(touched)` updates only those element ranges.
    * `groups`: {name: [params]} partitions the parameters into sets that may start receiving gradients
      at different times (the gate vs everything else: ini_stage / baseline epochs, --freeze); Adam keeps
      one device step counter per group so its bias correction matches torch's per-parameter counters.
    * lr, momentum/beta1 are device scalars (OneCycleLR changes both every epoch, train.py:119-128).
    * NaN guard: the total loss of the step is inspected on the device; a non-finite loss skips the update
      and latches `nan_step` (train.py:334-335 raises on the host every step; here the host looks once
      per epoch or whenever it wants with check_finite())."""

    HYPER = 4

    def __init__(self, flatp, flat_grads, groups=None):
        assert flatp.flat.numel() == flat_grads.numel()
        self.fp, self.p, self.g = flatp, flatp.flat, flat_grads
        dev = self.p.device
        self.hyper = torch.zeros(self.HYPER, device=dev, dtype=torch.float32)
        self.nan_flag = torch.zeros(1, device=dev, dtype=torch.int32)
        groups = groups or {'all': list(flatp.params)}
        self.group_of = {}
        self.group_names = list(groups)
        for gi, name in enumerate(self.group_names):
            for q in groups[name]:
                self.group_of[id(q)] = gi
        self.steps = torch.zeros(max(1, len(self.group_names)), device=dev, dtype=torch.int32)
        self._all = [id(q) for q in flatp.params]

    # -- host-visible knobs (device scalars: visible to an already-captured graph) ------------------
    def set_lr(self, lr):
        self.hyper[0:1].fill_(float(lr))

    def set_momentum(self, m):
        self.hyper[1:2].fill_(float(m))

    def plan(self, touched=None):
        """[(group index, [(lo, hi), ...])] for the touched parameter ids (None = every parameter)."""
        ids = self._all if touched is None else [i for i in self._all if i in touched]
        per = {}
        for i in ids:
            per.setdefault(self.group_of[i], []).append(self.fp.pspan[i])
        return [(gi, _merge(sp)) for gi, sp in sorted(per.items())]

    grad_scale = 1.0           # multiplies every gradient inside the update kernel (data parallel: 1/world of the SUM)

    def step(self, touched=None, loss=None):
        lib = L.load()
        ops.note_mutation()                 # parameters are rewritten through raw pointers
        st = torch.cuda.current_stream().cuda_stream
        lp = None if loss is None else loss.data_ptr()
        for gi, ranges in self.plan(touched):
            sp = self.steps.data_ptr() + 4 * gi
            L.check(lib.dynmm_opt_tick(sp, st), 'opt_tick')
            for lo, hi in ranges:
                self._launch(lib, lo, hi, sp, lp, st)

    # -- checkpoints: torch.optim's own state_dict layout (train.py:131-135 / src/utils.py:118-175), so that a run can be
    # resumed by either implementation.  Parameter indices follow model.parameters() (what train.py hands to torch.optim);
    # TrainStep sets `param_index` / `n_params`; per-parameter tensors are views of the flat state buffers.
    param_index = None
    n_params = None
    STATE_KEYS = ()

    def _index(self):
        if self.param_index is not None:
            return self.param_index, self.n_params
        return {id(q): i for i, q in enumerate(self.fp.params)}, len(self.fp.params)

    def _group_dict(self, n):
        raise NotImplementedError

    def state_dict(self):
        index, n = self._index()
        steps = self.steps.tolist()
        state = {}
        for q in self.fp.params:
            if steps[self.group_of[id(q)]] == 0:
                continue                                   # torch.optim keeps no state for a parameter it never stepped
            lo, hi = self.fp.span[id(q)]
            ent = {k: buf[lo:hi].view_as(q).clone() for k, buf in zip(self.STATE_KEYS, self.state_tensors()[:len(self.STATE_KEYS)])}
            if 'exp_avg' in ent:
                ent['step'] = torch.tensor(float(steps[self.group_of[id(q)]]))
            state[index[id(q)]] = ent
        return {'state': state, 'param_groups': [self._group_dict(n)],
                'dynmm': {'steps': self.steps.clone(), 'hyper': self.hyper.clone(), 'groups': list(self.group_names)}}

    def load_state_dict(self, sd):
        index, _ = self._index()
        if 'param_groups' not in sd:                       # rounds 1-2 layout: the flat buffers themselves
            for key, buf in zip(self.STATE_KEYS, self.state_tensors()[:len(self.STATE_KEYS)]):
                if key in sd:
                    buf.copy_(sd[key])
            if 'steps' in sd:
                self.steps.copy_(sd['steps'])
            return
        state = sd['state']
        seen = {}
        with torch.no_grad():
            for q in self.fp.params:
                ent = state.get(index[id(q)])
                if ent is None:
                    continue
                lo, hi = self.fp.span[id(q)]
                for k, buf in zip(self.STATE_KEYS, self.state_tensors()[:len(self.STATE_KEYS)]):
                    if ent.get(k) is not None:
                        buf[lo:hi].copy_(ent[k].reshape(-1).to(buf.device))
                seen[self.group_of[id(q)]] = int(float(ent['step'])) if 'step' in ent else 1
        extra = sd.get('dynmm')
        if extra is not None and list(extra.get('groups', [])) == list(self.group_names):
            self.steps.copy_(extra['steps'].to(self.steps.device))
        else:                                              # a torch.optim checkpoint: per-parameter step counters
            for gi, n_ in seen.items():
                self.steps[gi:gi + 1].fill_(n_)
        grp = sd['param_groups'][0]
        if 'lr' in grp:
            self.set_lr(grp['lr'])

    def check_finite(self):
        """Raise the reference's error if any step since the last call saw a non-finite loss (one D2H read)."""
        v = int(self.nan_flag.item())
        if v:
            self.nan_flag.zero_()
            raise ValueError(f'Loss is None (non-finite total loss at optimizer step {v})')


class SGDNesterov(_FlatOptimizer):
    """torch.optim.SGD(nesterov=True, weight_decay=L2) — train.py:557-563 — as one kernel per touched range."""

    def __init__(self, flatp, flat_grads, lr, momentum=0.9, weight_decay=1e-4, groups=None):
        super().__init__(flatp, flat_grads, groups)
        self.buf = torch.zeros_like(self.p)
        self.weight_decay = float(weight_decay)
        self.set_lr(lr)
        self.set_momentum(momentum)

    def _launch(self, lib, lo, hi, sp, lp, st):
        L.check(lib.dynmm_sgd_nesterov(self.p.data_ptr(), self.g.data_ptr(), self.buf.data_ptr(),
                                       C.c_size_t(lo), C.c_size_t(hi), self.hyper.data_ptr(), self.weight_decay,
                                       float(self.grad_scale), lp, self.nan_flag.data_ptr(), sp, st), 'sgd_nesterov')

    STATE_KEYS = ('momentum_buffer',)

    def _group_dict(self, n):
        h = self.hyper.tolist()
        return {'lr': h[0], 'momentum': h[1], 'dampening': 0, 'weight_decay': self.weight_decay, 'nesterov': True,
                'maximize': False, 'foreach': None, 'differentiable': False, 'fused': None, 'params': list(range(n))}

    def load_state_dict(self, sd):
        super().load_state_dict(sd)
        if 'param_groups' in sd and 'momentum' in sd['param_groups'][0]:
            self.set_momentum(sd['param_groups'][0]['momentum'])

    def state_tensors(self):
        return [self.buf, self.steps]


class Adam(_FlatOptimizer):
    """torch.optim.Adam(betas=(0.9, 0.999), eps=1e-8, weight_decay=L2) — train.py:564-570."""

    def __init__(self, flatp, flat_grads, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4, groups=None,
                 decoupled=False):
        super().__init__(flatp, flat_grads, groups)
        self.decoupled = bool(decoupled)          # True: torch.optim.AdamW
        self.grad_scale_dev = None                # optional device scalar (gradient-norm clip coefficient)
        self.m, self.v = torch.zeros_like(self.p), torch.zeros_like(self.p)
        self.weight_decay = float(weight_decay)
        self.set_lr(lr)
        self.set_momentum(betas[0])
        self.hyper[2:3].fill_(float(betas[1]))
        self.hyper[3:4].fill_(float(eps))

    def _launch(self, lib, lo, hi, sp, lp, st):
        L.check(lib.dynmm_adam(self.p.data_ptr(), self.g.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                               C.c_size_t(lo), C.c_size_t(hi), self.hyper.data_ptr(), sp, self.weight_decay, float(self.grad_scale),
                               lp, self.nan_flag.data_ptr(), int(self.decoupled),
                               None if self.grad_scale_dev is None else self.grad_scale_dev.data_ptr(), st), 'adam')

    STATE_KEYS = ('exp_avg', 'exp_avg_sq')

    def _group_dict(self, n):
        h = self.hyper.tolist()
        return {'lr': h[0], 'betas': (h[1], h[2]), 'eps': h[3], 'weight_decay': self.weight_decay, 'amsgrad': False,
                'maximize': False, 'foreach': None, 'capturable': False, 'differentiable': False, 'fused': None,
                'params': list(range(n))}

    def state_tensors(self):
        return [self.m, self.v, self.steps]


@contextlib.contextmanager
def direct_gradients(async_wgrad):
    """Scope of the in-place gradient protocol (ops.DIRECT_GRAD / ops.ASYNC_WGRAD): inside, backward kernels
    OVERWRITE the `.grad` views (one backward per zero()) and return None to autograd; outside, every other
    backward in the process keeps torch's accumulate semantics."""
    saved = (ops.DIRECT_GRAD, ops.ASYNC_WGRAD)
    ops.DIRECT_GRAD, ops.ASYNC_WGRAD = True, bool(async_wgrad)
    try:
        yield
    finally:
        try:
            ops.flush_wgrad_groups()         # nothing queued may outlive the protocol's scope
        finally:
            ops.DIRECT_GRAD, ops.ASYNC_WGRAD = saved


class TrainStep:
    """One optimisation step of train.py's hot loop.  Build it AFTER model.freeze() when --freeze is used:
    only parameters with requires_grad take part (flat buffers, reducer buckets and optimizer state cover
    exactly those, as torch.optim skips parameters without gradients)."""

    MAX_GRAPHS = 4             # captures kept alive at once (train / ragged last batch / hard vs soft gates)

    def __init__(self, model, class_weight, lr, momentum=0.9, weight_decay=1e-4, loss_ratio=0.0,
                 flop_budget=0.0, use_graph=False, bucket_mb=32.0, multi_stream=True, optimizer='SGD',
                 overlap=True, fuse_tail=None, prepack=True, exchange='wgrad'):
        self.model = model
        ops.stream_plan(next(model.parameters()).device)      # the step's side streams exist before any capture
        self.cw = torch.as_tensor(class_weight, dtype=torch.float32, device=next(model.parameters()).device)
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        if not named:
            raise ValueError('TrainStep: the model has no trainable parameter')
        trainable = [p for _, p in named]
        self.flatp = FlatParameters(trainable)
        self.reducer = dp.GradBucketReducer(trainable, bucket_mb=bucket_mb, overlap=overlap and not use_graph, exchange=exchange)
        groups = {'gate': [p for n, p in named if 'gate' in n], 'rest': [p for n, p in named if 'gate' not in n]}
        groups = {k: v for k, v in groups.items() if v}
        if optimizer == 'SGD':
            self.opt = SGDNesterov(self.flatp, self.reducer.flat, lr, momentum, weight_decay, groups)
        elif optimizer == 'Adam':
            self.opt = Adam(self.flatp, self.reducer.flat, lr, weight_decay=weight_decay, groups=groups)
        else:
            raise NotImplementedError(f'Currently only SGD and Adam as optimizers are supported. Got {optimizer}')
        allp = list(model.parameters())                 # torch.optim's parameter numbering (train.py:557-570)
        self.opt.param_index, self.opt.n_params = {id(q): i for i, q in enumerate(allp)}, len(allp)
        # 3-stream schedule: RGB encoder | depth encoder | conv weight gradients (see nn/net.py, ops.py)
        self.multi_stream = bool(multi_stream)
        # None: on unless DYNMM_NO_FUSED_TAIL is set (A/B switch for bench.py / tests)
        self.fuse_tail = True if fuse_tail is None else bool(fuse_tail)
        self.prepack = ops.PackedWeights() if prepack else None
        if hasattr(model, 'dual_stream'):
            model.dual_stream = self.multi_stream
        self.loss_ratio, self.flop_budget = float(loss_ratio), float(flop_budget)
        self.use_graph = use_graph
        self._graphs = {}      # key -> (graph, static inputs, static outputs, touched set, gate temperature)
        self.last = None       # dict of device tensors: losses[4], loss_flop, total
        self.census = None     # {'streams': n, 'roles': [...]} of the last eager step / capture (ops.stream_census)
        self._touched = None

    # ------------------------------------------------------------------------------------------------
    def _body(self, rgb, depth, targets):
        dec = getattr(self.model, 'decoder', None) if self.fuse_tail else None
        with direct_gradients(self.multi_stream), self._prepacked():
            ops.touched_reset()
            ops.stream_census_reset()
            self.reducer.zero()
            if dec is not None:
                dec.defer_tail = True       # last up-sampling + full-resolution CE as one kernel pair (csrc/tail.hip)
            try:
                res = self.model(rgb, depth)
            finally:
                if dec is not None:
                    dec.defer_tail = False
            if len(res) == 2 and isinstance(res[0], (tuple, list)):
                outs, lf = res                                   # SkipGateESANet: ((out, out8, out16, out32), flop loss)
            else:
                outs, lf = res, torch.zeros((), device=rgb.device)   # SkipESANet: the four outputs only
            # weighted 4-scale CE, total-loss rule and the seeds of the backward pass on the device (no PyTorch
            # arithmetic kernels between the forward and the backward of the model)
            red = self.reducer
            shared = (red.world > 1 or red.force) and red.enabled
            # data parallel: the total is written into the slot that rides in the last gradient bucket, so after
            # finish() it holds the mean over ranks and every rank takes the same non-finite-loss decision
            self.last = ops.multi_scale_loss_backward(outs, targets, self.cw, lf if self.loss_ratio > 0 else None,
                                                      self.loss_ratio, self.flop_budget,
                                                      total_out=red.loss_slot if shared else None)
            self.last['loss_flop'] = lf.detach()
            ops.join_async()
            self._touched = ops.touched_ids()
            self.census = ops.stream_census(check=True)     # raises on a fifth busy stream (ops.MAX_BUSY_STREAMS)

    @contextlib.contextmanager
    def _prepacked(self):
        """all conv weights of the step packed by one launch (ops.PackedWeights); per-conv packs on the first step"""
        prev = ops.PREPACK
        ops.PREPACK = self.prepack
        if self.prepack is not None:
            self.prepack.pack()
        try:
            yield
        finally:
            if self.prepack is not None:
                self.prepack.invalidate()
            ops.PREPACK = prev

    def _rank_dependent_touch(self):
        """Can the set of parameters that received a gradient differ between ranks?  Dense execution launches the
        same kernels whatever the data; gate-decision compaction (a rank whose shard holds no sample for a depth
        stage skips that stage's kernels) and the host-drawn ini_stage branches do not."""
        m = self.model
        return bool(getattr(m, 'compact_train', False) or getattr(m, 'ini_stage', False))

    def _finish(self):
        red = self.reducer
        # the all-reduce leaves the SUM over ranks in the flat buffer; 1/world is applied inside the optimizer kernel
        # (its grad_scale argument) instead of by a separate pass over the 130 MB of gradients
        red.finish(average=False)
        self.opt.grad_scale = red.pending_scale
        touched = self._touched
        if red.world > 1 and red.enabled and touched is not None and self._rank_dependent_touch():
            # the all-reduced gradient of a parameter is non-zero on EVERY rank as soon as one rank touched it: update the
            # union, or the replicas drift apart (one MAX all-reduce of a per-parameter mask; these modes already pay a
            # host read per forward for the stage counts)
            ids = self.opt._all
            mask = torch.tensor([1 if i in touched else 0 for i in ids], dtype=torch.int32, device=red.flat.device)
            dp.dist.all_reduce(mask, op=dp.dist.ReduceOp.MAX, group=red.group)
            touched = {i for i, v in zip(ids, mask.tolist()) if v}
        self.opt.step(touched, red.reduced_loss(self.last['total']))
        if self.last['total'].data_ptr() == red.loss_slot.data_ptr():
            # data parallel: `total` was written into the reducer's loss slot (mean over ranks after finish()); the slot
            # is cleared by the next step's zero(), so callers that keep per-step losses (train.py appends them) get a copy
            self.last['total'] = red.loss_slot * red.pending_scale          # a new tensor: the mean over ranks

    def _graph_key(self, rgb, depth, targets):
        m = self.model
        return (tuple(rgb.shape), tuple(depth.shape), tuple(tuple(t.shape) for t in targets), bool(m.training),
                bool(getattr(m, 'hard_gate', False)), bool(getattr(m, 'baseline', False)),
                tuple(getattr(m, 'block_rule', ()) or ()))

    def __call__(self, rgb, depth, targets):
        """targets: list of 4 label maps (0 = void), uint8/float/int, at scales 1, 1/8, 1/16, 1/32."""
        targets = [t if t.dtype == torch.uint8 else t.to(torch.uint8) for t in targets]
        # ini_stage draws its branches with the host RNG on every call (…globalgate.py:267-270): a captured
        # graph would freeze one draw, so those steps always run eagerly.
        if not self.use_graph or getattr(self.model, 'ini_stage', False):
            self._body(rgb, depth, targets)
            self._finish()
            return self.last
        # hard_gate / baseline / block_rule reach the kernels as host branches: a capture is valid for one combination
        # of them (and of the input shapes) only, so captures are keyed on it.  The gate temperature is a by-value
        # kernel argument that train.py changes EVERY epoch (ExpDecayTemp): it is not part of the key — one capture
        # per key is kept; when the temperature has moved the old capture is reset (its private memory pool goes back
        # to the allocator) before the new one is taken, so a long run holds a handful of step-sized pools, not one
        # per epoch.
        key = self._graph_key(rgb, depth, targets)
        temp = float(getattr(self.model, 'temp', 0.0))
        entry = self._graphs.get(key)
        if entry is not None and entry[4] != temp:
            entry[0].reset()
            entry = None
        if entry is None:
            if len(self._graphs) >= self.MAX_GRAPHS and key not in self._graphs:
                old = next(iter(self._graphs))             # oldest capture (dict order = insertion order)
                self._graphs.pop(old)[0].reset()
            entry = self._capture(rgb, depth, targets) + (temp,)
            self._graphs[key] = entry
        graph, (s_rgb, s_depth, s_t), s_last, touched, _ = entry
        s_rgb.copy_(rgb)
        s_depth.copy_(depth)
        for a, b in zip(s_t, targets):
            a.copy_(b)
        graph.replay()
        ops.note_mutation()                 # the replayed step updated running statistics / parameters
        self._touched = touched
        self.last = {k: v.clone() for k, v in s_last.items()}     # the static tensors are overwritten by the next replay
        if self.reducer.world > 1:
            self._finish()
        return self.last

    def _capture(self, rgb, depth, targets):
        static = (rgb.clone(), depth.clone(), [t.clone() for t in targets])
        # snapshot BEFORE the side stream forks, so the warm-up cannot race the clones
        sd = {k: v.clone() for k, v in self.model.state_dict().items()}
        opt_state = [t.clone() for t in self.opt.state_tensors()]
        # warm-up outside capture (allocator, lazy init, weight registration) on the CALLER's stream: a stream created for it
        # would be one outside ops.stream_plan()
        self._body(*static)

        def restore():
            self.model.load_state_dict(sd)       # undo the BN running-stat updates of warm-up / capture
            for t, c in zip(self.opt.state_tensors(), opt_state):
                t.copy_(c)
        restore()
        if self.prepack is not None and self.prepack.reg and self.prepack.dirty:
            self.prepack._layout()           # the warm-up registered the weights: lay the arena out before capturing
        graph = torch.cuda.CUDAGraph()
        with ops.capture_scope(), torch.cuda.graph(graph):
            self._body(*static)
            if self.reducer.world == 1:
                self.opt.step(self._touched, self.last['total'])
        restore()
        return graph, static, self.last, self._touched


class InferStep:
    """The inference forward (eval.py:107-115; …globalgate.py:255-322 with `test=True`) as hipGraph replays — BASELINE configs[1]
    enqueues ~400 launches for 10 ms of kernels, so the eager forward runs at the speed of the host's Python (1160 … 1590 img/s
    on the boxes of round 5); a replay costs one launch.

    `step(rgb, depth, return_weight=False)` returns what `model(rgb, depth, test=True[, return_weight=True])` returns under
    `torch.no_grad()` in eval mode — bit-identical (tests/test_engine.py) — as STATIC tensors: valid until the next call.

    Two graphs per forward of a gated model (nn/net.py forward_front / forward_back): the front (stems, stem fusion, gate head,
    device side of the compaction decision) and, after the ONE 16-byte host read a compacted hard-gate forward needs
    (`stage_counts`; none when the host knows the branches: `baseline`, `branch_override`), the back for exactly those stage
    counts.  Back graphs are captured on the `capture_after`-th sighting of a count tuple (an eager `forward_back` serves the
    others) and kept in an LRU of MAX_GRAPHS.  Everything that feeds a capture is part of its key: input shapes, the gate flags
    and temperature, `compact`, `dual_stream`, the injected branches — and a stamp over the version counters of every parameter
    and buffer plus ops' mutation generation, because the folded BatchNorm factors / packed filters a capture reads are the
    cached ones (ops.conv2d_fused_eval) and must die with the weights they were made from.
    Eager fallbacks: `ini_stage` without injected branches (drawn with the host RNG per call), a training-mode model, models without a
    front / back split that take host decisions (SkipESANet's per-stage gates); a model without gates (the static ESANet) is
    one graph."""

    MAX_GRAPHS = 8

    def __init__(self, model, capture_after=2, policy='replay'):
        """policy: 'replay' — replay wherever a capture applies; 'auto' — per key, once both graphs of a forward whose stage counts
        the host knows exist, 3 replays are timed against 3 eager forwards (events, one-off) and the faster way is kept: a replay
        keeps kernels and dependencies but not streams (the depth encoder no longer runs beside the RGB one), so on a host whose
        Python enqueues faster than the GPU executes the eager forward is 1 - 2 % faster, and on a slow host 30 % slower."""
        if policy not in ('replay', 'auto'):
            raise ValueError(f"policy must be 'replay' or 'auto', got {policy!r}")
        self.policy = policy
        self._choice = {}                             # 'auto': bkey -> 'replay' | 'eager'
        self.model = model
        self.capture_after = int(capture_after)
        dev = next(model.parameters()).device
        ops.stream_plan(dev)                          # the side streams exist before any capture
        self._pool = torch.cuda.graph_pool_handle()
        self._front = {}                              # key -> (graph, static rgb, static depth, front state | outputs)
        self._back = {}                               # key + counts -> (graph, outputs)            (dict order = LRU order)
        self._seen = {}
        self._stamp = None
        self.replays = {'front': 0, 'back': 0, 'eager_back': 0, 'eager': 0, 'captures': 0}
        self.launch = 'eager'                         # what the last call did: 'hipGraph replay' | 'eager' | 'hipGraph front + eager back'

    # ---- validity -------------------------------------------------------------------------------
    def _weights_stamp(self):
        m = self.model
        return (ops._MUTATION_GEN[0],) + tuple(t._version for t in m.parameters()) + tuple(t._version for t in m.buffers())

    def reset(self):
        for g in list(self._front.values()) + list(self._back.values()):
            g[0].reset()
        self._front.clear()
        self._back.clear()
        self._seen.clear()
        self._choice.clear()

    def _key(self, rgb, depth):
        m = self.model
        bo = getattr(m, 'branch_override', None)
        if getattr(m, 'ini_stage', False):
            bo = ('ini',) + tuple(int(v) for v in m.ini_branches)
        return (tuple(rgb.shape), tuple(depth.shape), bool(getattr(m, 'baseline', False)), bool(getattr(m, 'hard_gate', False)),
                float(getattr(m, 'temp', 0.0)), bool(getattr(m, 'compact', False)), bool(getattr(m, 'dual_stream', False)),
                None if bo is None else tuple(bo))

    def _eager(self, rgb, depth, return_weight):
        self.replays['eager'] += 1
        self.launch = 'eager'
        m = self.model
        if hasattr(m, 'hard_gate') or hasattr(m, 'block_rule'):
            return m(rgb, depth, True, True) if return_weight else m(rgb, depth, True)
        return m(rgb, depth)

    def _capture(self, fn):
        """fn() once eagerly (allocator, lazy initialisation, the inference caches of ops.conv2d_fused_eval), then captured."""
        keep = fn()
        torch.cuda.current_stream().synchronize()
        del keep
        graph = torch.cuda.CUDAGraph()
        saved, ops.CAPTURE_EVAL_CACHE = ops.CAPTURE_EVAL_CACHE, True
        try:
            with ops.capture_scope(), torch.cuda.graph(graph, pool=self._pool):
                out = fn()
        finally:
            ops.CAPTURE_EVAL_CACHE = saved
        self.replays['captures'] += 1
        return graph, out

    # ---- call -----------------------------------------------------------------------------------
    @torch.no_grad()
    def __call__(self, rgb, depth, return_weight=False):
        m = self.model
        gated = hasattr(m, 'forward_front')
        host_decisions = (getattr(m, 'ini_stage', False) and getattr(m, 'ini_branches', None) is None) or \
            (not gated and (hasattr(m, 'hard_gate') or hasattr(m, 'block_rule')))
        if m.training or host_decisions:
            return self._eager(rgb, depth, return_weight)
        stamp = self._weights_stamp()
        if stamp != self._stamp:
            self.reset()
            self._stamp = stamp
        key = self._key(rgb, depth)
        if self.policy == 'auto' and self._choice.get(key) == 'eager':
            return self._eager(rgb, depth, return_weight)
        fe = self._front.get(key)
        if fe is None:
            s_rgb, s_depth = rgb.clone(), depth.clone()
            if gated:
                graph, st = self._capture(lambda: m.forward_front(s_rgb, s_depth))
            else:
                graph, st = self._capture(lambda: m(s_rgb, s_depth))
            fe = self._front[key] = (graph, s_rgb, s_depth, st)
        graph, s_rgb, s_depth, st = fe
        s_rgb.copy_(rgb)
        s_depth.copy_(depth)
        graph.replay()
        self.replays['front'] += 1
        self.launch = 'hipGraph replay'
        if not gated:
            return st
        if getattr(m, 'save_weight_info', False):
            m.weight_list = torch.cat((m.weight_list, st['weight'].detach().cpu()))
        counts = m.stage_counts(st)                   # (the one host read of a compacted data-dependent forward)
        bkey = key + (None if counts is None else tuple(counts),)
        be = self._back.pop(bkey, None)
        if be is None:
            n = self._seen[bkey] = self._seen.get(bkey, 0) + 1
            known = counts is None or st['host_branch'] is not None
            if not known and n < self.capture_after:
                self.replays['eager_back'] += 1
                self.launch = 'hipGraph front + eager back'
                return m.forward_back(st, counts, True, return_weight)
            while len(self._back) >= self.MAX_GRAPHS:
                self._back.pop(next(iter(self._back)))[0].reset()
            be = self._capture(lambda: m.forward_back(st, counts, True, True))
        self._back[bkey] = be                          # most recently used last
        be[0].replay()
        self.replays['back'] += 1
        if self.policy == 'auto' and key not in self._choice and (counts is None or st['host_branch'] is not None):
            self._choice[key] = self._faster(fe, be, rgb, depth)
        if counts is not None:
            m.last_stage_batch = list(counts)
        out, weight = be[1]
        return (out, weight) if return_weight else out


def _infer_faster(self, fe, be, rgb, depth, reps=3):
    """'replay' or 'eager' for this key: `reps` forwards each way between events on the current stream (one-off)."""
    m = self.model
    graph, s_rgb, s_depth, _ = fe

    def timed(fn):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.current_stream().synchronize()
        import time
        t0 = time.perf_counter()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        return max(e0.elapsed_time(e1) * 1e-3, time.perf_counter() - t0)

    def replay():
        s_rgb.copy_(rgb)
        s_depth.copy_(depth)
        graph.replay()
        be[0].replay()
    t_replay = timed(replay)
    t_eager = timed(lambda: m(rgb, depth, True))
    self.auto_timing = {'replay_ms': round(1e3 * t_replay / reps, 3), 'eager_ms': round(1e3 * t_eager / reps, 3)}
    return 'replay' if t_replay <= t_eager else 'eager'


InferStep._faster = _infer_faster


def _dist_world(group=None):
    return (dp.dist.get_rank(group), dp.dist.get_world_size(group)) if dp.dist.is_initialized() else (0, 1)


@torch.no_grad()
def evaluate(model, batches, num_classes=40, hard=True, class_weight=None, shard=True, group=None, losses=None, infer_step=None):
    """batches: iterable of (rgb, depth, label_orig[N,H0,W0] with 0 = void[, label[N,H,W] at the network's resolution]).
    Returns (mIoU*100, cm).

    Data parallel (new; the reference is single-device): with `shard` and an initialised process group, rank r runs
    batches r, r + world, ... and the 40x40 confusion matrix (int64: exact) is all-reduced once at the end — every rank
    returns the same mIoU for 1/world of the forward passes.  The module buffers (BatchNorm running statistics) are
    broadcast from rank 0 first, so the result is rank 0's model evaluated on the whole set.

    `infer_step` (an InferStep of `model`): the forward is replayed as hipGraphs instead of enqueued launch by launch.

    `losses` (a dict) + `class_weight`: also accumulate validate()'s two validation losses (train.py:432-440;
    src/utils.py:53-97) from batches that carry the 4th element; the dict receives `sum_weighted`, `weight_sum`,
    `sum_unweighted`, `pixels` (python floats, summed over ranks)."""
    was_training = model.training
    model.eval()
    old_hard, model.hard_gate = getattr(model, 'hard_gate', False), hard
    dev = next(model.parameters()).device
    cm = torch.zeros(num_classes, num_classes, dtype=torch.int64, device=dev)
    acc4 = torch.zeros(4, dtype=torch.float64, device=dev)
    cw = None if class_weight is None else torch.as_tensor(class_weight, dtype=torch.float32, device=dev)
    rank, world = _dist_world(group) if shard else (0, 1)
    if world > 1:
        # the replicas' BatchNorm running statistics differ (each rank updates them from its own shard): a sharded
        # evaluation must describe ONE model — rank 0's, the one train.py checkpoints
        dp.broadcast_buffers(model, 0, group)
        ops.note_mutation()
    for i, batch in enumerate(batches):
        if i % world != rank:
            continue
        rgb, depth, label = batch[:3]
        logits = infer_step(rgb, depth) if infer_step is not None else model(rgb, depth, True)
        if losses is not None and cw is not None and len(batch) > 3:
            ops.validation_loss_accumulate(logits, batch[3], cw, acc4)
        ops.eval_confusion(logits, label, cm)     # resize + argmax + void mask + bincount, one kernel
    model.hard_gate = old_hard
    model.train(was_training)
    if world > 1:
        if dp.dist.get_backend(group) == 'gloo':
            cm_h, acc_h = cm.cpu(), acc4.cpu()
            dp.dist.all_reduce(cm_h, group=group)
            dp.dist.all_reduce(acc_h, group=group)
            cm, acc4 = cm_h.to(dev), acc_h.to(dev)
        else:
            dp.dist.all_reduce(cm, group=group)
            dp.dist.all_reduce(acc4, group=group)
    if losses is not None:
        losses.update(zip(('sum_weighted', 'weight_sum', 'sum_unweighted', 'pixels'), acc4.tolist()))
    cmd = cm.double()
    iou = cmd.diag() / (cmd.sum(1) + cmd.sum(0) - cmd.diag() + 1e-15)
    return iou.mean().item() * 100.0, cm.cpu()


def validate(model, loaders_by_camera, class_weight, logs=None, split='test', soft_eval=False, dynamic=True,
             weighted_pixel_sum=None, num_classes=40, shard=True, group=None):
    """FusionDynMM/train.py:368-551 `validate` on the HIP path: one confusion matrix per camera (all images of a camera
    share a resolution, :398-404), mIoU per camera under `mIoU_{split}_{camera}`, the class-weighted validation loss
    `loss_{split}` = sum_px w[t]*CE / weighted_pixel_sum (src/utils.py:53-74; `weighted_pixel_sum` = sum_c pixels_c*w_c
    over the validation labels, train.py:104-108 — when None it is accumulated from the labels the loss is evaluated on,
    which is the same number whenever the loader's labels are the data set's) and `loss_{split}_unweighted`
    (:77-97), hard gates unless `soft_eval` (:384), gate statistics through start_weight / end_weight (:385-387, :512).
    `loaders_by_camera`: {camera: iterable of dicts with image / depth / label_orig / label}.  Returns (miou, logs)
    with `miou[camera]` and the confusion matrices under logs['confusion_matrices'] (train.py pickles them, :521-525)."""
    import time
    t0 = time.time()
    logs = {} if logs is None else logs
    if dynamic and hasattr(model, 'start_weight'):
        model.start_weight()
    miou, cms, tot = {}, {}, dict(sum_weighted=0.0, weight_sum=0.0, sum_unweighted=0.0, pixels=0.0)
    for camera, loader in loaders_by_camera.items():
        part = {}
        batches = ((s['image'], s['depth'], s['label_orig'], s['label']) for s in loader)
        miou[camera], cms[camera] = evaluate(model, batches, num_classes, hard=not soft_eval, class_weight=class_weight,
                                             shard=shard, group=group, losses=part)
        for k in tot:
            tot[k] += part[k]
    if dynamic and hasattr(model, 'end_weight'):
        model.end_weight(print_each=True)
    wsum = float(weighted_pixel_sum) if weighted_pixel_sum is not None else tot['weight_sum']
    logs[f'loss_{split}'] = tot['sum_weighted'] / wsum if wsum else float('nan')
    logs[f'loss_{split}_unweighted'] = tot['sum_unweighted'] / tot['pixels'] if tot['pixels'] else float('nan')
    for camera in loaders_by_camera:
        logs[f'mIoU_{split}_{camera}'] = miou[camera]
    logs['time_validation'] = time.time() - t0
    logs['confusion_matrices'] = cms
    return miou, logs
