"""The callers either side of the hot path, on the HIP path:

  TrainStep  = FusionDynMM/train.py:289-324 — zero grads, forward, weighted multi-scale CE,
               total = sum(CE_s) + ratio*max(0, flop_loss - budget), backward, [DP all-reduce],
               SGD-Nesterov — as ONE hipGraph-capturable sequence: parameters, gradients and momentum
               live in flat fp32 buffers (every nn.Parameter / .grad is a view), the learning rate is a
               device scalar, so a whole optimisation step replays with no host work.
  evaluate   = FusionDynMM/eval.py:104-146 — forward(test=True), bilinear resize to the label size,
               arg-max, void mask, confusion matrix, mIoU*100.
"""
import ctypes as C

import torch

from . import dp, ops
from . import lib as L


class FlatParameters:
    """Re-home every parameter of `module` into one contiguous fp32 buffer (views keep state_dict,
    load_state_dict and autograd working unchanged)."""

    def __init__(self, module):
        self.params = [p for p in module.parameters()]
        dev = self.params[0].device
        total = sum(p.numel() for p in self.params)
        self.flat = torch.empty(total, device=dev, dtype=torch.float32)
        off = 0
        with torch.no_grad():
            for p in reversed(self.params):        # same order as dp.GradBucketReducer
                n = p.numel()
                self.flat[off:off + n].copy_(p.detach().reshape(-1))
                p.data = self.flat[off:off + n].view_as(p)
                off += n


class SGDNesterov:
    """Fused flat SGD with Nesterov momentum and L2 weight decay (torch.optim.SGD semantics,
    train.py:557-563): one kernel over all parameters."""

    def __init__(self, flat_params, flat_grads, lr, momentum=0.9, weight_decay=1e-4):
        assert flat_params.numel() == flat_grads.numel()
        self.p, self.g = flat_params, flat_grads
        self.buf = torch.zeros_like(flat_params)
        self.lr = torch.tensor([float(lr)], device=flat_params.device, dtype=torch.float32)
        self.momentum, self.weight_decay = float(momentum), float(weight_decay)

    def set_lr(self, lr):
        self.lr.fill_(float(lr))            # device scalar: visible to an already-captured graph

    def step(self):
        lib = L.load()
        ops.note_mutation()                 # parameters are rewritten through raw pointers
        L.check(lib.dynmm_sgd_nesterov(self.p.data_ptr(), self.g.data_ptr(), self.buf.data_ptr(),
                                       C.c_size_t(self.p.numel()), self.lr.data_ptr(), self.momentum,
                                       self.weight_decay, 1.0, torch.cuda.current_stream().cuda_stream),
                'sgd_nesterov')

    def state_dict(self):
        return {'momentum_buffer': self.buf, 'lr': self.lr}


class TrainStep:
    def __init__(self, model, class_weight, lr, momentum=0.9, weight_decay=1e-4, loss_ratio=0.0,
                 flop_budget=0.0, use_graph=False, bucket_mb=32.0, multi_stream=True):
        self.model = model
        self.cw = torch.as_tensor(class_weight, dtype=torch.float32, device=next(model.parameters()).device)
        self.flatp = FlatParameters(model)
        self.reducer = dp.GradBucketReducer(model.parameters(), bucket_mb=bucket_mb, overlap=not use_graph)
        frozen = [p for p in model.parameters() if not p.requires_grad]
        if frozen:
            raise NotImplementedError('TrainStep with frozen parameters: build it after model.freeze() is not '
                                      'supported yet; use per-parameter torch.optim.SGD for --freeze runs')
        self.opt = SGDNesterov(self.flatp.flat, self.reducer.flat, lr, momentum, weight_decay)
        ops.DIRECT_GRAD = True      # one backward per zero(): gradients are written in place, not accumulated
        # 3-stream schedule: RGB encoder | depth encoder | conv weight gradients (see nn/net.py, ops.py)
        ops.ASYNC_WGRAD = bool(multi_stream)
        if hasattr(model, 'dual_stream'):
            model.dual_stream = bool(multi_stream)
        self.loss_ratio, self.flop_budget = float(loss_ratio), float(flop_budget)
        self.use_graph = use_graph
        self._graph = None
        self._static = None
        self.last = None       # dict of device tensors: losses[4], loss_flop, total

    def _body(self, rgb, depth, targets):
        self.reducer.zero()
        res = self.model(rgb, depth)
        if len(res) == 2 and isinstance(res[0], (tuple, list)):
            outs, lf = res                                   # SkipGateESANet: ((out, out8, out16, out32), flop loss)
        else:
            outs, lf = res, torch.zeros((), device=rgb.device)   # SkipESANet: the four outputs only
        losses = [ops.cross_entropy_2d(o, t, self.cw) for o, t in zip(outs, targets)]
        seg = losses[0]
        for l in losses[1:]:
            seg = seg + l
        total = seg + self.loss_ratio * torch.clamp(lf - self.flop_budget, min=0.0) if self.loss_ratio > 0 else seg
        total.backward()
        ops.join_async()
        self.last = {'losses': torch.stack([l.detach() for l in losses]), 'loss_flop': lf.detach(),
                     'total': total.detach()}

    def __call__(self, rgb, depth, targets):
        """targets: list of 4 label maps (0 = void), uint8/float/int, at scales 1, 1/8, 1/16, 1/32."""
        targets = [t if t.dtype == torch.uint8 else t.to(torch.uint8) for t in targets]
        if not self.use_graph:
            self._body(rgb, depth, targets)
            self.reducer.finish()
            self.opt.step()
            return self.last
        if self._graph is None:
            self._static = (rgb.clone(), depth.clone(), [t.clone() for t in targets])
            # snapshot BEFORE the side stream forks, so the warm-up cannot race the clones
            sd = {k: v.clone() for k, v in self.model.state_dict().items()}
            mom = self.opt.buf.clone()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):       # warm-up outside capture (allocator, lazy init)
                self._body(*self._static)
            torch.cuda.current_stream().wait_stream(side)
            self.model.load_state_dict(sd)       # undo the warm-up's BN running-stat updates
            self.opt.buf.copy_(mom)
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph):
                self._body(*self._static)
                if self.reducer.world == 1:
                    self.opt.step()
            self.model.load_state_dict(sd)
            self.opt.buf.copy_(mom)
        s_rgb, s_depth, s_t = self._static
        s_rgb.copy_(rgb)
        s_depth.copy_(depth)
        for a, b in zip(s_t, targets):
            a.copy_(b)
        self._graph.replay()
        ops.note_mutation()                 # the replayed step updated running statistics / parameters
        if self.reducer.world > 1:
            self.reducer.finish()
            self.opt.step()
        return self.last


@torch.no_grad()
def evaluate(model, batches, num_classes=40, hard=True):
    """batches: iterable of (rgb, depth, label_orig[N,H0,W0] with 0 = void).  Returns (mIoU*100, cm)."""
    was_training = model.training
    model.eval()
    old_hard, model.hard_gate = model.hard_gate, hard
    cm = torch.zeros(num_classes, num_classes, dtype=torch.int64, device=next(model.parameters()).device)
    for rgb, depth, label in batches:
        ops.eval_confusion(model(rgb, depth, True), label, cm)     # resize + argmax + void mask + bincount, one kernel
    model.hard_gate = old_hard
    model.train(was_training)
    cmd = cm.double()
    iou = cmd.diag() / (cmd.sum(1) + cmd.sum(0) - cmd.diag() + 1e-15)
    return iou.mean().item() * 100.0, cm.cpu()
