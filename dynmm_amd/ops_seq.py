"""autograd shells for the modality-level DynMM path (ModalityDynMM/affect/affect_dyn.py, BASELINE configs[4]).

Activations are [B, D, T] fp32 (the layout the reference feeds its Conv1d after `x.permute([0, 2, 1])`); every Linear
/ Conv1d(k=1) is `linear_bdt` = the implicit-GEMM MFMA convolution of ops.conv2d with H = 1, W = T.  The kernels
added for this path live in csrc/seq.hip.  As everywhere in dynmm_amd there is no CPU / eager fallback.
"""
import ctypes as C
import itertools

import torch
from torch.autograd import Function

from . import lib as L
from . import ops
from .ops import _chk, _grad_dst, _grads_enqueued, _lib, _p, _ptr_array, _stream


def linear_bdt(x, weight, bias=None, act=None, defer_mask=False, mask_input=False, link=None):
    """act(W x + b) over the channel axis of x [B, Ci, T] (or [B, Ci]): weight [Co, Ci] (nn.Linear,
    in_proj_weight) or [Co, Ci, 1] (nn.Conv1d).  defer_mask / mask_input: the ReLU backward of a `defer_mask` layer is
    applied in the input-gradient epilogue of its single consumer (`mask_input`), as in the conv blocks.  link: ops.GradLink
    through which x's OTHER consumer (the residual input of the layer's LayerNorm) hands its gradient to this op's
    input-gradient epilogue."""
    squeeze = x.dim() == 2
    x4 = x.reshape(x.shape[0], x.shape[1], 1, -1) if not squeeze else x.reshape(x.shape[0], x.shape[1], 1, 1)
    shape4 = (weight.shape[0], weight.shape[1], 1, 1)
    train_w = weight.requires_grad and torch.is_grad_enabled()
    # the parameter itself owns the gradient (same memory layout as its [Co, Ci, 1, 1] alias): the in-place gradient
    # protocol writes straight into weight.grad, plain autograd gets it back through _Alias
    w4 = _Alias.apply(weight, shape4) if train_w else weight.detach().view(shape4)
    fuse = torch.is_grad_enabled() and x.requires_grad
    y = ops.conv2d(x4, w4, bias, 1, 0, act, defer_mask=defer_mask and fuse, mask_input=mask_input and fuse,
                   link=link if fuse else None, w_owner=weight if train_w else None)
    return y.reshape(y.shape[0], y.shape[1]) if squeeze else y.reshape(y.shape[0], y.shape[1], -1)


# ---------------------------------------------------------------------------------------------------------------
# dropout (nn.TransformerEncoderLayer trains with p = 0.1 at four sites per layer)
# ---------------------------------------------------------------------------------------------------------------
# A site is an integer; its keep decisions are Philox(seed, (site << 32) + device step counter, element index)
# (csrc/seq.hip): nothing is stored for the backward, and a captured step draws new masks at every replay because the
# counter lives on the device (advance_dropout_step(), called once per training step).
MASKS = None            # tests: callable(site_name, shape) -> uint8 keep flags (device tensor) or None
_SITE_IDS = itertools.count(1)
_STEP = {}


def new_sites(n):
    """n consecutive site ids (one encoder layer takes 4)."""
    first = next(_SITE_IDS)
    for _ in range(n - 1):
        next(_SITE_IDS)
    return first


def dropout_step(device):
    t = _STEP.get(device)
    if t is None:
        t = _STEP[device] = torch.zeros(1, device=device, dtype=torch.int64)
    return t


def advance_dropout_step(device):
    dropout_step(device).add_(1)


class Drop:
    """One dropout site of one call: probability, site id and (tests) the injected keep flags."""

    def __init__(self, p, site, name, shape, device):
        self.p, self.site = float(p), int(site)
        self.mask = None
        if MASKS is not None and self.p > 0:
            m = MASKS(name, tuple(shape))
            if m is not None:
                if m.dtype != torch.uint8 or tuple(m.shape) != tuple(shape) or not m.is_cuda:
                    raise L.DynmmHipError(f'dropout mask for {name}: expected uint8 {tuple(shape)} on the device')
                self.mask = m.contiguous()
        self.step = dropout_step(device) if self.p > 0 else None

    def desc(self):
        return L.Dropout(_p(self.mask), _p(self.step), ops._PHILOX_SEED, self.site << 32, self.p)


def _drop_arg(d):
    return C.byref(d.desc()) if d is not None and d.p > 0 else None


class _DropoutBDT(Function):
    @staticmethod
    def forward(ctx, x, drop):
        x = _chk(x, 'x')
        y = torch.empty_like(x)
        L.check(_lib().dynmm_dropout_apply(_p(x), _p(y), C.c_size_t(x.numel()), _drop_arg(drop), _stream()), 'dropout')
        ctx.drop = drop
        return y

    @staticmethod
    def backward(ctx, g):
        g = _chk(g, 'grad')
        dx = torch.empty_like(g)
        L.check(_lib().dynmm_dropout_apply(_p(g), _p(dx), C.c_size_t(g.numel()), _drop_arg(ctx.drop), _stream()), 'dropout')
        return dx, None


def dropout_bdt(x, p, site, name='dropout'):
    """x * keep / (1 - p) (nn.Dropout in training mode); identity when p == 0."""
    if not p > 0:
        return x
    return _DropoutBDT.apply(x, Drop(p, site, name, x.shape, x.device))


class _Alias(Function):
    """weight viewed as [Co, Ci, 1, 1]; the gradient comes back in the same memory layout."""

    @staticmethod
    def forward(ctx, w, shape):
        ctx.shape = tuple(w.shape)
        # under the in-place gradient protocol the convolution returns no gradient for the alias: without this autograd would
        # materialise a zero tensor for it and ADD it to the parameter's .grad (round 6 census of the MOSEI step: 61 zeros + 61
        # add_ launches per step)
        ctx.set_materialize_grads(False)
        return w.view(shape)

    @staticmethod
    def backward(ctx, g):
        return (None if g is None else g.reshape(ctx.shape)), None


def _ln_ws(lib, like, B, D, T, params):
    """(workspace, bytes) of the LayerNorm backward's per-workgroup parameter sums; (None, 0) without parameter gradients."""
    if not params:
        return None, 0
    nb = lib.dynmm_layernorm_bwd_workspace_bytes(B, D, T)
    return torch.empty(nb // 4, device=like.device, dtype=torch.float32), nb


class _LayerNormBDT(Function):
    @staticmethod
    def forward(ctx, x, res, gamma, beta, eps, drop=None, res_link=None):
        lib = _lib()
        ctx.res_link = res_link
        x, res, gamma, beta = _chk(x, 'x'), _chk(res, 'res'), _chk(gamma, 'gamma'), _chk(beta, 'beta')
        B, D, T = x.shape
        y = torch.empty_like(x)
        mean = torch.empty(B * T, device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        L.check(lib.dynmm_layernorm_drop_fwd(_p(x), _p(res), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), B, D, T,
                                             float(eps), _drop_arg(drop), _stream()), 'layernorm_fwd')
        ctx.drop = drop
        ctx.save_for_backward(x, res, gamma, mean, rstd)
        ctx.g_param, ctx.b_param = gamma, beta
        return y

    @staticmethod
    def backward(ctx, g):
        lib = _lib()
        x, res, gamma, mean, rstd = ctx.saved_tensors
        g = _chk(g, 'grad')
        B, D, T = x.shape
        need_x = ctx.needs_input_grad[0]
        need_res = res is not None and ctx.needs_input_grad[1]
        dropping = ctx.drop is not None and ctx.drop.p > 0
        # without dropout the two branches of the sum receive the same tensor; with it x gets its keep factor on top
        dres = torch.empty_like(x) if (need_res or (need_x and not dropping)) else None
        dx = torch.empty_like(x) if (need_x and dropping) else None
        dg = dg_ret = db = db_ret = None
        if ctx.needs_input_grad[2]:
            dg, dg_ret = _grad_dst(ctx.g_param)
            db, db_ret = _grad_dst(ctx.b_param)
        ws, nb = _ln_ws(lib, g, B, D, T, dg is not None)
        L.check(lib.dynmm_layernorm_drop_bwd_ws(_p(g), _p(x), _p(res), _p(gamma), _p(mean), _p(rstd), _p(dx), _p(dres), _p(dg),
                                                _p(db), B, D, T, _drop_arg(ctx.drop), _p(ws), nb, _stream()), 'layernorm_bwd')
        _grads_enqueued()
        if need_res and ctx.res_link is not None:
            # the residual input's other consumer adds this gradient in its own input-gradient epilogue (ops.GradLink): no
            # accumulation pass by autograd
            ctx.res_link.dres = dres.reshape(B, D, 1, T)
            need_res = False
        return ((dx if dropping else dres) if need_x else None), (dres if need_res else None), dg_ret, db_ret, None, None, None


def layernorm_bdt(x, gamma, beta, eps=1e-5, residual=None, drop=None, res_link=None):
    """LayerNorm over D of (dropout(x) + residual), x [B, D, T]; drop = (p, site, name) or None.  res_link: an ops.GradLink
    shared with the linear_bdt that also consumes `residual` and whose output this layer's x descends from (so its backward
    runs after this one): the residual branch's gradient is added there instead of by autograd."""
    d = Drop(drop[0], drop[1], drop[2], x.shape, x.device) if drop is not None and drop[0] > 0 else None
    if not (res_link is not None and residual is not None and torch.is_grad_enabled() and residual.requires_grad):
        res_link = None
    return _LayerNormBDT.apply(x, residual, gamma, beta, eps, d, res_link)


FFN_FUSED = True         # (module attribute: tests compare with the unfused feed-forward)


def ffn_fused_ok(x, w1, b1, w2, b2):
    """the one-launch feed-forward block (csrc/seq_ffn.hip) serves this layer"""
    if not (FFN_FUSED and x.dim() == 3 and b1 is not None and b2 is not None):
        return False
    B, D, T = x.shape
    return bool(_lib().dynmm_ffn_supported(B, D, T, w1.shape[0])) and \
        all(t.data_ptr() % 16 == 0 and t.is_contiguous() for t in (w1, b1, w2))


def _wgrad_1x1(x3, gy3, w_param, b_param):
    """weight (+ bias) gradient of a Linear over the channel axis of x3 [B, Ci, T] given gy3 [B, Co, T]: queued for a grouped
    launch under the in-place gradient protocol, launched at once otherwise.  Returns (dw, db) for autograd (None = written in
    place)."""
    lib = _lib()
    B, Ci, T = x3.shape
    Co = gy3.shape[1]
    g = L.ConvGeom(B, Ci, 1, T, Co, 1, T, 1, 1, 1, 1, 0, 0, Ci)
    if (ops.DIRECT_GRAD and ops.WGRAD_GROUP > 1 and ops._direct_ok(w_param) and ops._direct_ok(b_param) and
            bool(lib.dynmm_conv2d_wgrad_groupable(C.byref(g)))):
        ops._queue_wgrad(g, x3, gy3, w_param, b_param)
        return None, None
    dw, dw_ret = _grad_dst(w_param)
    db, db_ret = _grad_dst(b_param)
    nbytes = lib.dynmm_conv2d_wgrad_workspace_bytes(C.byref(g))
    ws = torch.empty(max(nbytes // 4, 1), device=x3.device, dtype=torch.float32)
    L.check(lib.dynmm_conv2d_wgrad(_p(x3), None, _p(gy3), _p(dw), _p(db), _p(ws), nbytes, C.byref(g), _stream()), 'conv2d_wgrad')
    _grads_enqueued()
    return dw_ret, db_ret


class _FFNBlock(Function):
    """LayerNorm(h + dropout2(linear2(dropout(relu(linear1(h)))))): the second half of a post-norm encoder layer.  Forward =
    ffn_kernel + ln_fwd_kernel; backward = LayerNorm backward, ffn_kernel<BWD>, two (queued) weight-gradient problems and one
    ordered sum of the residual branch's gradient with the partial sums of the feed-forward branch's."""

    @staticmethod
    def forward(ctx, h, w1, b1, w2, b2, gamma, beta, eps, drop_f, drop2):
        lib = _lib()
        h = _chk(h, 'x')
        B, D, T = h.shape
        F = w1.shape[0]
        ns = lib.dynmm_ffn_nsplit(B, D, T, F)
        hidden = torch.empty((B, F, T), device=h.device, dtype=torch.float32)
        parts = torch.empty((ns, B, D, T), device=h.device, dtype=torch.float32)
        L.check(lib.dynmm_ffn_fwd(_p(h), _p(w1), _p(b1), _p(w2), _p(hidden), _p(parts), B, D, T, F, ns, _drop_arg(drop_f),
                                  _stream()), 'ffn_fwd')
        y, xsum = torch.empty_like(h), torch.empty_like(h)
        mean = torch.empty(B * T, device=h.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        L.check(lib.dynmm_layernorm_parts_fwd(_p(parts), ns, _p(b2), _p(xsum), _p(h), _p(gamma), _p(beta), _p(y), _p(mean),
                                              _p(rstd), B, D, T, float(eps), _drop_arg(drop2), _stream()), 'layernorm_parts_fwd')
        ctx.drop_f, ctx.drop2, ctx.ns = drop_f, drop2, ns
        ctx.save_for_backward(h, hidden, xsum, mean, rstd, w1, w2, gamma)
        ctx.params = (w1, b1, w2, b2, gamma, beta)
        return y

    @staticmethod
    def backward(ctx, g):
        lib = _lib()
        h, hidden, xsum, mean, rstd, w1, w2, gamma = ctx.saved_tensors
        pw1, pb1, pw2, pb2, pgamma, pbeta = ctx.params
        g = _chk(g, 'grad')
        B, D, T = h.shape
        F, ns = w1.shape[0], ctx.ns
        st = _stream()
        dropping2 = ctx.drop2 is not None and ctx.drop2.p > 0
        # slab ns = the residual branch's gradient, slabs 0 .. ns-1 = the feed-forward branch's partial sums
        slabs = torch.empty((ns + 1, B, D, T), device=h.device, dtype=torch.float32)
        dres = slabs[ns]
        dout = torch.empty_like(h) if dropping2 else dres
        dg = dg_ret = db = db_ret = None
        if ctx.needs_input_grad[5]:
            dg, dg_ret = _grad_dst(pgamma)
            db, db_ret = _grad_dst(pbeta)
        ws, nb = _ln_ws(lib, g, B, D, T, dg is not None)
        L.check(lib.dynmm_layernorm_drop_bwd_ws(_p(g), _p(xsum), _p(h), _p(gamma), _p(mean), _p(rstd),
                                                _p(dout) if dropping2 else None, _p(dres), _p(dg), _p(db), B, D, T,
                                                _drop_arg(ctx.drop2), _p(ws), nb, st), 'layernorm_bwd')
        _grads_enqueued()
        dhid = torch.empty_like(hidden)
        pf = ctx.drop_f.p if ctx.drop_f is not None else 0.0
        L.check(lib.dynmm_ffn_bwd_data(_p(dout), _p(hidden), _p(w1), _p(w2), _p(dhid), _p(slabs), B, D, T, F, ns, float(pf), st),
                'ffn_bwd_data')
        # (a layer's weight and bias gradients come out of one weight-gradient problem: either of the pair asks for it)
        dw1 = db1 = dw2 = db2 = None
        if ctx.needs_input_grad[3] or ctx.needs_input_grad[4]:
            dw2, db2 = _wgrad_1x1(hidden, dout, pw2, pb2)
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            dw1, db1 = _wgrad_1x1(h, dhid, pw1, pb1)
        dw1, db1, dw2, db2 = (t if need else None for t, need in zip((dw1, db1, dw2, db2), ctx.needs_input_grad[1:5]))
        dh = None
        if ctx.needs_input_grad[0]:
            dh = torch.empty_like(h)
            L.check(lib.dynmm_reduce_slabs(_p(slabs), _p(dh), h.numel(), ns + 1, st), 'reduce_slabs')
        return dh, dw1, db1, dw2, db2, dg_ret, db_ret, None, None, None


def ffn_block(h, layer, drop_f, drop2):
    """norm2(h + dropout2(linear2(dropout(relu(linear1(h)))))) of an nn.TransformerEncoderLayer; drop_* = (p, site, name)."""
    B, D, T = h.shape
    F = layer.linear1.weight.shape[0]
    df = Drop(drop_f[0], drop_f[1], drop_f[2], (B, F, T), h.device) if drop_f[0] > 0 else None
    d2 = Drop(drop2[0], drop2[1], drop2[2], h.shape, h.device) if drop2[0] > 0 else None
    return _FFNBlock.apply(h, layer.linear1.weight, layer.linear1.bias, layer.linear2.weight, layer.linear2.bias,
                           layer.norm2.weight, layer.norm2.bias, layer.norm2.eps, df, d2)


class _MHACore(Function):
    @staticmethod
    def forward(ctx, qkv, heads, drop=None):
        lib = _lib()
        qkv = _chk(qkv, 'qkv')
        B, D3, T = qkv.shape
        D = D3 // 3
        out = torch.empty((B, D, T), device=qkv.device, dtype=torch.float32)
        probs = torch.empty((B * heads, T, T), device=qkv.device, dtype=torch.float32)
        L.check(lib.dynmm_mha_drop_fwd(_p(qkv), _p(out), _p(probs), B, D, T, heads, _drop_arg(drop), _stream()), 'mha_fwd')
        ctx.drop = drop
        ctx.save_for_backward(qkv, probs)
        ctx.heads = heads
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib()
        qkv, probs = ctx.saved_tensors
        g = _chk(g, 'grad')
        B, D3, T = qkv.shape
        dqkv = torch.empty_like(qkv)
        L.check(lib.dynmm_mha_drop_bwd(_p(g), _p(qkv), _p(probs), _p(dqkv), B, D3 // 3, T, ctx.heads, _drop_arg(ctx.drop),
                                       _stream()), 'mha_bwd')
        return dqkv, None, None


def mha_core(qkv, heads, drop=None):
    """dropout(softmax(q k^T / sqrt(dh))) v per head for qkv [B, 3D, T] (q | k | v along channels) -> [B, D, T];
    drop = (p, site, name) or None: dropout on the [B*heads, T, T] probabilities."""
    B, D3, T = qkv.shape
    d = Drop(drop[0], drop[1], drop[2], (B * heads, T, T), qkv.device) if drop is not None and drop[0] > 0 else None
    return _MHACore.apply(qkv, heads, d)


class _MoEBlend(Function):
    """out[B,1] = sum_k w_k pred_k, w = DiffSoftmax(logits/temp, hard); aux = mean w[:, K-1]  (affect_dyn.py:152-165)."""

    @staticmethod
    def forward(ctx, logits, temp, hard, *preds):
        lib = _lib()
        logits = _chk(logits, 'logits')
        preds = [_chk(p.reshape(-1), 'pred') for p in preds]
        B, K = logits.shape
        f32 = dict(device=logits.device, dtype=torch.float32)
        out, weight, scal = torch.empty(B, **f32), torch.empty((B, K), **f32), torch.empty(3, **f32)
        L.check(lib.dynmm_moe_head(_p(logits), _ptr_array(preds), K, None, float(temp), int(bool(hard)), 0.0, _p(out),
                                   _p(weight), _p(scal), None, None, B, _stream()), 'moe_head')
        ctx.save_for_backward(logits, weight, *preds)
        ctx.temp = float(temp)
        ctx.mark_non_differentiable(weight)
        return out.reshape(B, 1), scal[1], weight

    @staticmethod
    def backward(ctx, d_out, d_aux, _dw):
        lib = _lib()
        logits, weight = ctx.saved_tensors[:2]
        preds = list(ctx.saved_tensors[2:])
        B, K = logits.shape
        f32 = dict(device=logits.device, dtype=torch.float32)
        d_out = None if d_out is None else _chk(d_out.reshape(-1), 'd_out')
        d_aux = None if d_aux is None else _chk(d_aux.reshape(1), 'd_aux')
        dps = [torch.empty(B, **f32) for _ in preds]
        dl = torch.empty((B, K), **f32)
        L.check(lib.dynmm_moe_blend_bwd(_p(d_out), _p(d_aux), _p(logits), _ptr_array(preds), K, _p(weight), ctx.temp,
                                        _ptr_array(dps), _p(dl), B, _stream()), 'moe_blend_bwd')
        return (dl, None, None, *[d.reshape(B, 1) for d in dps])


def moe_blend(logits, preds, temp=1.0, hard=False):
    """(out [B,1], aux scalar, weight [B,K]) — the gated mixture of the experts' predictions."""
    return _MoEBlend.apply(logits, temp, hard, *preds)


def moe_loss_backward(logits, preds, target, temp, hard, reg):
    """Supervised_Learning.py:120-141 for a DynMM mixture, on the device: blend, L1 loss, loss1 + reg * aux, and the
    backward pass seeded straight from the kernel (no PyTorch arithmetic kernels).  Returns
    {'out': [B,1], 'weight': [B,K], 'loss1', 'aux', 'total'}."""
    lib = _lib()
    logits = _chk(logits, 'logits')
    flat = [_chk(p.reshape(-1), 'pred') for p in preds]
    tgt = _chk(target.reshape(-1).float(), 'target')
    B, K = logits.shape
    f32 = dict(device=logits.device, dtype=torch.float32)
    out, weight, scal = torch.empty(B, **f32), torch.empty((B, K), **f32), torch.empty(3, **f32)
    dps = [torch.empty(B, **f32) if p.requires_grad else None for p in preds]
    dl = torch.empty((B, K), **f32)
    arr = (C.c_void_p * K)(*[(None if d is None else d.data_ptr()) for d in dps])
    L.check(lib.dynmm_moe_head(_p(logits.detach()), _ptr_array(flat), K, _p(tgt), float(temp), int(bool(hard)), float(reg),
                               _p(out), _p(weight), _p(scal), arr, _p(dl), B, _stream()), 'moe_head')
    roots, grads = [], []
    for p, d in zip(preds, dps):
        if d is not None:
            roots.append(p)
            grads.append(d.reshape(p.shape))
    if logits.requires_grad:
        roots.append(logits)
        grads.append(dl)
    if roots:
        torch.autograd.backward(roots, grads)
    return {'out': out.reshape(B, 1), 'weight': weight, 'loss1': scal[0:1], 'aux': scal[1:2], 'total': scal[2:3]}


def clip_grad_norm(flat_grad, max_norm):
    """(norm, coef) device tensor [2]: coef = min(1, max_norm / (norm + 1e-6)), torch.nn.utils.clip_grad_norm_."""
    lib = _lib()
    ws = torch.empty(lib.dynmm_clip_grad_norm_workspace_bytes() // 8, device=flat_grad.device, dtype=torch.float64)
    out = torch.empty(2, device=flat_grad.device, dtype=torch.float32)
    L.check(lib.dynmm_clip_grad_norm(_p(flat_grad), C.c_size_t(flat_grad.numel()), float(max_norm), ws.data_ptr(), _p(out),
                                     _stream()), 'clip_grad_norm')
    return out
