"""Modality-level DynMM on CMU-MOSEI features (ModalityDynMM/affect/affect_dyn.py) on the HIP path.

PARITY UNPINNED.  The reference builds its experts from MultiBench (`unimodals.common_models.Transformer / MLP`,
`fusions.common_fusions.Concat`, `training_structures.Supervised_Learning.MMDL`), which is neither vendored in
/root/reference nor pinned to a commit (ModalityDynMM README; affect_dyn.py:12-15).  The modules below restate
MultiBench's published definitions:

  Transformer(n_features, dim)   Conv1d(n_features, dim, 1, bias=False) over the feature axis, then
                                 nn.TransformerEncoder(nn.TransformerEncoderLayer(d_model=dim, nhead=5), num_layers=5)
                                 (post-norm, ReLU, dim_feedforward 2048), output = the LAST time step.  The padding
                                 lengths that accompany the input are ignored (`x = x[0]`).
  MLP(indim, hiddim, outdim)     fc -> ReLU -> fc2
  Concat                         torch.cat(..., dim=1)
  MMDL(encoders, fusion, head)   head(fusion([enc_i([x_i, len_i])]))          (Supervised_Learning.py:16-51 — vendored)

and, from the reference's own file, DynMMNetV2 (affect_dyn.py:107-175: expert 1 = text Transformer + MLP head,
expert 2 = 3-modality late-fusion MMDL, gate = Transformer(409, 10) + Linear(10, 2), DiffSoftmax, convex blend,
regulariser mean(w[:, 1])) and DynMMNet (affect_dyn.py:31-104: three uni-modal experts, 3-way gate).

Modules are parameter containers with torch's own state_dict keys (`conv.weight`, `transformer.layers.N.self_attn.
in_proj_weight`, ...), so `expert.state_dict()` of a trained MultiBench module loads with load_state_dict.  Dropout
(p = 0.1 inside nn.TransformerEncoderLayer) is applied in training mode at torch's four sites per layer, with a Philox
stream of its own (ops.manual_seed; torch's generator cannot be reproduced bit for bit, the tests inject the keep flags
on both sides); `.eval()` switches it off as in torch.
"""

import torch
import torch.nn as nn

from .. import ops
from .. import ops_seq as S

FEATURES = {'visual': 35, 'audio': 74, 'text': 300}      # CMU-MOSEI (affect/count_flop.py:52)

# The gate transformer and the experts' encoders are independent until the mixture: each runs on a HIP stream of its own
# (five 5-layer transformers on 50-token sequences are chains of ~10-100 us kernels that do not fill 256 CUs one at a time).
# Autograd replays a node on its forward stream, so the backward is concurrent too; a captured step keeps the branches as
# parallel paths of the hipGraph.
BRANCH_STREAMS = True     # (module attribute; False: one stream)
LINK_RESIDUAL = True      # (module attribute; False: autograd sums the two gradients of a layer's input)
_POOL, _ALL = [], []


def run_branches(fns):
    """[f() for f in fns], fns[1:] each on a side stream forked from / joined to the current one."""
    if not (BRANCH_STREAMS and len(fns) > 1 and torch.cuda.is_available()):
        return [f() for f in fns]
    main = torch.cuda.current_stream()
    taken = []
    for _ in fns[1:]:
        if not _POOL:
            _POOL.append(torch.cuda.Stream())
            _ALL.append(_POOL[-1])
        taken.append(_POOL.pop())
    # fork BEFORE branch 0 is enqueued on `main`: a side stream that waited for `main` afterwards would wait for the whole of
    # branch 0 (in DynMMNetV2: the gate transformer), i.e. one branch followed by four instead of five in parallel — also in
    # the captured hipGraph.  The event marks the inputs' readiness only.
    fork = torch.cuda.Event()
    fork.record(main)
    if torch.cuda.is_current_stream_capturing():
        from .. import ops
        ops._CAPTURE_EVENTS.append(fork)           # an event recorded into a capture must outlive it (ops._queue_wgrad)
    first = fns[0]()                               # host order = list order (the injected-mask tests count calls)
    outs = []
    for st, f in zip(taken, fns[1:]):
        st.wait_event(fork)
        with torch.cuda.stream(st):
            outs.append(f())
    capturing = torch.cuda.is_current_stream_capturing()
    for st, o in zip(taken, outs):
        main.wait_stream(st)
        if not capturing:
            for t in (o if isinstance(o, (list, tuple)) else [o]):
                if torch.is_tensor(t):
                    t.record_stream(main)          # allocated on the side stream, consumed on `main`
    _POOL.extend(taken)
    return [first] + outs


def join_branches():
    """After a backward pass: the calling stream waits for every branch stream (their nodes ran there)."""
    from .. import ops
    ops.flush_wgrad_groups()                       # queued weight-gradient groups go out on their branch's stream
    if _ALL:
        main = torch.cuda.current_stream()
        for st in _ALL:
            main.wait_stream(st)


def encoder_layer(h, layer, heads):
    """nn.TransformerEncoderLayer.forward (post-norm): h [B, D, T].  In training mode the layer's four dropouts act where
    torch applies them: on the attention probabilities (MultiheadAttention.dropout), on the attention block's output
    (dropout1, fused into norm1's kernel), on the feed-forward hidden layer (dropout) and on the feed-forward output
    (dropout2, fused into norm2's kernel)."""
    sa = layer.self_attn
    train = layer.training
    site = getattr(layer, '_dynmm_sites', None)
    if site is None:
        site = layer._dynmm_sites = S.new_sites(4)
    p_att = float(sa.dropout) if train else 0.0
    p1, pf, p2 = ((float(m.p) if train else 0.0) for m in (layer.dropout1, layer.dropout, layer.dropout2))
    # h feeds in_proj and norm1's residual input: norm1's backward (which runs first) leaves the residual branch's gradient with
    # the link and in_proj's input-gradient epilogue adds it (no accumulation pass by autograd)
    link = ops.GradLink() if LINK_RESIDUAL else None
    qkv = S.linear_bdt(h, sa.in_proj_weight, sa.in_proj_bias, link=link)
    a = S.mha_core(qkv, heads, drop=(p_att, site, 'attn'))
    o = S.linear_bdt(a, sa.out_proj.weight, sa.out_proj.bias)
    h1 = S.layernorm_bdt(o, layer.norm1.weight, layer.norm1.bias, layer.norm1.eps, residual=h, drop=(p1, site + 1, 'dropout1'),
                         res_link=link)
    if S.ffn_fused_ok(h1, layer.linear1.weight, layer.linear1.bias, layer.linear2.weight, layer.linear2.bias):
        # linear1 -> ReLU -> dropout -> linear2 -> dropout2 -> + h1 -> norm2: two launches (csrc/seq_ffn.hip)
        return S.ffn_block(h1, layer, (pf, site + 2, 'dropout'), (p2, site + 3, 'dropout2'))
    if pf > 0:
        f = S.linear_bdt(h1, layer.linear1.weight, layer.linear1.bias, act='relu')
        f = S.dropout_bdt(f, pf, site + 2, 'dropout')
        f = S.linear_bdt(f, layer.linear2.weight, layer.linear2.bias)
    else:
        # the ReLU backward of linear1 is applied in the input-gradient epilogue of linear2 (its only consumer)
        f = S.linear_bdt(h1, layer.linear1.weight, layer.linear1.bias, act='relu', defer_mask=True)
        f = S.linear_bdt(f, layer.linear2.weight, layer.linear2.bias, mask_input=True)
    return S.layernorm_bdt(f, layer.norm2.weight, layer.norm2.bias, layer.norm2.eps, residual=h1, drop=(p2, site + 3, 'dropout2'))


class Transformer(nn.Module):
    def __init__(self, n_features, dim, nhead=5, num_layers=5, dim_feedforward=2048):
        super().__init__()
        self.embed_dim, self.nhead = dim, nhead
        self.conv = nn.Conv1d(n_features, dim, kernel_size=1, padding=0, bias=False)
        layer = nn.TransformerEncoderLayer(d_model=dim, nhead=nhead, dim_feedforward=dim_feedforward)
        self.transformer = nn.TransformerEncoder(layer, num_layers=num_layers, enable_nested_tensor=False)

    def forward(self, x):
        if isinstance(x, (list, tuple)):
            x = x[0]                                            # [x, lengths]: the lengths are ignored
        h = S.linear_bdt(x.permute(0, 2, 1).contiguous(), self.conv.weight)      # [B, dim, T]
        for layer in self.transformer.layers:
            h = encoder_layer(h, layer, self.nhead)
        return h[:, :, -1].contiguous()                         # `self.transformer(x)[-1]`: the last time step


class MLP(nn.Module):
    def __init__(self, indim, hiddim, outdim):
        super().__init__()
        self.fc = nn.Linear(indim, hiddim)
        self.fc2 = nn.Linear(hiddim, outdim)

    def forward(self, x):
        return S.linear_bdt(S.linear_bdt(x, self.fc.weight, self.fc.bias, act='relu'), self.fc2.weight, self.fc2.bias)


class Concat(nn.Module):
    def forward(self, modalities):
        return torch.cat([m.flatten(1) for m in modalities], dim=1)


class MMDL(nn.Module):
    """Supervised_Learning.py:16-51 with has_padding=True and tensor-valued encoders."""

    def __init__(self, encoders, fusion, head, has_padding=True):
        super().__init__()
        self.encoders = nn.ModuleList(encoders)
        self.fuse, self.head, self.has_padding = fusion, head, has_padding

    def forward(self, inputs):
        return self.head(self.fuse(run_branches(self.branch_fns(inputs))))

    def branch_fns(self, inputs):
        if self.has_padding:
            return [lambda i=i, enc=enc: enc([inputs[0][i], inputs[1][i]]) for i, enc in enumerate(self.encoders)]
        return [lambda i=i, enc=enc: enc(inputs[i]) for i, enc in enumerate(self.encoders)]


def late_fusion_transformer():
    """affect_mm.py:61-66 (`--fusion 3`, saved as lf_tran.pt): the second expert of DynMMNetV2."""
    return MMDL([Transformer(35, 60), Transformer(74, 120), Transformer(300, 120)], Concat(), MLP(300, 128, 1))


class _GatedMixture(nn.Module):
    def _init_gate(self, branch_num, temp, hard_gate):
        self.branch_num = branch_num
        self.gate = nn.Sequential(Transformer(409, 10), nn.Linear(10, branch_num))       # affect_dyn.py:41,120
        self.temp, self.hard_gate = temp, hard_gate
        self.weight_list = torch.Tensor()
        self.store_weight = False
        self.infer_mode = 0

    @staticmethod
    def freeze_branch(m):
        for p in m.parameters():
            p.requires_grad = False

    def reset_weight(self):
        self.weight_list = torch.Tensor()
        self.store_weight = True

    def cal_flop(self):
        tmp = torch.mean(self.weight_list, dim=0)
        total = (self.flop * tmp).sum()
        print(f'Total Flops {total.item():.2f}M')
        return total.item()

    def gate_logits(self, inputs):
        x = torch.cat(inputs[0], dim=2)                                                   # [B, T, 409]
        return S.linear_bdt(self.gate[0]([x, inputs[1][0]]), self.gate[1].weight, self.gate[1].bias)

    def _mix(self, logits, preds):
        # affect_dyn.py:152-165: the gate's own DiffSoftmax weight is computed and RECORDED first, whatever infer_mode then
        # does with the prediction (cal_flop / weight_stat read weight_list after every evaluation mode)
        out, aux, weight = S.moe_blend(logits, preds, self.temp, self.hard_gate)
        if self.store_weight:
            self.weight_list = torch.cat((self.weight_list, weight.detach().cpu()))
        if self.infer_mode > 0:
            return preds[self.infer_mode - 1], 0
        if self.infer_mode == -1:                       # uniform weights (affect_dyn.py:161-162)
            out, aux, _ = S.moe_blend(torch.zeros_like(logits), preds, self.temp, False)
        return out, aux


class DynMMNetV2(_GatedMixture):
    """affect_dyn.py:107-175.  The reference loads pickled experts (`torch.load(model_name_list[i])`); here they are
    constructed (random init) and filled with load_state_dict."""

    def __init__(self, temp=1.0, hard_gate=False, freeze=False, model_name_list=None):
        super().__init__()
        self.text_encoder = Transformer(300, 120)            # affect_uni.py:68-73 (`--enc transformer`, text)
        self.text_head = MLP(120, 64, 1)
        self.branch2 = late_fusion_transformer()
        if model_name_list:
            raise NotImplementedError('pickled MultiBench modules cannot be loaded without MultiBench; export their '
                                      'state_dict and use load_state_dict')
        if freeze:
            for m in (self.text_encoder, self.text_head, self.branch2):
                self.freeze_branch(m)
        self._init_gate(2, temp, hard_gate)
        self.flop = torch.Tensor([135.13226, 320.03205])     # affect_dyn.py:126

    def experts(self, inputs):
        return [self.text_head(self.text_encoder([inputs[0][2], inputs[1][2]])), self.branch2(inputs)]

    def gate_and_experts(self, inputs):
        """(gate logits, [expert predictions]) with the five transformers side by side."""
        b2 = self.branch2
        enc = run_branches([lambda: self.gate_logits(inputs), lambda: self.text_encoder([inputs[0][2], inputs[1][2]])]
                           + b2.branch_fns(inputs))
        return enc[0], [self.text_head(enc[1]), b2.head(b2.fuse(enc[2:]))]

    def forward(self, inputs):
        return self._mix(*self.gate_and_experts(inputs))

    def weight_stat(self):
        tmp = torch.mean(self.weight_list, dim=0)
        print(f'mean branch weight {tmp[0].item():.4f}, {tmp[1].item():.4f}')
        self.store_weight = False
        return tmp[1].item()


class DynMMNet(_GatedMixture):
    """affect_dyn.py:31-104 (`forward2`): three uni-modal experts (visual, audio, text), 3-way gate."""

    def __init__(self, temp=1.0, hard_gate=False, freeze=True):
        super().__init__()
        self.encoders = nn.ModuleList([Transformer(FEATURES[m], 120) for m in ('visual', 'audio', 'text')])
        self.heads = nn.ModuleList([MLP(120, 64, 1) for _ in range(3)])
        if freeze:
            self.freeze_branch(self.encoders)
            self.freeze_branch(self.heads)
        self._init_gate(3, temp, hard_gate)

    def experts(self, inputs):
        return [self.heads[i](self.encoders[i]([inputs[0][i], inputs[1][i]])) for i in range(3)]

    def gate_and_experts(self, inputs):
        outs = run_branches([lambda: self.gate_logits(inputs)]
                            + [lambda i=i: self.heads[i](self.encoders[i]([inputs[0][i], inputs[1][i]])) for i in range(3)])
        return outs[0], outs[1:]

    def forward(self, inputs):
        return self._mix(*self.gate_and_experts(inputs))


class AffectTrainStep:
    """One iteration of Supervised_Learning.train's loop (:104-144) for a DynMM mixture (`moe_model`,
    additional_loss=True): forward, L1 objective + lossw * gate regulariser, backward, clip_grad_norm_(8), AdamW —
    flat parameter / gradient / moment buffers, loss + backward seeds + clip coefficient computed on the device."""

    def __init__(self, model, lr=1e-6, weight_decay=1e-4, lossw=0.0, clip_val=8.0, use_graph=False):
        from .. import engine
        self.model = model
        params = [p for p in model.parameters() if p.requires_grad]
        self.flatp = engine.FlatParameters(params, align=4)      # 16-byte aligned weights for the fused feed-forward kernel
        self.flat_g = torch.zeros_like(self.flatp.flat)
        for p in params:
            lo, hi = self.flatp.span[id(p)]
            p.grad = self.flat_g[lo:hi].view_as(p)
        self.opt = engine.Adam(self.flatp, self.flat_g, lr, weight_decay=weight_decay, decoupled=True)   # AdamW
        self.lossw, self.clip_val = float(lossw), float(clip_val)
        self.last = None
        depth = max([len(m.layers) for m in model.modules() if isinstance(m, nn.TransformerEncoder)] or [1])
        self.wgrad_group = min(8, depth)                       # (attribute: A/B against the library's default; 8 = its group limit)
        # The step is ~700 small launches (5-layer transformers on 50-token sequences): launch-bound when issued
        # eagerly, so it can be replayed as ONE hipGraph (lr / step counter are device scalars; temp, hard_gate and
        # the batch shape are frozen into a capture, which is re-made when they change).
        self.use_graph = bool(use_graph)
        self._graphs = {}
        # every Linear / Conv1d weight of the step re-laid for the MFMA kernels by ONE launch (ops.PackedWeights) instead of
        # one pack launch per layer call (r2 profile: 2 310 of 14 525 dispatches were pack_weight_kernel)
        from .. import ops
        self.prepack = ops.PackedWeights()

    def _body(self, inputs, target):
        from .. import engine, ops
        m = self.model
        self.flat_g.zero_()
        S.advance_dropout_step(self.flat_g.device)   # new dropout masks every step (also under hipGraph replay)
        prev, ops.PREPACK = ops.PREPACK, self.prepack
        # a transformer is num_layers same-shape layers on one stream: its linear1 / linear2 / in_proj / out_proj weight gradients go
        # out as ONE grouped launch each (the library's default group of 4 left every fifth layer to a launch of its own, split
        # 16 ways over the pixel range to fill the chip)
        prev_group, ops.WGRAD_GROUP = ops.WGRAD_GROUP, max(ops.WGRAD_GROUP, self.wgrad_group)
        self.prepack.pack()
        try:
            with engine.direct_gradients(False):     # kernels write parameter gradients straight into flat_g
                ops.touched_reset()
                logits, preds = m.gate_and_experts(inputs)
                self.last = S.moe_loss_backward(logits, preds, target, m.temp, m.hard_gate, self.lossw)
                join_branches()
        finally:
            self.prepack.invalidate()                # the optimizer below rewrites the weights
            ops.PREPACK = prev
            ops.WGRAD_GROUP = prev_group
        nc = S.clip_grad_norm(self.flat_g, self.clip_val)
        self.opt.grad_scale_dev = nc[1:2]
        self.opt.step(None, self.last['total'])
        self.last['grad_norm'] = nc[0:1]

    def __call__(self, inputs, target):
        if not self.use_graph:
            self._body(inputs, target)
            return self.last
        m = self.model
        key = (tuple(tuple(x.shape) for x in inputs[0]), float(m.temp), bool(m.hard_gate), bool(m.training))
        entry = self._graphs.get(key)
        if entry is None:
            static_in = [[x.clone() for x in inputs[0]], inputs[1]]
            static_y = target.clone()
            snap = [self.flatp.flat.clone()] + [t.clone() for t in self.opt.state_tensors()]
            self._body(static_in, static_y)                  # warm-up outside capture (allocator, lazy init)
            self.flatp.flat.copy_(snap[0])                   # undo the warm-up's parameter update
            for t, c in zip(self.opt.state_tensors(), snap[1:]):
                t.copy_(c)
            if self.prepack.reg and self.prepack.dirty:
                self.prepack._layout()                       # the warm-up registered the weights: lay the arena out before capturing
            from .. import ops
            graph = torch.cuda.CUDAGraph()
            with ops.capture_scope(), torch.cuda.graph(graph):
                self._body(static_in, static_y)
            entry = (graph, static_in, static_y, self.last)
            self._graphs[key] = entry
        graph, static_in, static_y, static_last = entry
        for a, b in zip(static_in[0], inputs[0]):
            a.copy_(b)
        static_y.copy_(target)
        graph.replay()
        self.last = {k: v.clone() for k, v in static_last.items()}
        return self.last
