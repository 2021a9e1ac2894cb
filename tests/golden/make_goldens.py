"""Generate the golden fixtures in this directory from the REFERENCE ITSELF.

Runs only in the build container, where /root/reference is mounted read-only; it is never run on
the GPU box (fixtures travel, the reference does not).  The reference is imported unmodified; the
only accommodation is a process-local `.cuda()` no-op because the reference hard-codes `.cuda()`
in SkipGateESANet.__init__/forward (FusionDynMM/src/models/model_skip_mod_globalgate.py:218-223,
265, 268) and this container has no GPU.

    python tests/golden/make_goldens.py          # rewrites tests/golden/*.npz

What is stored (data only: inputs are regenerated from dynmm_amd.synth on both sides):
  model_<cfg>_<HxW>.npz   per mode: strided logits, per-(n,class) sums, gate weight, flop loss;
                           train modes add side outputs, per-parameter gradient norms, a few
                           full small gradients and BN running-stat checksums.
  nyu8_P_se.npz            BASELINE config[0] restated on 8 synthetic NYUv2-like pairs, 480x640,
                           eval --baseline: strided logits, argmax histogram, CM and mIoU.
  ops.npz                  DiffSoftmax cases, Upsample fixed init, CE loss, temperature schedule.
  valid_loss.npz           validate()'s weighted / unweighted validation losses from the reference's own classes.
  esanet_P_se_96x128.npz   the STATIC model (src/models/model.py:19-241, what build_model returns without --dynamic):
                           state_dict contract, eval logits, train outputs + gradient norms / samples.
  skip_P_96x128.npz        SkipESANet (per-stage Gumbel gates, model_skip_mod.py): per mode the Exp(1) draws
                           injected into F.gumbel_softmax (Tensor.exponential_ patched process-locally),
                           the four gate weights, strided logits and, for train modes, gradient norms.
"""
import os
import sys
import warnings

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, '/root/reference/FusionDynMM')
torch.Tensor.cuda = lambda self, *a, **k: self            # no GPU here
warnings.filterwarnings('ignore')

from src.models.model_skip_mod_globalgate import SkipGateESANet, DiffSoftmax  # noqa: E402
from src.models.model_skip_mod import SkipESANet                               # noqa: E402
from src.models.model import Upsample, ESANet                                  # noqa: E402
from src import utils as ref_utils                                             # noqa: E402

from dynmm_amd import synth                                                    # noqa: E402

torch.set_num_threads(8)
torch.manual_seed(0)

CFGS = {
    'P_se': dict(encoder_block='NonBottleneck1D', fuse_depth_in_rgb_encoder='SE-add'),
    'P_add': dict(encoder_block='NonBottleneck1D', fuse_depth_in_rgb_encoder='add'),
    'S_se': dict(encoder_block='BasicBlock', fuse_depth_in_rgb_encoder='SE-add'),
    'S_add': dict(encoder_block='BasicBlock', fuse_depth_in_rgb_encoder='add'),
    # the reference constructor's defaults: ResNet-18 encoders, BasicBlock, one decoder block per module
    'R18_se': dict(encoder_block='BasicBlock', fuse_depth_in_rgb_encoder='SE-add', encoder='resnet18',
                   nr_decoder_blocks=[1, 1, 1]),
}
# the reference CLI's default encoder (src/args.py:105): ResNet-50 / Bottleneck (encoder_block is ignored)
CFGS['R50_se'] = dict(encoder_block='BasicBlock', fuse_depth_in_rgb_encoder='SE-add', encoder='resnet50')
MODES = ['eval_baseline', 'eval_soft', 'eval_hard', 'eval_ini', 'train_soft', 'train_hard']
STRIDE = 8


def build(cfg, h, w):
    kw = dict(CFGS[cfg])
    enc = kw.pop('encoder', 'resnet34')
    nb = kw.pop('nr_decoder_blocks', [3, 3, 3])
    m = SkipGateESANet(height=h, width=w, num_classes=40, encoder_rgb=enc,
                       encoder_depth=enc, channels_decoder=[128, 128, 128],
                       nr_decoder_blocks=nb, pretrained_on_imagenet=False, **kw)
    synth.fill_state_dict(m.state_dict(), seed=0)
    return m


def ini_index(n):
    return torch.tensor([(3 * i + 1) % 5 for i in range(n)])


def grad_probe(shape, tag):
    r = np.random.Generator(np.random.PCG64([99, sum(shape), len(tag)]))
    return torch.from_numpy(r.standard_normal(size=shape).astype(np.float32))


def train_loss(outs, loss_flop):
    """Scalar used for the backward goldens: sum_s mean(out_s * G_s) + 3 * flop_loss."""
    total = 3.0 * loss_flop
    for i, o in enumerate(outs):
        total = total + (o * grad_probe(tuple(o.shape), f's{i}')).mean()
    return total


def summarize_logits(out):
    o = out.detach()
    return dict(strided=o[:, :, ::STRIDE, ::STRIDE].contiguous().numpy(),
                csum=o.sum(dim=(2, 3)).numpy(), cabs=o.abs().sum(dim=(2, 3)).numpy())


def run_mode(cfg, h, w, n, mode):
    m = build(cfg, h, w)                      # fresh weights / running stats for every mode
    rgb, depth = synth.synth_inputs(n, h, w, seed=1234)
    res = {}
    training = mode.startswith('train')
    m.train() if training else m.eval()
    m.baseline = mode == 'eval_baseline'
    m.ini_stage = mode == 'eval_ini'
    m.hard_gate = mode in ('eval_hard', 'train_hard')
    m.temp = 0.5 if mode == 'train_hard' else 1.0
    if m.ini_stage:
        real_randint = torch.randint
        torch.randint = lambda *a, **k: ini_index(n)
    try:
        if training:
            outs, lf = m(rgb, depth)
            loss = train_loss(outs, lf)
            loss.backward()
            for k, v in summarize_logits(outs[0]).items():
                res[k] = v
            for i, o in enumerate(outs[1:]):
                res[f'side{i}'] = o.detach().numpy()
            res['loss_flop'] = np.float32(lf.item())
            res['loss'] = np.float32(loss.item())
            names, norms = [], []
            for name, p in m.named_parameters():
                names.append(name)
                norms.append(0.0 if p.grad is None else p.grad.norm().item())
            res['grad_names'] = np.array(names)
            res['grad_norms'] = np.array(norms, np.float64)
            for name in ('gate_layer.fc.weight', 'gate_layer.conv.0.bias', 'encoder_depth.conv1.weight',
                         'decoder.upsample2.conv.weight', 'decoder.conv_out.bias',
                         'encoder_rgb.layer2.0.downsample.0.weight'):
                res['grad:' + name] = dict(m.named_parameters())[name].grad.numpy()
            sd = m.state_dict()
            for name in ('encoder_rgb.bn1', 'encoder_depth.layer3.2.bn2', 'gate_layer.conv.4',
                         'context_module.features.0.1.bn', 'decoder.decoder_module_3.conv3x3.bn'):
                if name + '.running_mean' not in sd:          # e.g. layer3.2 does not exist in ResNet-18
                    continue
                res['rm:' + name] = sd[name + '.running_mean'].numpy().copy()
                res['rv:' + name] = sd[name + '.running_var'].numpy().copy()
        else:
            with torch.no_grad():
                out, weight = m(rgb, depth, test=True, return_weight=True)
                _, lf = m(rgb, depth)
            for k, v in summarize_logits(out).items():
                res[k] = v
            res['weight'] = weight.numpy()
            res['loss_flop'] = np.float32(lf.item())
    finally:
        if m.ini_stage:
            torch.randint = real_randint
    return res


def train_weight(cfg, h, w, n, mode):
    """Gate weights of the train modes (the reference does not return them from forward)."""
    m = build(cfg, h, w)
    m.train()
    m.hard_gate = mode == 'train_hard'
    m.temp = 0.5 if mode == 'train_hard' else 1.0
    rgb, depth = synth.synth_inputs(n, h, w, seed=1234)
    with torch.no_grad():
        outs, weight = m(rgb, depth, test=True, return_weight=True)
    return weight.numpy()


def model_fixture(cfg, h, w, n, modes):
    blob = {}
    for mode in modes:
        print(f'  {cfg} {h}x{w} n={n} {mode}', flush=True)
        for k, v in run_mode(cfg, h, w, n, mode).items():
            blob[f'{mode}/{k}'] = v
        if mode.startswith('train'):
            blob[f'{mode}/weight'] = train_weight(cfg, h, w, n, mode)
    blob['meta'] = np.array([h, w, n, STRIDE])
    np.savez_compressed(os.path.join(HERE, f'model_{cfg}_{h}x{w}.npz'), **blob)


def nyu8_fixture():
    """BASELINE.json configs[0]: eval --baseline on 8 (synthetic) NYUv2 RGB-D pairs."""
    h, w, n = 480, 640, 8
    m = build('P_se', h, w)
    m.eval()
    m.baseline = True
    rgb, depth = synth.synth_inputs(n, h, w, seed=77, nyu_like=True)
    label = synth.synth_labels(n, h, w, seed=78)
    with torch.no_grad():
        out = m(rgb, depth, test=True)
    pred = out.argmax(1)
    mask = label > 0
    lab, prd = label[mask] - 1, pred[mask]
    cm = torch.bincount(40 * lab + prd, minlength=1600).reshape(40, 40)
    cmd = cm.double()
    iou = cmd.diag() / (cmd.sum(1) + cmd.sum(0) - cmd.diag() + 1e-15)
    hist = torch.stack([torch.bincount(pred[i].flatten(), minlength=40) for i in range(n)])
    np.savez_compressed(os.path.join(HERE, 'nyu8_P_se.npz'),
                        strided=out[:, :, ::32, ::32].contiguous().numpy(),
                        csum=out.sum(dim=(2, 3)).numpy(), cabs=out.abs().sum(dim=(2, 3)).numpy(),
                        hist=hist.numpy(), cm=cm.numpy(), miou=np.float64(iou.mean().item()))


def ops_fixture():
    blob = {}
    r = np.random.Generator(np.random.PCG64(5))
    logits = torch.from_numpy(r.standard_normal(size=(6, 5, 1, 1)).astype(np.float32))
    blob['ds/logits'] = logits.numpy()
    for tau in (1.0, 0.1, 0.001):
        for hard in (False, True):
            blob[f'ds/{tau}/{int(hard)}'] = DiffSoftmax(logits, tau=tau, hard=hard, dim=1).numpy()
    up = Upsample(mode='learned-3x3-zeropad', channels=3)
    blob['up/weight'] = up.conv.weight.detach().numpy()
    blob['up/bias'] = up.conv.bias.detach().numpy()
    x = torch.from_numpy(r.standard_normal(size=(2, 3, 4, 5)).astype(np.float32))
    blob['up/x'] = x.numpy()
    blob['up/y'] = up(x).detach().numpy()
    # weighted multi-scale CE (src/utils.py:18-50)
    cw = r.uniform(0.5, 2.0, size=40).astype(np.float32)
    ce = ref_utils.CrossEntropyLoss2d(torch.device('cpu'), cw)
    xs = [torch.from_numpy(r.standard_normal(size=(2, 40, s, s + 2)).astype(np.float32)) for s in (12, 6)]
    ts = [torch.from_numpy(r.integers(0, 41, size=(2, s, s + 2)).astype(np.float32)) for s in (12, 6)]
    losses = ce(xs, ts)
    blob['ce/weight'] = cw
    for i in range(2):
        blob[f'ce/x{i}'] = xs[i].numpy()
        blob[f'ce/t{i}'] = ts[i].numpy()
        blob[f'ce/loss{i}'] = np.float32(losses[i].item())
    sched = ref_utils.ExpDecayTemp(1.0, 0.001, 300)
    blob['temp/epochs'] = np.array([0, 1, 50, 299, 300, 400])
    blob['temp/values'] = np.array([sched.get_t(int(e)) for e in blob['temp/epochs']])
    np.savez_compressed(os.path.join(HERE, 'ops.npz'), **blob)


def valid_loss_fixture():
    """validate()'s two losses exactly as train.py:104-115 builds them and :432-440 feeds them: the reference's own
    CrossEntropyLoss2dForValidData (weighted_pixel_sum = sum_c pixels_c * w_c over the validation labels) and
    CrossEntropyLoss2dForValidDataUnweighted (src/utils.py:53-97), two batches added, then compute_whole_loss()."""
    r = np.random.Generator(np.random.PCG64(11))
    cw = r.uniform(0.5, 2.0, size=40).astype(np.float32)
    xs = [torch.from_numpy(r.standard_normal(size=(3, 40, 12, 14)).astype(np.float32) * 2) for _ in range(2)]
    ts = [torch.from_numpy(r.integers(0, 41, size=(3, 12, 14)).astype(np.uint8)) for _ in range(2)]
    pixels = np.zeros(40)
    for t in ts:
        pixels += np.bincount(t.numpy().reshape(-1), minlength=41)[1:]          # dataset.compute_class_weights('linear')
    wps = np.sum(pixels * cw)
    lw = ref_utils.CrossEntropyLoss2dForValidData(torch.device('cpu'), cw, wps)
    lu = ref_utils.CrossEntropyLoss2dForValidDataUnweighted(torch.device('cpu'))
    lw.reset_loss()
    lu.reset_loss()
    for x, t in zip(xs, ts):
        lw.add_loss_of_batch(x, t.long())
        lu.add_loss_of_batch(x, t.long())
    blob = {'weight': cw, 'weighted_pixel_sum': np.float64(wps),
            'loss_weighted': np.float64(lw.compute_whole_loss()), 'loss_unweighted': np.float64(lu.compute_whole_loss())}
    for i in range(2):
        blob[f'x{i}'], blob[f't{i}'] = xs[i].numpy(), ts[i].numpy()
    np.savez_compressed(os.path.join(HERE, 'valid_loss.npz'), **blob)


def train_steps_fixture():
    """two deterministic optimisation steps exactly as train.py:289-324 runs them —
    reference model + reference CrossEntropyLoss2d (src/utils.py:18-50) + torch SGD-Nesterov
    (train.py:557-563) with the total-loss rule of train.py:313-321 (loss_ratio, flop_budget)."""
    h, w, n = 96, 128, 2
    m = build('P_se', h, w)
    m.train()
    m.temp, m.hard_gate = 0.8, False
    rgb, depth = synth.synth_inputs(n, h, w, seed=1234)
    labels = [synth.synth_labels(n, h // s, w // s, seed=300 + s).float() for s in (1, 8, 16, 32)]
    cw = np.linspace(0.5, 2.0, 40).astype(np.float32)
    ce = ref_utils.CrossEntropyLoss2d(torch.device('cpu'), cw)
    opt = torch.optim.SGD(m.parameters(), lr=0.002, weight_decay=1e-4, momentum=0.9, nesterov=True)
    ratio, budget = 0.5, 1.0
    blob = {'meta': np.array([h, w, n]), 'cw': cw, 'hyper': np.array([0.002, 1e-4, 0.9, ratio, budget, 0.8])}
    for step in range(2):
        m.start_weight()
        opt.zero_grad()
        outs, lf = m(rgb, depth)
        losses = ce(outs, labels)
        total = sum(losses) + ratio * max(torch.zeros_like(lf), lf - budget)
        total.backward()
        opt.step()
        blob[f'step{step}/losses'] = np.array([l.item() for l in losses], np.float64)
        blob[f'step{step}/loss_flop'] = np.float64(lf.item())
        blob[f'step{step}/total'] = np.float64(total.item())
        blob[f'step{step}/weight'] = m.weight_list.detach().numpy().copy()
        m.end_weight()
    sd = m.state_dict()
    blob['param_norms'] = np.array([sd[k].double().norm().item() for k in sd if sd[k].dtype.is_floating_point])
    blob['param_names'] = np.array([k for k in sd if sd[k].dtype.is_floating_point])
    np.savez_compressed(os.path.join(HERE, 'train_steps_P_se.npz'), **blob)


def grad_sample(t, limit=128):
    """Deterministic sub-sample of a gradient tensor (whole tensor when it has <= `limit` elements)."""
    f = t.detach().reshape(-1)
    step = max(1, -(-f.numel() // limit))
    return f[::step]


def train_n8_fixture():
    """Train-mode parity pin at a better-conditioned batch: config P, SE-add, 160x192,
    N = 8 (PyramidPooling's 1x1 branch normalises over 8 values instead of 2-4), soft gates tau = 1, the
    reference's own weighted 4-scale CrossEntropyLoss2d (src/utils.py:18-50) and the total-loss rule of
    train.py:313-321.  The reference is run twice — float32 and float64 (`model.double()`) — so that the
    consumer can state its gradient error against the fp64 truth next to the reference's own fp32 error."""
    h, w, n = 160, 192, 8
    cw = np.linspace(0.5, 2.0, 40).astype(np.float32)
    ratio = 0.5
    blob = {'meta': np.array([h, w, n, 16]), 'cw': cw, 'ratio': np.float32(ratio)}
    for tag, dt in (('f32', torch.float32), ('f64', torch.float64)):
        m = build('P_se', h, w)
        m.train()
        m.temp, m.hard_gate = 1.0, False
        m.flop = m.flop.to(dt)
        m.depth_enc_flop = m.depth_enc_flop.to(dt)
        m.total_flop = m.total_flop.to(dt)
        m = m.to(dt)
        rgb, depth = synth.synth_inputs(n, h, w, seed=1234)
        labels = [synth.synth_labels(n, h // s_, w // s_, seed=300 + s_).float() for s_ in (1, 8, 16, 32)]
        ce = ref_utils.CrossEntropyLoss2d(torch.device('cpu'), cw)
        if dt == torch.float64:
            ce.ce_loss = torch.nn.CrossEntropyLoss(torch.from_numpy(cw).double(), reduction='none', ignore_index=-1)
        outs, lf = m(rgb.to(dt), depth.to(dt))
        losses = ce(outs, labels)
        total = sum(losses) + ratio * max(torch.zeros_like(lf), lf)
        total.backward()
        blob[f'{tag}/losses'] = np.array([l.item() for l in losses], np.float64)
        blob[f'{tag}/loss_flop'] = np.float64(lf.item())
        blob[f'{tag}/total'] = np.float64(total.item())
        if dt == torch.float32:                 # outputs: fp32 run only (they agree with fp64 to ~1e-5), sub-sampled
            o = outs[0].detach()
            blob['out/strided'] = o[:, :, ::16, ::16].contiguous().numpy()
            blob['out/csum'] = o.sum(dim=(2, 3)).numpy()
            blob['out/cabs'] = o.abs().sum(dim=(2, 3)).numpy()
            for i, (o, st) in enumerate(zip(outs[1:], (4, 2, 1))):
                blob[f'out/side{i}'] = o.detach()[:, :, ::st, ::st].contiguous().numpy()
        names, norms = [], []
        for name, p in m.named_parameters():
            names.append(name)
            g = p.grad
            norms.append(0.0 if g is None else g.double().norm().item())
            if g is not None:
                blob[f'{tag}/g:{name}'] = grad_sample(g).float().numpy()
        blob['grad_names'] = np.array(names)
        blob[f'{tag}/grad_norms'] = np.array(norms, np.float64)
        sd = m.state_dict()
        for name in ('encoder_rgb.bn1', 'encoder_depth.layer3.2.bn2', 'gate_layer.conv.4',
                     'context_module.features.0.1.bn', 'decoder.decoder_module_3.conv3x3.bn'):
            blob[f'{tag}/rm:{name}'] = sd[name + '.running_mean'].float().numpy().copy()
            blob[f'{tag}/rv:{name}'] = sd[name + '.running_var'].float().numpy().copy()
    np.savez_compressed(os.path.join(HERE, 'train_n8_P_se_160x192.npz'), **blob)


SKIP_MODES = {
    # name: (training, test, hard_gate, temp, block_rule)
    'eval_test': (False, True, False, 1.0, [2, 2, 2, 2]),
    'eval_soft': (False, False, False, 1.0, [2, 2, 2, 2]),
    'eval_mixed_rule': (False, False, False, 0.7, [2, 1, 0, 2]),
    'train_soft': (True, False, False, 0.7, [2, 2, 2, 2]),
    'train_hard': (True, False, True, 0.5, [2, 2, 2, 2]),
}


def skip_noise(n, mode):
    r = np.random.Generator(np.random.PCG64([4242, n, len(mode)]))
    return [torch.from_numpy(r.exponential(size=(n, 2)).astype(np.float32)) for _ in range(4)]


def skip_fixture():
    """SkipESANet (model_skip_mod.py:20-311) with the Gumbel draws pinned: F.gumbel_softmax calls
    `empty_like(logits).exponential_()`; that method is replaced for the duration of the forward by one
    that hands out the recorded Exp(1) samples in call order."""
    h, w, n = 96, 128, 3
    blob = {'meta': np.array([h, w, n, STRIDE])}
    real_exp = torch.Tensor.exponential_
    for mode, (training, test, hard, temp, rule) in SKIP_MODES.items():
        print(f'  skip {mode}', flush=True)
        m = SkipESANet(height=h, width=w, num_classes=40, encoder_rgb='resnet34', encoder_depth='resnet34',
                       encoder_block='NonBottleneck1D', channels_decoder=[128, 128, 128],
                       nr_decoder_blocks=[3, 3, 3], pretrained_on_imagenet=False,
                       fuse_depth_in_rgb_encoder='SE-add', upsampling='learned-3x3-zeropad', temp=temp,
                       block_rule=rule)
        synth.fill_state_dict(m.state_dict(), seed=0)
        m.train() if training else m.eval()
        m.hard_gate = hard
        rgb, depth = synth.synth_inputs(n, h, w, seed=1234)
        noise = skip_noise(n, mode)
        calls = [0]

        def fake_exponential(self, *a, **k):
            self.copy_(noise[calls[0]])
            calls[0] += 1
            return self
        torch.Tensor.exponential_ = fake_exponential
        try:
            m.start_weight()
            if training:
                outs = m(rgb, depth, test=test)
                loss = train_loss(outs, torch.zeros(()))
                loss.backward()
            else:
                with torch.no_grad():
                    outs = m(rgb, depth, test=test)
        finally:
            torch.Tensor.exponential_ = real_exp
        assert calls[0] == 4
        out = outs[0] if training else outs
        for k, v in summarize_logits(out).items():
            blob[f'{mode}/{k}'] = v
        for j in range(4):
            blob[f'{mode}/noise{j}'] = noise[j].numpy()
            blob[f'{mode}/weight{j}'] = m.weight_list[j].detach().numpy().copy()
        blob[f'{mode}/cfg'] = np.array([int(training), int(test), int(hard)] + list(rule), np.int64)
        blob[f'{mode}/temp'] = np.float32(temp)
        if training:
            blob[f'{mode}/loss'] = np.float32(loss.item())
            for i, o in enumerate(outs[1:]):
                blob[f'{mode}/side{i}'] = o.detach().numpy()
            names, norms = [], []
            for name, prm in m.named_parameters():
                names.append(name)
                norms.append(0.0 if prm.grad is None else prm.grad.norm().item())
            blob[f'{mode}/grad_names'] = np.array(names)
            blob[f'{mode}/grad_norms'] = np.array(norms, np.float64)
            for name in ('gate_layer0.se.fc.0.weight', 'gate_layer2.se.fc.2.bias', 'gate_layer3.se.fc.0.bias',
                         'encoder_depth.conv1.weight', 'decoder.conv_out.bias'):
                blob[f'{mode}/grad:' + name] = dict(m.named_parameters())[name].grad.numpy()
    sd = m.state_dict()
    blob['keys'] = np.array(list(sd.keys()))
    blob['shapes'] = np.array([','.join(map(str, v.shape)) for v in sd.values()])
    blob['dtypes'] = np.array([str(v.dtype) for v in sd.values()])
    np.savez_compressed(os.path.join(HERE, 'skip_P_96x128.npz'), **blob)


def esanet_fixture():
    """ESANet (src/models/model.py:19-241): the static fuse-at-every-stage network `build_model` returns when --dynamic is
    not given (src/build_model.py:93-113); config P (ResNet-34 / NonBottleneck1D / SE-add / 3 decoder blocks)."""
    h, w, n = 96, 128, 2
    blob = {'meta': np.array([h, w, n, STRIDE])}

    def make():
        m = ESANet(height=h, width=w, num_classes=40, encoder_rgb='resnet34', encoder_depth='resnet34',
                   encoder_block='NonBottleneck1D', channels_decoder=[128, 128, 128], nr_decoder_blocks=[3, 3, 3],
                   pretrained_on_imagenet=False, fuse_depth_in_rgb_encoder='SE-add', upsampling='learned-3x3-zeropad')
        synth.fill_state_dict(m.state_dict(), seed=0)
        return m
    m = make()
    sd = m.state_dict()
    blob['keys'] = np.array(list(sd.keys()))
    blob['shapes'] = np.array([','.join(map(str, v.shape)) for v in sd.values()])
    blob['dtypes'] = np.array([str(v.dtype) for v in sd.values()])
    rgb, depth = synth.synth_inputs(n, h, w, seed=1234)
    m.eval()
    with torch.no_grad():
        out = m(rgb, depth)
    for k, v in summarize_logits(out).items():
        blob[f'eval/{k}'] = v
    m = make()
    m.train()
    outs = m(rgb, depth)
    loss = train_loss(outs, torch.zeros(()))
    loss.backward()
    for k, v in summarize_logits(outs[0]).items():
        blob[f'train/{k}'] = v
    for i, o in enumerate(outs[1:]):
        blob[f'train/side{i}'] = o.detach().numpy()
    blob['train/loss'] = np.float32(loss.item())
    names = [k for k, _ in m.named_parameters()]
    blob['train/grad_names'] = np.array(names)
    blob['train/grad_norms'] = np.array([p.grad.norm().item() for _, p in m.named_parameters()], np.float64)
    for name in ('encoder_depth.conv1.weight', 'decoder.conv_out.bias', 'se_layer2.se_rgb.fc.0.weight'):
        blob['train/grad:' + name] = dict(m.named_parameters())[name].grad.numpy()
    np.savez_compressed(os.path.join(HERE, 'esanet_P_se_96x128.npz'), **blob)


def contract_fixture():
    """state_dict keys / shapes / dtypes of the reference model (the strict-load contract, eval.py:61)."""
    blob = {}
    for cfg in CFGS:
        sd = build(cfg, 96, 128).state_dict()
        blob[f'{cfg}/keys'] = np.array(list(sd.keys()))
        blob[f'{cfg}/shapes'] = np.array([','.join(map(str, v.shape)) for v in sd.values()])
        blob[f'{cfg}/dtypes'] = np.array([str(v.dtype) for v in sd.values()])
    np.savez_compressed(os.path.join(HERE, 'contract.npz'), **blob)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'contract':
        contract_fixture()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'r18':
        model_fixture('R18_se', 96, 128, 2, ['eval_hard', 'train_soft'])
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'r50':
        model_fixture('R50_se', 96, 128, 2, ['eval_hard', 'train_soft'])
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'valid':
        valid_loss_fixture()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'esanet':
        esanet_fixture()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'skip':
        skip_fixture()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'train_steps':
        train_steps_fixture()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'train_n8':
        train_n8_fixture()
        sys.exit(0)
    contract_fixture()
    train_steps_fixture()
    ops_fixture()
    model_fixture('P_se', 96, 128, 2, MODES)
    model_fixture('P_add', 96, 128, 2, ['eval_soft', 'train_soft'])
    model_fixture('S_se', 96, 128, 2, ['eval_hard', 'train_soft'])
    model_fixture('S_add', 96, 128, 2, ['eval_baseline'])
    model_fixture('P_se', 160, 192, 3, ['eval_hard', 'train_soft'])
    model_fixture('R18_se', 96, 128, 2, ['eval_hard', 'train_soft'])
    model_fixture('R50_se', 96, 128, 2, ['eval_hard', 'train_soft'])
    nyu8_fixture()
    skip_fixture()
    esanet_fixture()
    train_n8_fixture()
    print('done')
