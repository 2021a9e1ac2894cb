"""ESANet (the static RGB-D network of FusionDynMM/src/models/model.py:19-241 — what build_model returns without --dynamic),
weight loading from local files (src/models/resnet.py:395-509, src/build_model.py:181-205) and --last_ckpt resume
(src/utils.py:145-175, train.py:131-135).

CPU: the oracle's restatement and the build's state_dict against the fixture the REFERENCE produced
(tests/golden/make_goldens.py: esanet_fixture); the loaders against checkpoint files written in the reference's formats.
GPU: the HIP model against the same fixture; save -> resume -> the next optimisation step is bit-identical."""
import argparse
import os

import numpy as np
import pytest
import torch

from dynmm_amd import synth
from oracle import dynmm_oracle as O
from tests import helpers as Hh

CFG = Hh.CFGS['P_se']


def esanet(h=96, w=128):
    from dynmm_amd.nn.esanet import ESANet
    return ESANet(height=h, width=w, num_classes=40, encoder_rgb='resnet34', encoder_depth='resnet34',
                  encoder_block='NonBottleneck1D', channels_decoder=[128, 128, 128], nr_decoder_blocks=[3, 3, 3],
                  pretrained_on_imagenet=False, fuse_depth_in_rgb_encoder='SE-add', upsampling='learned-3x3-zeropad')


def fixture(golden_dir):
    return np.load(os.path.join(golden_dir, 'esanet_P_se_96x128.npz'))


# ------------------------------------------------------------------------------------------------ CPU
def test_esanet_state_dict_matches_reference(golden_dir):
    g = fixture(golden_dir)
    sd = esanet().state_dict()
    assert list(sd.keys()) == [str(k) for k in g['keys']]                      # 892 entries, no gate parameters
    assert [','.join(map(str, v.shape)) for v in sd.values()] == [str(s) for s in g['shapes']]
    assert [str(v.dtype) for v in sd.values()] == [str(s) for s in g['dtypes']]
    assert not any('gate' in k for k in sd)


def test_esanet_oracle_matches_reference_fixture(golden_dir):
    g = fixture(golden_dir)
    h, w, n, stride = [int(v) for v in g['meta']]
    rgb, depth = synth.synth_inputs(n, h, w, seed=1234)
    sd = {k: v.clone() for k, v in esanet().state_dict().items()}
    synth.fill_state_dict(sd, seed=0)
    with torch.no_grad():
        out = O.forward_esanet(sd, rgb, depth, CFG)
    assert Hh.rel_err(out[:, :, ::stride, ::stride], g['eval/strided']) < 2e-5
    assert Hh.rel_err(out.sum(dim=(2, 3)), g['eval/csum']) < 1e-4
    params = {k: v.requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and 'running_' not in k}
    outs = O.forward_esanet(sd, rgb, depth, CFG, training=True)
    assert isinstance(outs, tuple) and len(outs) == 4                           # model.py:306-308
    loss = Hh.train_loss(outs, torch.zeros(()))
    loss.backward()
    assert abs(loss.item() - float(g['train/loss'])) < 1e-4 * max(1.0, abs(float(g['train/loss'])))
    assert Hh.rel_err(outs[0].detach()[:, :, ::stride, ::stride], g['train/strided']) < 2e-4
    names = [str(s) for s in g['train/grad_names']]
    norms = np.array([params[k].grad.norm().item() for k in names])
    ref = g['train/grad_norms']
    assert np.all(np.abs(norms - ref) <= 0.05 * np.maximum(ref, 1e-2 * ref.max()))


def _args(**kw):
    from dynmm_amd.src.args import ArgumentParserRGBDSegmentation
    p = ArgumentParserRGBDSegmentation()
    p.set_common_args()
    a = p.parse_args(['--encoder', 'resnet34', '--encoder_block', 'NonBottleneck1D', '--decoder_channels_mode', 'constant',
                      '--height', '96', '--width', '128'])
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def test_build_model_static_and_loud_failures(tmp_path):
    from dynmm_amd.nn.esanet import ESANet
    from dynmm_amd.src.build_model import build_model
    model, _ = build_model(_args(pretrained_on_imagenet=False), 40)
    assert type(model) is ESANet and not hasattr(model, 'gate_layer')
    # accepted-but-unimplemented flags fail loudly
    with pytest.raises(NotImplementedError):
        build_model(_args(pretrained_on_imagenet=False, modality='rgb'), 40)
    with pytest.raises(FileNotFoundError):                                       # ImageNet weights asked for, none on disk
        build_model(_args(pretrained_on_imagenet=True, pretrained_dir=str(tmp_path / 'nowhere')), 40)
    with pytest.raises(FileNotFoundError):
        build_model(_args(pretrained_on_imagenet=False, pretrained_scenenet=str(tmp_path / 'missing.pth')), 40)
    with pytest.raises(NotImplementedError):
        model.freeze()


def test_imagenet_weights_from_local_files(tmp_path, monkeypatch):
    """resnet.py:395-466 (torchvision layout, conv1 summed over RGB for the depth trunk, fc dropped, strict) and :469-509
    (NonBottleneck1D checkpoints: 'encoder.'-prefixed keys of the authors' ImageNet run, strict=False)."""
    from dynmm_amd.nn.blocks import ResNetEncoder
    from dynmm_amd.src import pretrained as P
    # (a) BasicBlock trunk from a torchvision-format file in the working directory
    src = ResNetEncoder('resnet18', 'BasicBlock', 3)
    synth.fill_state_dict(src.state_dict(), seed=5)
    tv = {k: v.clone() for k, v in src.state_dict().items()}
    tv['fc.weight'], tv['fc.bias'] = torch.zeros(1000, 512), torch.zeros(1000)
    monkeypatch.chdir(tmp_path)
    torch.save(tv, tmp_path / P.TORCHVISION_FILES['resnet18'])
    rgb, dep = ResNetEncoder('resnet18', 'BasicBlock', 3), ResNetEncoder('resnet18', 'BasicBlock', 1)
    P.load_imagenet_encoder(rgb, 'resnet18', 'BasicBlock', 3, str(tmp_path / 'unused'))
    P.load_imagenet_encoder(dep, 'resnet18', 'BasicBlock', 1, str(tmp_path / 'unused'))
    assert all(torch.equal(v, src.state_dict()[k]) for k, v in rgb.state_dict().items())
    assert torch.equal(dep.conv1.weight, src.conv1.weight.sum(1, keepdim=True))
    assert torch.equal(dep.layer3[1].conv2.weight, src.layer3[1].conv2.weight)
    # (b) NonBottleneck1D trunk from <pretrained_dir>/r34_NBt1D.pth
    nb = ResNetEncoder('resnet34', 'NonBottleneck1D', 3)
    synth.fill_state_dict(nb.state_dict(), seed=6)
    ck = {'state_dict': {**{'encoder.' + k: v.clone() for k, v in nb.state_dict().items()},
                         'fc.weight': torch.zeros(1000, 512), 'fc.bias': torch.zeros(1000)}, 'epoch': 1}
    d = tmp_path / 'imagenet'
    d.mkdir()
    torch.save(ck, d / 'r34_NBt1D.pth')
    from dynmm_amd.nn.net import SkipGateESANet
    m = SkipGateESANet(height=96, width=128, encoder_block='NonBottleneck1D', fuse_depth_in_rgb_encoder='SE-add',
                       pretrained_on_imagenet=True, pretrained_dir=str(d))
    assert torch.equal(m.encoder_rgb.layer2[1].conv3x1_1.weight, nb.layer2[1].conv3x1_1.weight)
    assert torch.equal(m.encoder_depth.conv1.weight, nb.conv1.weight.sum(1, keepdim=True))
    assert torch.equal(m.encoder_depth.layer4[2].bn2.running_var, nb.layer4[2].bn2.running_var)


def test_scenenet_weights_skip_output_layers(tmp_path):
    """build_model.py:181-205: everything but the (side) outputs and the two last learned up-samplings."""
    from dynmm_amd.src.pretrained import load_scenenet
    src, dst = esanet(), esanet()
    synth.fill_state_dict(src.state_dict(), seed=9)
    before = {k: v.clone() for k, v in dst.state_dict().items()}
    path = tmp_path / 'scenenet.pth'
    torch.save({'state_dict': src.state_dict()}, path)
    load_scenenet(dst, str(path))
    for k, v in dst.state_dict().items():
        kept = 'out' in k or 'decoder.upsample1' in k or 'decoder.upsample2' in k
        assert torch.equal(v, before[k] if kept else src.state_dict()[k]), k


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_esanet_hip_matches_reference_fixture(golden_dir):
    g = fixture(golden_dir)
    h, w, n, stride = [int(v) for v in g['meta']]
    rgb, depth = synth.synth_inputs(n, h, w, seed=1234, device='cuda')
    m = esanet()
    synth.fill_state_dict(m.state_dict(), seed=0)
    m = m.cuda().eval()
    with torch.no_grad():
        out = m(rgb, depth).cpu()
    assert Hh.rel_err(out[:, :, ::stride, ::stride], g['eval/strided']) < 2e-4
    assert Hh.rel_err(out.sum(dim=(2, 3)), g['eval/csum']) < 1e-3
    m = esanet()
    synth.fill_state_dict(m.state_dict(), seed=0)
    m = m.cuda().train()
    outs = m(rgb, depth)
    assert isinstance(outs, tuple) and len(outs) == 4
    loss = Hh.train_loss(outs, torch.zeros((), device='cuda'))
    loss.backward()
    assert abs(loss.item() - float(g['train/loss'])) < 1e-3 * max(1.0, abs(float(g['train/loss'])))
    assert Hh.rel_err(outs[0].detach().cpu()[:, :, ::stride, ::stride], g['train/strided']) < 1e-3
    for i, o in enumerate(outs[1:]):
        assert Hh.rel_err(o.detach().cpu(), g[f'train/side{i}']) < 1e-3
    params = dict(m.named_parameters())
    names = [str(s) for s in g['train/grad_names']]
    norms = np.array([params[k].grad.norm().item() for k in names])
    ref = g['train/grad_norms']
    bad = np.abs(norms - ref) > 0.2 * np.maximum(ref, 1e-2 * ref.max())        # fp32 conditioning: see tests/test_hip_model.py
    assert not bad.any(), [(names[i], norms[i], ref[i]) for i in np.nonzero(bad)[0][:8]]


@pytest.mark.gpu
@pytest.mark.parametrize('optimizer', ['SGD', 'Adam'])
def test_resume_reproduces_the_next_step(tmp_path, optimizer):
    """train.py:131-135 / src/utils.py:145-175: model + optimizer state + epoch from --last_ckpt; after the resume the next
    optimisation step must be bit-identical to the uninterrupted run's (momentum / Adam moments / step counters)."""
    from dynmm_amd import engine, train
    from dynmm_amd.nn.net import SkipGateESANet
    from dynmm_amd.src.pretrained import load_ckpt
    cw = np.linspace(0.5, 2.0, 40)
    h, w, n = 96, 128, 2
    rgb, depth = synth.synth_inputs(n, h, w, seed=3, device='cuda')
    labels = [synth.synth_labels(n, h // s, w // s, seed=40 + s, device='cuda').to(torch.uint8) for s in (1, 8, 16, 32)]

    def fresh():
        m = SkipGateESANet(height=h, width=w, encoder_block='NonBottleneck1D', fuse_depth_in_rgb_encoder='SE-add')
        synth.fill_state_dict(m.state_dict(), seed=0)
        m = m.cuda().train()
        return m, engine.TrainStep(m, cw, lr=0.01, loss_ratio=0.1, optimizer=optimizer, multi_stream=False)
    m1, s1 = fresh()
    for _ in range(2):
        s1(rgb, depth, labels)
    path = train.save_ckpt(str(tmp_path), m1, s1.opt, 7, best_miou=12.5, best_miou_epoch=4)
    ck = torch.load(path, map_location='cpu')
    assert set(ck) >= {'epoch', 'state_dict', 'optimizer', 'best_miou', 'best_miou_epoch'}
    assert set(ck['optimizer']) >= {'state', 'param_groups'}                     # torch.optim's own layout
    n_train = sum(p.requires_grad for p in m1.parameters())
    assert len(ck['optimizer']['state']) == n_train
    s1(rgb, depth, labels)                                                       # the uninterrupted third step
    torch.cuda.synchronize()
    m2, s2 = fresh()
    epoch, best, best_ep = load_ckpt(m2, s2.opt, path)
    assert (epoch, best, best_ep) == (7, 12.5, 4)
    s2.flatp.refresh()
    s2(rgb, depth, labels)
    torch.cuda.synchronize()
    for (k, a), (_, b) in zip(m1.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    with pytest.raises(FileNotFoundError):
        load_ckpt(m2, s2.opt, str(tmp_path / 'missing.pth'))


@pytest.mark.gpu
def test_esanet_through_evaluate_and_the_train_driver(tmp_path):
    """build_model returns ESANet for non-dynamic runs, and both callers of a model — engine.evaluate
    (eval.py / validate: model(rgb, depth, test=True)) and the training driver with its epoch-0 validation — must accept
    it.  Also the resume protocol of train.py:207 (validation at the FIRST epoch of a resumed run) and the rule that a
    resume without a new best does not overwrite the best checkpoint's file."""
    from dynmm_amd import engine, train
    m = esanet()
    synth.fill_state_dict(m.state_dict(), seed=0)
    m = m.cuda().eval()
    rgb, depth = synth.synth_inputs(2, 96, 128, seed=5, device='cuda')
    label = synth.synth_labels(2, 96, 128, seed=6, device='cuda')
    miou, cm = engine.evaluate(m, [(rgb, depth, label)], num_classes=40)
    assert np.isfinite(miou) and int(cm.sum()) == int((label > 0).sum())
    with torch.no_grad():
        out, weight = m(rgb, depth, test=True, return_weight=True)
    assert out.shape == (2, 40, 96, 128) and torch.equal(weight.cpu()[:, 4], torch.ones(2))   # no gate: always "fuse all"
    common = ['--encoder', 'resnet34', '--encoder_block', 'NonBottleneck1D', '--decoder_channels_mode', 'constant',
              '--no_imagenet_pretraining', '--dataset', 'synthetic', '--height', '96', '--width', '128',
              '--batch_size', '4', '--synthetic_samples', '8', '--eval-every', '5', '--results_dir', str(tmp_path)]
    logs = train.train_main(common + ['--epochs', '2'])
    assert len(logs) == 2 and all(np.isfinite(r['loss_train_total']) for r in logs)
    assert 'mIoU_test_kv1' in logs[0] and 'loss_test' in logs[0] and 'loss_test_unweighted' in logs[0]
    assert 'mIoU_test' not in logs[1]                       # eval_every = 5: only the first epoch validates
