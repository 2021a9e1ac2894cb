"""CPU: the drop-in boundary — state_dict contract, C-ABI exports, error behaviour, reference-named
import surface.  No GPU compute is issued here."""
import os
import re

import numpy as np
import pytest
import torch

from tests import helpers as Hh

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('cfg', ['P_se', 'P_add', 'S_se', 'S_add'])
def test_state_dict_matches_reference(golden_dir, cfg):
    g = np.load(os.path.join(golden_dir, 'contract.npz'))
    sd = Hh.state_dict_template(Hh.CFGS[cfg])
    assert list(sd.keys()) == [str(k) for k in g[f'{cfg}/keys']]
    assert [','.join(map(str, v.shape)) for v in sd.values()] == [str(s) for s in g[f'{cfg}/shapes']]
    assert [str(v.dtype) for v in sd.values()] == [str(s) for s in g[f'{cfg}/dtypes']]
    if cfg == 'P_se':
        assert len(sd) == 907


def test_library_exports_every_declared_symbol():
    from dynmm_amd import lib
    header = open(os.path.join(REPO, 'include', 'dynmm_hip.h')).read()
    declared = set(re.findall(r'\b(dynmm_[a-z0-9_]+)\s*\(', header))
    declared -= {'dynmm_conv_geom', 'dynmm_dropout'}
    assert declared == set(lib.SIGNATURES), declared ^ set(lib.SIGNATURES)
    if not os.path.exists(lib.LIB_PATH):
        lib.build()
    handle = lib.load()            # getattr() on every symbol; raises if one is missing
    assert handle.dynmm_abi_version() == 4
    assert b'gfx950' in handle.dynmm_build_info()


def test_product_path_has_no_cpu_fallback():
    """Forward on CPU tensors must fail loudly, never silently compute through PyTorch or the oracle."""
    from dynmm_amd.lib import DynmmHipError
    from dynmm_amd.nn.net import SkipGateESANet
    m = SkipGateESANet(height=96, width=128).eval()
    with pytest.raises(DynmmHipError):
        with torch.no_grad():
            m(torch.randn(1, 3, 96, 128), torch.randn(1, 1, 96, 128), test=True)
    for root, _, files in os.walk(os.path.join(REPO, 'dynmm_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(root, f)).read()
                assert 'import oracle' not in src and 'from oracle' not in src, f


def test_constructor_errors_like_reference():
    from dynmm_amd.nn.net import SkipGateESANet
    with pytest.raises(NotImplementedError):
        SkipGateESANet(activation='gelu')
    with pytest.raises(NotImplementedError):
        SkipGateESANet(encoder_rgb='vgg16')
    with pytest.raises(NotImplementedError):
        SkipGateESANet(encoder_block='Foo')


def test_reference_import_surface():
    """A user of FusionDynMM/src switches `src.` -> `dynmm_amd.src.` and finds the same names."""
    from dynmm_amd.src.build_model import build_model
    from dynmm_amd.src.models.model_skip_mod_globalgate import SkipGateESANet, GlobalGate, DiffSoftmax  # noqa: F401
    from dynmm_amd.src.models.resnet import ResNet34, NonBottleneck1D, BasicBlock  # noqa: F401
    from dynmm_amd.src.models.model import Decoder, Upsample, ESANet  # noqa: F401
    from dynmm_amd.src.models.rgb_depth_fusion import SqueezeAndExciteFusionAdd  # noqa: F401
    from dynmm_amd.src.models.context_modules import get_context_module  # noqa: F401
    from dynmm_amd.src.args import ArgumentParserRGBDSegmentation
    p = ArgumentParserRGBDSegmentation()
    p.set_common_args()
    args = p.parse_args(['--dynamic', '--global-gate', '--encoder', 'resnet34', '--encoder_block', 'NonBottleneck1D',
                         '--height', '96', '--width', '128', '--decoder_channels_mode', 'constant',
                         '--nr_decoder_blocks', '3'])
    # the CLI default asks for ImageNet weights (src/args.py); none are on this machine: loud, not a random init
    with pytest.raises(FileNotFoundError):
        build_model(args, n_classes=40)
    args.pretrained_on_imagenet = False
    model, device = build_model(args, n_classes=40)
    assert type(model).__name__ == 'SkipGateESANet'
    assert len(model.state_dict()) == 907
    model.freeze()
    assert all(('gate' in n) == p_.requires_grad for n, p_ in model.named_parameters())
    args.block_rule = '111'
    with pytest.raises(AssertionError):
        build_model(args, n_classes=40)
    # --dynamic without --global-gate selects the per-stage Gumbel variant (src/build_model.py:52-91)
    from dynmm_amd.src.models.model_skip_mod import SkipESANet  # noqa: F401
    from dynmm_amd.src.models.rgb_depth_fusion import SqueezeAndExciteReweigh  # noqa: F401
    args.block_rule, args.global_gate = '2222', False
    model, _ = build_model(args, n_classes=40)
    assert type(model).__name__ == 'SkipESANet' and model.block_rule == [2, 2, 2, 2]
    # no --dynamic: the static ESANet (src/build_model.py:93-113); the single-modality networks are out of scope, loudly
    args.dynamic = False
    model, _ = build_model(args, n_classes=40)
    assert type(model).__name__ == 'ESANet' and len(model.state_dict()) == 892
    args.modality = 'depth'
    with pytest.raises(NotImplementedError):
        build_model(args, n_classes=40)


def test_chain_contract_is_withdrawn_when_a_hook_can_see_the_intermediate_tensor():
    """nn/blocks.py `chain_ok`: block i may absorb block i - 1's BatchNorm backward only when nothing else can consume
    block i - 1's output — a forward hook on it (a feature tap) withdraws the offer."""
    import torch
    from dynmm_amd.nn import blocks as B
    a, b = B.NonBottleneck1D(64, 64), B.NonBottleneck1D(64, 64)
    assert B.chain_ok(a, b)
    h = a.register_forward_hook(lambda m, i, o: None)
    assert not B.chain_ok(a, b)
    h.remove()
    assert B.chain_ok(a, b)
    h = b.register_forward_pre_hook(lambda m, i: None)
    assert not B.chain_ok(a, b)
    h.remove()
    g = torch.nn.modules.module.register_module_forward_hook(lambda m, i, o: None)
    try:
        assert not B.chain_ok(a, b)
    finally:
        g.remove()
    assert B.chain_ok(a, b)
