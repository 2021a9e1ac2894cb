"""Counterpart of FusionDynMM/src/models/resnet.py (ResNet-18/34/50 trunks used by the hot path)."""
from ...nn.blocks import BasicBlock, Bottleneck, NonBottleneck1D, ResNetEncoder as ResNet  # noqa: F401
from ..pretrained import load_imagenet_encoder


def _make(name, block='BasicBlock', pretrained_on_imagenet=False, pretrained_dir='./trained_models/imagenet',
          input_channels=3, activation=None):
    block = block if isinstance(block, str) else block.__name__
    model = ResNet(name, block, input_channels)
    if pretrained_on_imagenet:             # resnet.py:395-466 / :469-509, from local files
        load_imagenet_encoder(model, name, 'Bottleneck' if name == 'resnet50' else block, input_channels, pretrained_dir)
    return model


def ResNet18(**kw):
    return _make('resnet18', **kw)


def ResNet34(**kw):
    return _make('resnet34', **kw)


def ResNet50(**kw):
    kw.pop('block', None)                 # resnet.py:450-452: always Bottleneck
    return _make('resnet50', block='Bottleneck', **kw)
