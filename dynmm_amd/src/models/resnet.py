"""Counterpart of FusionDynMM/src/models/resnet.py (ResNet-18/34/50 trunks used by the hot path)."""
from ...nn.blocks import BasicBlock, Bottleneck, NonBottleneck1D, ResNetEncoder as ResNet  # noqa: F401


def _make(name, block='BasicBlock', pretrained_on_imagenet=False, pretrained_dir=None,
          input_channels=3, activation=None):
    if pretrained_on_imagenet:
        raise NotImplementedError('offline build: load weights with load_state_dict instead')
    return ResNet(name, block if isinstance(block, str) else block.__name__, input_channels)


def ResNet18(**kw):
    return _make('resnet18', **kw)


def ResNet34(**kw):
    return _make('resnet34', **kw)


def ResNet50(**kw):
    kw.pop('block', None)                 # resnet.py:450-452: always Bottleneck
    return _make('resnet50', block='Bottleneck', **kw)
