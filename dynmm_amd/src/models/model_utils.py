"""Counterpart of FusionDynMM/src/models/model_utils.py."""
from ...nn.blocks import ConvBNAct  # noqa: F401
from ...nn.fusion import SqueezeAndExcitation  # noqa: F401
