"""Counterpart of FusionDynMM/src/models/rgb_depth_fusion.py."""
from ...nn.fusion import SqueezeAndExciteFusionAdd, SqueezeAndExciteReweigh  # noqa: F401
