"""Counterpart of FusionDynMM/src/models/model_skip_mod.py."""
from ...nn.net_skip import SkipESANet  # noqa: F401
