"""Counterpart of FusionDynMM/src/models/model_skip_mod_globalgate.py."""
from ...nn.net import DiffSoftmax, GlobalGate, SkipGateESANet  # noqa: F401
