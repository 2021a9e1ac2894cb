"""Counterpart of FusionDynMM/src/models/context_modules.py."""
from ...nn.context import PyramidPoolingModule, get_context_module  # noqa: F401
