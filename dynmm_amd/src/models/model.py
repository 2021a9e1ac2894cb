"""Counterpart of FusionDynMM/src/models/model.py (shared decoder pieces)."""
from ...nn.decoder import Decoder, DecoderModule, Upsample  # noqa: F401
