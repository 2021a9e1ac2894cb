"""Counterpart of FusionDynMM/src/models/model.py: the static ESANet and the shared decoder pieces."""
from ...nn.decoder import Decoder, DecoderModule, Upsample  # noqa: F401
from ...nn.esanet import ESANet  # noqa: F401
