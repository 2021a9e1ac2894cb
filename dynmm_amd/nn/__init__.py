"""nn.Module layer of the HIP path: parameter containers with the reference's state_dict keys whose
forward passes dispatch to dynmm_amd.ops (C-ABI kernels)."""
from .net import SkipGateESANet, GlobalGate, DiffSoftmax  # noqa: F401
