"""ctypes binding of libdynmm_hip.so (the C ABI declared in include/dynmm_hip.h).

There is deliberately NO fallback: if the library is missing or a call fails, this module raises.
The product path never routes through PyTorch ops or the CPU oracle for the work the HIP kernels do.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libdynmm_hip.so')
CSRC = os.path.join(_HERE, 'csrc')

ACT_NONE, ACT_RELU, ACT_TANH = 0, 1, 2
DYNMM_OK, DYNMM_EINVAL, DYNMM_EUNSUPPORTED, DYNMM_EWORKSPACE = 0, -1, -2, -3
ACT = {None: 0, 'none': 0, 'relu': 1, 'tanh': 2}

c_f = C.c_void_p       # device pointers travel as void* (tensor.data_ptr() or None)
c_i = C.c_int
c_fl = C.c_float
c_sz = C.c_size_t


class ConvGeom(C.Structure):
    """dynmm_conv_geom (include/dynmm_hip.h)."""
    _fields_ = [(n, C.c_int) for n in ('N', 'Ci', 'H', 'W', 'Co', 'Ho', 'Wo',
                                       'KH', 'KW', 'SH', 'SW', 'PH', 'PW', 'c_split')]


class Dropout(C.Structure):
    """dynmm_dropout (include/dynmm_hip.h)."""
    _fields_ = [('mask', C.c_void_p), ('step', C.c_void_p), ('seed', C.c_ulonglong), ('offset', C.c_ulonglong),
                ('p', C.c_float)]


_GP = C.POINTER(ConvGeom)
_DP = C.POINTER(Dropout)
_PP = C.POINTER(C.c_void_p)

# name -> (restype, argtypes); mirrors include/dynmm_hip.h one to one
SIGNATURES = {
    'dynmm_abi_version': (c_i, []),
    'dynmm_build_info': (C.c_char_p, []),
    'dynmm_debug_set_igemm_v5': (c_i, [c_i]),
    'dynmm_conv2d_uses_operand_ring': (c_i, [_GP, c_i]),
    'dynmm_packed_weight_floats': (c_sz, [c_i, c_i, c_i, c_i, c_i]),
    'dynmm_pack_weight': (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_f]),
    'dynmm_pack_weight_multi': (c_i, [c_f, c_f, c_f, c_i, c_i, c_f]),
    'dynmm_pack_weight_multi_blocks': (c_i, [c_i] * 5),
    'dynmm_conv2d_fwd': (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, _GP, c_i, c_f]),
    'dynmm_conv2d_stem_fwd_stats_supported': (c_i, [_GP]),
    'dynmm_conv2d_stem_fwd_stats': (c_i, [c_f, c_f, c_f, c_f, c_f, _GP, c_f]),
    'dynmm_conv2d_dgrad': (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, _GP, c_f]),
    'dynmm_conv2d_wino_supported': (c_i, [_GP, c_i]),
    'dynmm_wino_packed_floats': (c_sz, [c_i, c_i, c_i, c_i]),
    'dynmm_wino_pack': (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_f]),
    'dynmm_wino_pack_multi_blocks': (c_i, [c_i] * 4),
    'dynmm_wino_pack_multi': (c_i, [c_f, c_f, c_f, c_i, c_i, c_f]),
    'dynmm_conv2d_wino_fwd': (c_i, [c_f, c_f, c_f, c_f, c_f, _GP, c_i, c_f]),
    'dynmm_conv2d_wino_dgrad_bnred_supported': (c_i, [_GP]),
    'dynmm_conv2d_wino_dgrad_bnred_slots': (c_i, [_GP]),
    'dynmm_conv2d_wino_dgrad_bnred': (c_i, [c_f] * 9 + [_GP, c_f]),
    'dynmm_conv2d_wino_dgrad_bnred2': (c_i, [c_f] * 9 + [_GP, c_f]),
    'dynmm_conv2d_wino_fwd_stats_supported': (c_i, [_GP]),
    'dynmm_conv2d_wino_fwd_stats_slots': (c_i, [_GP]),
    'dynmm_conv2d_wino_fwd_stats': (c_i, [c_f, c_f, c_f, c_f, c_f, c_i, _GP, c_f]),
    'dynmm_conv2d_wino_dgrad': (c_i, [c_f, c_f, c_f, c_f, c_f, _GP, c_f]),
    'dynmm_conv2d_wino2d_supported': (c_i, [_GP, c_i]),
    'dynmm_wino2d_packed_floats': (c_sz, [c_i, c_i]),
    'dynmm_wino2d_pack': (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, c_f]),
    'dynmm_wino2d_pack_multi_blocks': (c_i, [c_i, c_i, c_i]),
    'dynmm_wino2d_pack_multi': (c_i, [c_f, c_f, c_f, c_i, c_i, c_f]),
    'dynmm_conv2d_wino2d_stats_slots': (c_i, [_GP]),
    'dynmm_conv2d_wino2d_fwd': (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_i, _GP, c_i, c_f]),
    'dynmm_conv2d_wino2d_dgrad': (c_i, [c_f, c_f, c_f, c_f, c_f, _GP, c_f]),
    'dynmm_conv2d_wino43_supported': (c_i, [_GP]),
    'dynmm_wino43_packed_floats': (c_sz, [c_i, c_i, c_i, c_i]),
    'dynmm_wino43_pack': (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, c_f]),
    'dynmm_wino43_pack_multi_blocks': (c_i, [c_i] * 4),
    'dynmm_wino43_pack_multi': (c_i, [c_f, c_f, c_f, c_i, c_i, c_f]),
    'dynmm_conv2d_wino43_dgrad': (c_i, [c_f, c_f, c_f, c_f, c_f, _GP, c_f]),
    'dynmm_conv2d_wgrad_workspace_bytes': (c_sz, [_GP]),
    'dynmm_conv2d_wgrad': (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_sz, _GP, c_f]),
    'dynmm_conv2d_wgrad_groupable': (c_i, [_GP]),
    'dynmm_conv2d_wgrad_variant': (c_i, [_GP]),
    'dynmm_conv2d_wgrad_group_workspace_bytes': (c_sz, [_GP, c_i]),
    'dynmm_conv2d_wgrad_group': (c_i, [c_i, c_f, c_f, c_f, c_f, c_f, c_sz, _GP, c_f]),
    'dynmm_act_bwd_bias_workspace_bytes': (c_sz, [c_i, c_i]),
    'dynmm_act_bwd_bias': (c_i, [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_f]),
    'dynmm_bn_stats': (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, c_f]),
    'dynmm_bn_relu_bits_words': (c_sz, [c_i, c_i, c_i]),
    'dynmm_bn_apply': (c_i, [c_f] * 11 + [c_i, c_i, c_i, c_fl, c_fl, c_i, c_i, c_f, c_f]),
    'dynmm_bn_bwd_reduce': (c_i, [c_f] * 8 + [c_i, c_i, c_i, c_i, c_i, c_f, c_f]),
    'dynmm_bn_bwd_apply': (c_i, [c_f] * 12 + [c_i, c_i, c_i, c_i, c_i, c_f, c_f]),
    'dynmm_bn_fold': (c_i, [c_f] * 7 + [c_i, c_fl, c_f]),
    'dynmm_maxpool3x3s2_fwd': (c_i, [c_f, c_f, c_f] + [c_i] * 6 + [c_f]),
    'dynmm_maxpool3x3s2_bwd': (c_i, [c_f, c_f, c_f] + [c_i] * 6 + [c_f]),
    'dynmm_adaptive_avgpool_fwd': (c_i, [c_f, c_f] + [c_i] * 5 + [c_f]),
    'dynmm_adaptive_avgpool_bwd': (c_i, [c_f, c_f] + [c_i] * 5 + [c_f]),
    'dynmm_nearest_into_fwd': (c_i, [c_f, c_f] + [c_i] * 8 + [c_f]),
    'dynmm_nearest_into_bwd': (c_i, [c_f, c_f] + [c_i] * 8 + [c_f]),
    'dynmm_upsample2x_dw3x3_fwd': (c_i, [c_f] * 5 + [c_i] * 4 + [c_f]),
    'dynmm_upsample2x_dw3x3_bwd_workspace_bytes': (c_sz, [c_i, c_i]),
    'dynmm_upsample2x_dw3x3_bwd': (c_i, [c_f] * 7 + [c_i] * 4 + [c_f]),
    'dynmm_gap2_fwd': (c_i, [c_f] * 4 + [c_i, c_i, c_f]),
    'dynmm_se_coeff_fwd': (c_i, [c_f, c_f, _PP, c_f, c_i] + [c_f] * 6 + [c_i, c_i, c_i, c_f]),
    'dynmm_se_coeff_bwd_workspace_bytes': (c_sz, [c_i, c_i]),
    'dynmm_se_coeff_bwd': (c_i, [c_f] * 4 + [_PP, c_f, c_i] + [c_f] * 4 + [_PP, c_f, c_f, c_f, c_i, c_f, c_i, c_i, c_i, c_f]),
    'dynmm_axpby_fwd': (c_i, [c_f] * 5 + [c_i, c_i, c_f]),
    'dynmm_axpby_bwd_reduce': (c_i, [c_f] * 5 + [c_i, c_i, c_f]),
    'dynmm_axpby_bwd_apply': (c_i, [c_f] * 5 + [c_fl, c_f, c_f, c_i, c_i, c_f]),
    'dynmm_axpby_pool_supported': (c_i, [c_i, c_i]),
    'dynmm_axpby_pool_fwd': (c_i, [c_f] * 9 + [c_i] * 4 + [c_f]),
    'dynmm_axpby_pool_bwd_reduce': (c_i, [c_f] * 7 + [c_i] * 4 + [c_f]),
    'dynmm_stem_bn_bwd_reduce': (c_i, [c_f] * 6 + [c_fl] + [c_f] * 6 + [c_i] * 5 + [c_f]),
    'dynmm_stem_bn_bwd_apply': (c_i, [c_f] * 6 + [c_fl] + [c_f] * 9 + [c_i] * 4 + [c_f]),
    'dynmm_gap2_bnrelu_fwd': (c_i, [c_f] * 3 + [c_i] + [c_f] * 2 + [c_i] * 2 + [c_f]),
    'dynmm_bn_finalize': (c_i, [c_f] * 10 + [c_i] * 3 + [c_fl, c_fl, c_f]),
    'dynmm_axpby_pool_bwd_apply': (c_i, [c_f] * 8 + [C.c_float] + [c_f] * 2 + [c_i] * 3 + [c_f]),
    'dynmm_reweigh_fwd': (c_i, [c_f, c_f, _PP, c_f, c_i, c_f, c_i, c_f, C.c_ulonglong, C.c_ulonglong, c_fl, c_i]
                          + [c_f] * 6 + [c_i, c_i, c_f]),
    'dynmm_reweigh_bwd_workspace_bytes': (c_sz, [c_i, c_i]),
    'dynmm_reweigh_bwd': (c_i, [c_f] * 5 + [_PP, c_f, c_i, c_f, c_f, c_f, _PP, c_f, c_f, c_f, c_f, c_f, c_fl, c_i, c_i, c_f]),
    'dynmm_gate_head_fwd': (c_i, [c_f] * 8 + [c_i, c_i, c_fl, c_i, c_i, c_f]),
    'dynmm_gate_decide': (c_i, [c_f] * 5 + [c_i, c_f]),
    'dynmm_gate_head_bwd': (c_i, [c_f] * 9 + [c_i, c_i, c_fl, c_f]),
    'dynmm_ce2d_fwd': (c_i, [c_f] * 4 + [c_i, c_i, c_i, c_i, c_f]),
    'dynmm_up2ce_fwd': (c_i, [c_f] * 7 + [c_i] * 5 + [c_f]),
    'dynmm_up2ce_bwd_workspace_bytes': (c_sz, [c_i] * 4),
    'dynmm_up2ce_bwd': (c_i, [c_f] * 11 + [c_i] * 4 + [c_f]),
    'dynmm_loss_head': (c_i, [c_f, c_i, c_f, c_fl, c_fl, c_f, c_f, c_f, c_f, c_f]),
    'dynmm_ce2d_valid': (c_i, [c_f] * 4 + [c_i, c_i, c_i, c_f]),
    'dynmm_ce2d_bwd': (c_i, [c_f] * 5 + [c_i, c_i, c_i, c_f]),
    'dynmm_eval_confusion': (c_i, [c_f, c_f, c_f] + [c_i] * 6 + [c_f]),
    'dynmm_batch_gather': (c_i, [c_f, c_f, c_f, c_i, c_sz, c_f]),
    'dynmm_batch_merge': (c_i, [c_f, c_f, c_f, c_f, c_i, c_sz, c_f]),
    'dynmm_add_n': (c_i, [_PP, c_i, c_f, c_sz, c_f]),
    'dynmm_reduce_slabs': (c_i, [c_f, c_f, c_i, c_i, c_f]),
    'dynmm_opt_tick': (c_i, [c_f, c_f]),
    'dynmm_sgd_nesterov': (c_i, [c_f, c_f, c_f, c_sz, c_sz, c_f, c_fl, c_fl, c_f, c_f, c_f, c_f]),
    'dynmm_adam': (c_i, [c_f, c_f, c_f, c_f, c_sz, c_sz, c_f, c_f, c_fl, c_fl, c_f, c_f, c_i, c_f, c_f]),
    'dynmm_layernorm_fwd': (c_i, [c_f] * 7 + [c_i, c_i, c_i, c_fl, c_f]),
    'dynmm_layernorm_bwd': (c_i, [c_f] * 9 + [c_i, c_i, c_i, c_f]),
    'dynmm_moe_blend_bwd': (c_i, [c_f, c_f, c_f, _PP, c_i, c_f, c_fl, _PP, c_f, c_i, c_f]),
    'dynmm_mha_fwd': (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_f]),
    'dynmm_mha_bwd': (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_f]),
    'dynmm_dropout_apply': (c_i, [c_f, c_f, c_sz, _DP, c_f]),
    'dynmm_layernorm_drop_fwd': (c_i, [c_f] * 7 + [c_i, c_i, c_i, c_fl, _DP, c_f]),
    'dynmm_layernorm_drop_bwd': (c_i, [c_f] * 10 + [c_i, c_i, c_i, _DP, c_f]),
    'dynmm_layernorm_bwd_workspace_bytes': (c_sz, [c_i, c_i, c_i]),
    'dynmm_layernorm_drop_bwd_ws': (c_i, [c_f] * 10 + [c_i, c_i, c_i, _DP, c_f, c_sz, c_f]),
    'dynmm_layernorm_parts_fwd': (c_i, [c_f, c_i] + [c_f] * 8 + [c_i, c_i, c_i, c_fl, _DP, c_f]),
    'dynmm_ffn_supported': (c_i, [c_i] * 4),
    'dynmm_ffn_nsplit': (c_i, [c_i] * 4),
    'dynmm_ffn_fwd': (c_i, [c_f] * 6 + [c_i] * 5 + [_DP, c_f]),
    'dynmm_ffn_bwd_data': (c_i, [c_f] * 6 + [c_i] * 5 + [c_fl, c_f]),
    'dynmm_mha_drop_fwd': (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, c_i, _DP, c_f]),
    'dynmm_mha_drop_bwd': (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, _DP, c_f]),
    'dynmm_moe_head': (c_i, [c_f, _PP, c_i, c_f, c_fl, c_i, c_fl, c_f, c_f, c_f, _PP, c_f, c_i, c_f]),
    'dynmm_clip_grad_norm_workspace_bytes': (c_sz, []),
    'dynmm_clip_grad_norm': (c_i, [c_f, c_sz, c_fl, c_f, c_f, c_f]),
}

ABI_VERSION = 4
_lib = None


class DynmmHipError(RuntimeError):
    pass


def build(force=False):
    """Compile libdynmm_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
    if force and os.path.exists(LIB_PATH):
        os.remove(LIB_PATH)
    subprocess.run(['make', '-C', CSRC, '-j8'], check=True)
    if not os.path.exists(LIB_PATH):
        raise DynmmHipError(f'build did not produce {LIB_PATH}')
    return LIB_PATH


def load():
    """Load the library (once) and set ctypes prototypes.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DynmmHipError(
            f'{LIB_PATH} not found: the HIP extension is required (no CPU/PyTorch fallback). '
            'Build it with `python -c "import __graft_entry__ as g; g.build()"` or `make -C dynmm_amd/csrc`.')
    # PyTorch-ROCm ships its own libamdhip64 in torch/lib.  Import torch FIRST so that our library's
    # DT_NEEDED libamdhip64.so.N binds to that already-loaded runtime; loading ours first would put a
    # second HIP runtime (from /opt/rocm) in the process, in which torch's device context and
    # allocations do not exist (launches then fail with hipErrorNoDevice).
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    runtimes = _mapped_hip_runtimes()
    if len(runtimes) > 1:
        raise DynmmHipError(f'two HIP runtimes are mapped in this process: {sorted(runtimes)}; '
                            'import torch before anything that links libamdhip64')
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.dynmm_abi_version() != ABI_VERSION:
        raise DynmmHipError('libdynmm_hip.so ABI version mismatch')
    _lib = lib
    return lib


def _mapped_hip_runtimes():
    try:
        with open('/proc/self/maps') as f:
            return {ln.split()[-1] for ln in f if 'libamdhip64' in ln}
    except OSError:
        return set()


def check(status, what):
    if status != 0:
        if status <= -1000:
            raise DynmmHipError(f'{what}: HIP error {-(status + 1000)}')
        raise DynmmHipError(f'{what}: status {status} '
                            f'({ {-1: "EINVAL", -2: "EUNSUPPORTED", -3: "EWORKSPACE"}.get(status, "?") })')
