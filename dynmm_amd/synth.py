"""Deterministic synthetic weights and inputs.

There is no dataset, checkpoint or network on the build/GPU boxes, so every
parity test, golden fixture and bench run fills the model from this module.  The filler is keyed on
the *state_dict key string* (NumPy PCG64 seeded with [seed, crc32(key)]), so it does not depend on
module construction order and produces identical tensors for the reference model, the CPU oracle and
the HIP model as long as their state_dict keys agree (the drop-in contract).
"""
import zlib

import numpy as np
import torch


def _rng(seed, key):
    return np.random.Generator(np.random.PCG64([int(seed), zlib.crc32(key.encode())]))


def synth_tensor(key, shape, seed=0, bn_keys=()):
    """Value for one state_dict entry.  `bn_keys` = set of prefixes that are BatchNorm modules."""
    r = _rng(seed, key)
    shape = tuple(shape)
    leaf = key.rsplit('.', 1)[-1]
    prefix = key.rsplit('.', 1)[0] if '.' in key else ''
    if leaf == 'num_batches_tracked':
        return np.zeros(shape, dtype=np.int64)
    if leaf == 'running_var':
        return r.uniform(0.5, 1.5, size=shape).astype(np.float32)
    if leaf == 'running_mean':
        return (0.1 * r.standard_normal(size=shape)).astype(np.float32)
    if prefix in bn_keys:
        if leaf == 'weight':
            # the last BN of every residual branch gets a small gain so that eval-mode activations
            # (running stats ~ (0, 1), i.e. no real normalisation) stay O(1) through ~40 blocks
            lo, hi = (0.2, 0.6) if prefix.endswith('bn2') else (0.5, 1.5)
            return r.uniform(lo, hi, size=shape).astype(np.float32)
        return (0.1 * r.standard_normal(size=shape)).astype(np.float32)
    if len(shape) == 4:  # conv weight [Co, Ci/groups, kh, kw]
        fan_in = shape[1] * shape[2] * shape[3]
        return (np.sqrt(1.0 / fan_in) * r.standard_normal(size=shape)).astype(np.float32)
    # conv / linear bias
    return (0.05 * r.standard_normal(size=shape)).astype(np.float32)


def fill_state_dict(state_dict, seed=0):
    """Overwrite every entry of `state_dict` (in place, any device) with the deterministic fill."""
    bn_keys = {k.rsplit('.', 1)[0] for k in state_dict if k.endswith('running_mean')}
    with torch.no_grad():
        for key, t in state_dict.items():
            v = synth_tensor(key, t.shape, seed, bn_keys)
            t.copy_(torch.from_numpy(v).to(t.dtype))
    return state_dict


def synth_inputs(n, height, width, seed=1234, device='cpu', nyu_like=False):
    """Synthetic RGB-D batch.  Default: N(0,1) post-normalisation tensors.

    nyu_like=True restates the reference's NYUv2 preprocessing constants
    (FusionDynMM/src/preprocessing.py:186-202, datasets/nyuv2/pytorch_dataset.py:57-58) on uint8 RGB
    and uint16 millimetre depth so the value ranges match a real NYUv2 pair.
    """
    r = np.random.Generator(np.random.PCG64([int(seed), n, height, width]))
    if nyu_like:
        rgb8 = r.integers(0, 256, size=(n, 3, height, width)).astype(np.float32)
        mean = np.array([0.485, 0.456, 0.406], np.float32).reshape(1, 3, 1, 1)
        std = np.array([0.229, 0.224, 0.225], np.float32).reshape(1, 3, 1, 1)
        rgb = (rgb8 / 255.0 - mean) / std
        d16 = r.integers(500, 10000, size=(n, 1, height, width)).astype(np.float32)
        depth = (d16 - 2841.94941272766) / 1417.2594281672277
    else:
        rgb = r.standard_normal(size=(n, 3, height, width)).astype(np.float32)
        depth = r.standard_normal(size=(n, 1, height, width)).astype(np.float32)
        # per-sample gain/offset so that samples differ in their global statistics (the gate sees
        # only globally pooled features)
        rgb = rgb * r.uniform(0.4, 1.6, size=(n, 1, 1, 1)).astype(np.float32) \
            + r.normal(0, 0.7, size=(n, 1, 1, 1)).astype(np.float32)
        depth = depth * r.uniform(0.4, 1.6, size=(n, 1, 1, 1)).astype(np.float32) \
            + r.normal(0, 0.7, size=(n, 1, 1, 1)).astype(np.float32)
    return (torch.from_numpy(np.ascontiguousarray(rgb, dtype=np.float32)).to(device),
            torch.from_numpy(np.ascontiguousarray(depth, dtype=np.float32)).to(device))


def synth_labels(n, height, width, num_classes=40, seed=4321, device='cpu'):
    """Labels in [0, num_classes]; 0 = void (FusionDynMM/src/utils.py:36-38)."""
    r = np.random.Generator(np.random.PCG64([int(seed), n, height, width]))
    lab = r.integers(0, num_classes + 1, size=(n, height, width)).astype(np.int64)
    return torch.from_numpy(lab).to(device)
