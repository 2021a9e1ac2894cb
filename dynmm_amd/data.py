"""Minimal data side for the drivers.  The reference's NYUv2 pipeline (FusionDynMM/src/datasets,
src/preprocessing.py: cv2/torchvision host code) is out of scope; the drivers
accept any iterable of dict batches with the reference's keys
    'image' [N,3,H,W] f32, 'depth' [N,1,H,W] f32, 'label' [N,H,W] (0 = void),
    'label_down' {8: ..., 16: ..., 32: ...}, optionally 'label_orig'
and ship a deterministic synthetic NYUv2-shaped source for smoke runs and benchmarks."""
import torch

from . import synth


class SyntheticRGBD:
    n_classes_without_void = 40
    cameras = ['kv1']
    split = 'test'

    def __init__(self, n_samples, batch_size, height=480, width=640, seed=0, device='cpu', nyu_like=True):
        self.n, self.bs, self.h, self.w, self.seed, self.device = n_samples, batch_size, height, width, seed, device
        self.nyu_like = nyu_like

    def __len__(self):
        return (self.n + self.bs - 1) // self.bs

    def __iter__(self):
        for i in range(len(self)):
            n = min(self.bs, self.n - i * self.bs)
            rgb, depth = synth.synth_inputs(n, self.h, self.w, seed=self.seed + 17 * i, device=self.device,
                                            nyu_like=self.nyu_like)
            label = synth.synth_labels(n, self.h, self.w, seed=self.seed + 17 * i + 1, device=self.device)
            down = {r: synth.synth_labels(n, self.h // r, self.w // r, seed=self.seed + 17 * i + r, device=self.device)
                    for r in (8, 16, 32)}
            yield {'image': rgb, 'depth': depth, 'label': label, 'label_down': down, 'label_orig': label}

    def compute_class_weights(self, weight_mode='median_frequency', c=1.02):
        return torch.linspace(0.5, 2.0, self.n_classes_without_void).numpy()
