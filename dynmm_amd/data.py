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

    def class_counts(self):
        """(pixels per class, pixels of the images containing the class) over this source's label maps, void included at
        index 0 — the two histograms src/datasets/dataset_base.py:160-186 accumulates.  Additive over shards: under data
        parallel the ranks SUM them before forming the weights (dynmm_amd/train.py), so every replica uses the weights of
        the whole training set, as the reference does."""
        import numpy as np
        n_cls = self.n_classes_without_void + 1
        per_class, with_class = np.zeros(n_cls), np.zeros(n_cls)
        for i in range(len(self)):
            n = min(self.bs, self.n - i * self.bs)
            label = synth.synth_labels(n, self.h, self.w, seed=self.seed + 17 * i + 1, device='cpu')
            for img in label.reshape(n, -1).to(torch.int64):
                dist = np.bincount(img.numpy(), minlength=n_cls)[:n_cls]
                per_class += dist
                with_class += (dist > 0) * img.numel()
        return per_class, with_class

    @staticmethod
    def weights_from_counts(per_class, with_class, weight_mode='median_frequency', c=1.02):
        """src/datasets/dataset_base.py:188-208 (void = class 0 removed): median_frequency = median(f) / f with
        f = pixels of the class / pixels of the images containing it; logarithmic = 1 / log(c + p); linear = counts."""
        import numpy as np
        if weight_mode not in ('median_frequency', 'logarithmic', 'linear'):
            raise ValueError(f'unknown class weighting {weight_mode!r}')
        per_class, with_class = per_class[1:], with_class[1:]
        if weight_mode == 'linear':
            w = per_class
        elif weight_mode == 'median_frequency':
            freq = per_class / with_class
            w = np.median(freq) / freq
        else:
            w = 1.0 / np.log(c + per_class / per_class.sum())
        if np.isnan(np.sum(w)):
            raise ValueError('class weighting contains NaNs')
        return w

    def compute_class_weights(self, weight_mode='median_frequency', c=1.02):
        return self.weights_from_counts(*self.class_counts(), weight_mode=weight_mode, c=c)
