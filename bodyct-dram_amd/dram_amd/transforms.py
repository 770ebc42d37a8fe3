"""On-device OneShot transforms: the tensor transforms the reference's affine-consistency
losses apply to [N,C,D,H,W] device tensors (dram/data_transforms.py:1140-1239, used by metrics.py:213-310), with the
reference's class names, constructor arguments, random parameter choice and sample-dict protocol (keys containing
"#" are transformed: "#image..." trilinear, "#reference..." nearest in Rescale3DOneShot).  torch.flip / torch.rot90 /
F.interpolate are replaced by the gather kernels of csrc/resample.hip; all are differentiable where the reference's
are (the loss resizes the predicted probabilities).  Rotate3DXOneShot (affine_grid + grid_sample, commented out of the
reference's transform pool) is not provided."""
import random
from itertools import combinations, permutations

import numpy as np

from . import functional as HF


class Identity:
    def __init__(self):
        pass

    def __call__(self, sample):
        return sample


class Flip3DOneShot:
    """torch.flip over 1-3 randomly chosen axes of 2..4 (data_transforms.py:1140-1159)."""

    def __init__(self, flip_axis=None, spatial_dim=2):
        self.spatial_dim = spatial_dim
        if flip_axis is None:
            toss_int = random.randint(1, 3)
            all_p = list(combinations([n for n in range(self.spatial_dim, 5)], toss_int))
            flip_axis = random.sample(all_p, 1)[0]
        self.flip_axis = flip_axis

    def _flip_axis(self, data):
        assert data.dim() == 5
        perm, flip = HF.signed_permutation([("flip", tuple(d % 5 for d in self.flip_axis))])
        return HF.spatial_permute_flip(data, perm, flip)

    def __call__(self, sample):
        return {k: (self._flip_axis(v) if "#" in k else v) for k, v in sample.items()}


class Rotate903DOneShot:
    """torch.rot90(rotate_times, rotate_axis) (data_transforms.py:1161-1181)."""

    def __init__(self, rotate_axis=None, rotate_times=None, spatial_dim=2):
        self.spatial_dim = spatial_dim
        if rotate_axis is None:
            all_p = list(permutations(list(range(self.spatial_dim, 5)), 2))
            rotate_axis = random.sample(all_p, 1)[0]
        self.rotate_axis = rotate_axis
        self.rotate_times = random.randint(1, 3) if rotate_times is None else rotate_times

    def _rotate_axis(self, data):
        assert data.dim() == 5
        perm, flip = HF.signed_permutation(HF.rot90_ops(self.rotate_times, tuple(d % 5 for d in self.rotate_axis)))
        return HF.spatial_permute_flip(data, perm, flip)

    def __call__(self, sample):
        return {k: (self._rotate_axis(v) if "#" in k else v) for k, v in sample.items()}


class Rescale3DOneShot:
    """F.interpolate to a size / by factors drawn from `rescale_factor_pool` (data_transforms.py:1202-1239)."""

    def __init__(self, rescale_factor_pool=None, scale_factor=None, mode='size'):
        self.rescale_factor_pool = rescale_factor_pool
        self.mode = mode
        if scale_factor is None:
            scale_factor = tuple(np.random.choice(self.rescale_factor_pool, 3))
        self.scale_factor = scale_factor

    def _rescale(self, data, mode):
        fn = HF.interpolate_trilinear if mode == 'trilinear' else HF.interpolate_nearest
        if self.mode == 'factor':
            return fn(data, scale_factor=tuple(float(s) for s in self.scale_factor))
        if self.mode == 'size':
            return fn(data, size=tuple(int(s) for s in self.scale_factor))
        return data

    def __call__(self, sample):
        new_sample = {}
        for k, v in sample.items():
            if "#" in k:
                if "image" in k:
                    mode = 'trilinear'
                elif "reference" in k:
                    mode = 'nearest'
                else:
                    raise NotImplementedError
                v = self._rescale(v, mode)
            new_sample[k] = v
        return new_sample
