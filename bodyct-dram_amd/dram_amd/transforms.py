"""On-device OneShot transforms.

The reference's affine-consistency losses (dram/metrics.py:213-310) push [N,C,D,H,W] device tensors through
`Flip3DOneShot`, `Rotate903DOneShot` and `Rescale3DOneShot` (dram/data_transforms.py:1140-1239): every dict entry
whose key contains "#" is transformed, "#image..." entries with trilinear and "#reference..." entries with nearest
interpolation when rescaling.  The classes here keep those names, constructor arguments, attributes
(`flip_axis`, `rotate_axis`, `rotate_times`, `scale_factor`) and the way a random instance draws its parameters
(same `random` / `numpy.random` calls in the same order), but run on the gather kernels of csrc/resample.hip instead
of torch.flip / torch.rot90 / F.interpolate, and stay differentiable where the reference's are (the loss transforms the
predicted probabilities).  `Rotate3DXOneShot` (affine_grid + grid_sample; commented out of the reference's transform
pool) is one gather kernel with a scatter adjoint."""
import itertools
import random

import numpy as np

from . import functional as HF

_SPATIAL_AXES = (2, 3, 4)


class _OneShot:
    """Sample-dict protocol shared by all transforms: tensors under keys with a '#' are transformed."""

    def _transform(self, key, tensor):   # pragma: no cover - overridden
        raise NotImplementedError

    def __call__(self, sample):
        out = {}
        for key, value in sample.items():
            out[key] = self._transform(key, value) if "#" in key else value
        return out


class Identity:
    def __init__(self):
        pass

    def __call__(self, sample):
        return sample


class Flip3DOneShot(_OneShot):
    """Mirror along `flip_axis` (a tuple of axes out of 2, 3, 4).  Without an argument: 1-3 axes, drawn as the
    reference does (one randint for the count, one sample from the combinations)."""

    def __init__(self, flip_axis=None, spatial_dim=2):
        self.spatial_dim = spatial_dim
        if flip_axis is None:
            count = random.randint(1, 3)
            candidates = list(itertools.combinations(range(self.spatial_dim, 5), count))
            flip_axis = random.sample(candidates, 1)[0]
        self.flip_axis = flip_axis

    def _transform(self, key, tensor):
        if tensor.dim() != 5:
            raise AssertionError("Flip3DOneShot expects [N,C,D,H,W] tensors")
        perm, flip = HF.signed_permutation([("flip", tuple(a % 5 for a in self.flip_axis))])
        return HF.spatial_permute_flip(tensor, perm, flip)


class Rotate903DOneShot(_OneShot):
    """`rotate_times` quarter turns in the plane `rotate_axis` = (a, b), torch.rot90 semantics.  Without arguments:
    an ordered axis pair sampled from the 6 permutations, then 1-3 turns."""

    def __init__(self, rotate_axis=None, rotate_times=None, spatial_dim=2):
        self.spatial_dim = spatial_dim
        if rotate_axis is None:
            pairs = list(itertools.permutations(range(self.spatial_dim, 5), 2))
            rotate_axis = random.sample(pairs, 1)[0]
        self.rotate_axis = rotate_axis
        self.rotate_times = rotate_times if rotate_times is not None else random.randint(1, 3)

    def _transform(self, key, tensor):
        if tensor.dim() != 5:
            raise AssertionError("Rotate903DOneShot expects [N,C,D,H,W] tensors")
        ops = HF.rot90_ops(self.rotate_times, tuple(a % 5 for a in self.rotate_axis))
        return HF.spatial_permute_flip(tensor, *HF.signed_permutation(ops))


class Rotate3DXOneShot(_OneShot):
    """Rotation by `theta` (radians; without an argument one np.random.uniform draw from the given range, as the
    reference does) about the x axis of the normalised grid: F.affine_grid + F.grid_sample with their defaults
    (trilinear, zeros padding, align_corners=False), data_transforms.py:1186-1208."""

    def __init__(self, theta=(0, np.pi)):
        self.theta = np.random.uniform(theta[0], theta[1], 1)

    def get_rot_mat(self):
        # (the reference evaluates cos / sin in float64 and casts the matrix to the data type)
        c, s = float(np.float32(np.cos(self.theta[0]))), float(np.float32(np.sin(self.theta[0])))
        return [1.0, 0.0, 0.0, 0.0, 0.0, c, -s, 0.0, 0.0, s, c, 0.0]

    def _transform(self, key, tensor):
        if tensor.dim() != 5:
            raise AssertionError("Rotate3DXOneShot expects [N,C,D,H,W] tensors")
        return HF.affine_sample(tensor, self.get_rot_mat())


class Rescale3DOneShot(_OneShot):
    """Resize to `scale_factor` read as a size (mode='size') or as per-axis factors (mode='factor'); without one,
    three independent draws from `rescale_factor_pool`.  Images: trilinear (align_corners=False), label maps: nearest."""

    def __init__(self, rescale_factor_pool=None, scale_factor=None, mode='size'):
        self.rescale_factor_pool = rescale_factor_pool
        self.mode = mode
        self.scale_factor = tuple(np.random.choice(rescale_factor_pool, 3)) if scale_factor is None else scale_factor

    def _transform(self, key, tensor):
        if "image" in key:
            resize = HF.interpolate_trilinear
        elif "reference" in key:
            resize = HF.interpolate_nearest
        else:
            raise NotImplementedError(f"Rescale3DOneShot: do not know how to interpolate {key!r}")
        if self.mode == 'size':
            return resize(tensor, size=tuple(int(v) for v in self.scale_factor))
        if self.mode == 'factor':
            return resize(tensor, scale_factor=tuple(float(v) for v in self.scale_factor))
        return tensor       # the reference leaves the tensor untouched for any other mode
