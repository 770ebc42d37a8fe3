"""Drop-in for the reference's flat module `dram/parts.py` on MI355X.

Same public names, constructor signatures, forward signatures, sub-module
attribute names (hence state-dict keys) and error behaviour as the reference;
every forward/backward runs on the hand-written gfx950 kernels of
libdram_hip.so (see dram_amd/).  Put this directory on sys.path (the reference
does the same with `dram/`) and `from parts import *` / `import models` work as
before.

Reference citations are to /root/reference/dram/parts.py.
"""
import functools  # noqa: F401  (re-exported: models.py star-imports these names, parts.py:1-8)
import math  # noqa: F401

import numpy as np  # noqa: F401
import torch
import torch.nn as nn
import torch.nn.functional as F  # noqa: F401
from torch.utils.checkpoint import checkpoint, checkpoint_sequential  # noqa: F401

from dram_amd import functional as HF
from dram_amd.modules import (HipBatchNorm3d, HipConv3d, HipGroupNorm, HipMaxPool3d, HipPReLU, HipReLU, HipSyncBatchNorm,
                              HipUpsample, run_conv_stack)


class Identity(nn.Module):
    """Pass-through used when no norm is selected (parts.py:10-15)."""

    def __init__(self, *args, **kwargs):
        super().__init__()

    def forward(self, x, args=None):
        return x


_NORMS = {
    # name -> factory(channels)            reference parts.py:18-33
    "bn": lambda c: HipBatchNorm3d(c),
    "bnt": lambda c: HipBatchNorm3d(c, affine=True, track_running_stats=False),
    "bntna": lambda c: HipBatchNorm3d(c, affine=False, track_running_stats=False),
    "ln": lambda c: HipGroupNorm(1, c),
    "lnna": lambda c: HipGroupNorm(1, c, affine=False),
    "in": lambda c: HipGroupNorm(c, c),
    "sbn": lambda c: HipSyncBatchNorm(c),
}


def normal_wrapper(normal_method, in_ch, in_ch_div=2):
    """Norm factory (parts.py:17-35).  Unknown names (and None) give Identity, as in the
    reference.  Deviation, on purpose: names are compared with `==`; the reference uses `is`,
    which silently yields Identity for a non-interned (runtime-built) string."""
    make = _NORMS.get(normal_method) if isinstance(normal_method, str) else None
    return make(in_ch) if make is not None else Identity()


def crop_concat_5d(t1, t2):
    """cat([t1, centre-crop of t2 to t1's D,H,W], dim=1); crop start = ceil((b-a)/2)
    (parts.py:37-46).  t1 is the smaller (upsampled) tensor and comes first."""
    assert (t1.dim() == t2.dim() == 5)
    assert (t1.shape[-1] <= t2.shape[-1])
    return HF.crop_concat(t1, t2)


def act_wrapper(act_method, num_parameters=1, init=0.25):
    """Activation factory (parts.py:48-54)."""
    if act_method == "relu":
        return HipReLU(inplace=True)
    if act_method == "prelu":
        return HipPReLU(num_parameters, init)
    raise NotImplementedError


def checkpoint_wrapper(module, segments, *tensors):
    """parts.py:57-64."""
    if segments > 0:
        return checkpoint(module, *tensors, use_reentrant=True)
    return module(*tensors)


def _per_conv(v, n):
    return list(v) if isinstance(v, (tuple, list)) else [v] * n


def _conv_stack(in_chs, out_chs, ksize, pad, stride, bias, norm_method, act_method, dropout, lite=False):
    """Sequential of [Conv3d, norm, act(, Dropout)] Sequentials -- the layout (and therefore the
    state-dict keys `conv_blocks.{j}.0.weight`, `conv_blocks.{j}.1.*`) of parts.py:84-110."""
    n = len(in_chs)
    ksize, pad, stride = _per_conv(ksize, n), _per_conv(pad, n), _per_conv(stride, n)
    stages = []
    for j, (ci, co) in enumerate(zip(in_chs, out_chs)):
        layers = [HipConv3d(ci, co, kernel_size=ksize[j], stride=stride[j], padding=pad[j], bias=bias)]
        if not lite:
            layers.append(normal_wrapper(norm_method, co))
        layers.append(act_wrapper(act_method))
        if dropout > 0 and not lite:
            layers.append(nn.Dropout(dropout))
        stages.append(nn.Sequential(*layers))
    return nn.Sequential(*stages)


class ConvBlock5d(nn.Module):
    """[conv -> norm -> ReLU] x len(in_chs) (parts.py:66-113).  NB the activation kwarg is
    spelt `act_methpd` in the reference; an `act_method=` passed by DC3D lands in **kwargs
    and the default 'relu' is used.  Kept as is."""

    def __init__(self, in_chs, base_chs, checkpoint_segments, conv_ksize,
                 conv_bias, conv_pad, dropout=0.1, conv_strides=1,
                 norm_method='bn', act_methpd='relu', lite=False,
                 **kwargs):
        super(ConvBlock5d, self).__init__()
        if dropout > 0 and not lite:
            print("use dropout in convs!")
        self.conv_blocks = _conv_stack(in_chs, base_chs, conv_ksize, conv_pad, conv_strides, conv_bias,
                                       norm_method, act_methpd, dropout, lite=lite)

    def forward(self, x, args=None):
        return run_conv_stack(self.conv_blocks, x)


class UpsampleConvBlock5d(nn.Module):
    """trilinear x`scale_factor` upsample -> crop_concat_5d(up, skip) -> conv stack
    (parts.py:116-155).  The concatenation is never materialised: the first conv reads both
    tensors (dram_conv3d_k3_fwd_ex)."""

    def __init__(self, in_chs, base_chs, checkpoint_segments, scale_factor,
                 conv_ksize, conv_bias, conv_pad, dropout=0.1,
                 norm_method='bn', act_methpd='relu', **kwargs):
        super(UpsampleConvBlock5d, self).__init__()
        self.checkpoint_segments = checkpoint_segments
        self.scale_factor = scale_factor
        self.conv_blocks = _conv_stack(in_chs, base_chs, conv_ksize, conv_pad, 1, conv_bias,
                                       norm_method, act_methpd, dropout)
        self.merge_func = kwargs.get('merge_func', crop_concat_5d)   # stored, never used (parts.py:148,153)
        self.upsample = HipUpsample(size=None, scale_factor=self.scale_factor, mode='trilinear', align_corners=True)

    def forward(self, inputs, cats, args=None):
        up_inputs = self.upsample(inputs)
        assert (up_inputs.dim() == cats.dim() == 5)
        assert (up_inputs.shape[-1] <= cats.shape[-1])
        first = self.conv_blocks[0][0]
        if isinstance(first, HipConv3d) and first.kernel_size == (3, 3, 3):
            return run_conv_stack(self.conv_blocks, up_inputs, skip=cats)
        return run_conv_stack(self.conv_blocks, crop_concat_5d(up_inputs, cats))


class ConvPoolBlock5d(nn.Module):
    """conv stack followed by MaxPool3d; returns (features, pooled) (parts.py:157-196)."""

    def __init__(self, in_ch_list, base_ch_list, checkpoint_segments,
                 conv_ksize, conv_bias, conv_pad,
                 pool_ksize, pool_strides, pool_pad, dropout=0.1,
                 conv_strdes=1, norm_method='bn', act_method="relu",
                 **kwargs):
        super(ConvPoolBlock5d, self).__init__()
        self.checkpoint_segments = checkpoint_segments
        self.conv_blocks = _conv_stack(in_ch_list, base_ch_list, conv_ksize, conv_pad, conv_strdes, conv_bias,
                                       norm_method, act_method, dropout)
        self.maxpool = HipMaxPool3d(kernel_size=pool_ksize, stride=pool_strides, padding=pool_pad)

    def forward(self, x, args=None):
        y = run_conv_stack(self.conv_blocks, x)
        pooled = self.maxpool(y)
        return y, pooled
