"""nn.Module leaves whose forward/backward run on the gfx950 kernels.

Each class subclasses the torch module the reference instantiates (same
constructor, same parameters/buffers, same state-dict keys) and only replaces
`forward`, so `isinstance(m, nn.BatchNorm3d)`-style code in callers (e.g. the
reference's HeNorm initialiser, dram/models.py:22-33) keeps working.
"""
import torch
import torch.nn as nn

from . import functional as HF


class HipConv3d(nn.Conv3d):
    """nn.Conv3d restricted to what the reference instantiates: 3x3x3/pad 1 (parts.py:95..185)
    and 1x1x1/pad 0 (models.py:109), stride 1, no dilation/groups."""

    def _kind(self):
        if self.stride != (1, 1, 1) or self.dilation != (1, 1, 1) or self.groups != 1 or self.padding_mode != "zeros":
            raise NotImplementedError("HipConv3d: only stride 1, dilation 1, groups 1, zero padding is implemented")
        if self.kernel_size == (3, 3, 3) and self.padding == (1, 1, 1):
            return 3
        if self.kernel_size == (1, 1, 1) and self.padding == (0, 0, 0):
            return 1
        raise NotImplementedError(f"HipConv3d: kernel {self.kernel_size} / padding {self.padding} is not implemented "
                                  f"(the DRAM models use 3x3x3 pad 1 and 1x1x1 pad 0)")

    def forward(self, x, skip=None):
        if self._kind() == 3:
            return HF.conv3d_k3(x, self.weight, self.bias, skip)
        if skip is not None:
            raise ValueError("HipConv3d: a skip input is only supported by the 3x3x3 kernel")
        return HF.conv3d_k1(x, self.weight, self.bias)


class _NormMixin:
    def forward(self, x, relu=False):   # pragma: no cover - overridden
        raise NotImplementedError


class HipBatchNorm3d(nn.BatchNorm3d, _NormMixin):
    """nn.BatchNorm3d ("bn", "bnt", "bntna" of normal_wrapper, parts.py:18-25)."""

    # number of times the running statistics absorb this batch (2 inside a block that the
    # reference would run twice per step through torch.utils.checkpoint; models.DC3D._run)
    stat_updates = 1

    def bookkeeping(self):
        """The per-call bookkeeping of torch.nn.modules.batchnorm._BatchNorm.forward, repeated stat_updates times with
        the same batch statistics (r <- (1-m) r + m b twice == once with 1-(1-m)^2).  Increments num_batches_tracked.
        Returns (use_batch_stats, exponential_average_factor, running_mean, running_var)."""
        eaf = 0.0 if self.momentum is None else self.momentum
        if self.training and self.track_running_stats and self.num_batches_tracked is not None:
            keep = 1.0
            for _ in range(max(1, int(self.stat_updates))):
                self.num_batches_tracked.add_(1)
                m = 1.0 / float(self.num_batches_tracked) if self.momentum is None else self.momentum
                keep *= (1.0 - m)
            eaf = 1.0 - keep
        use_batch = self.training or (self.running_mean is None and self.running_var is None)
        rm = self.running_mean if (not self.training or self.track_running_stats) else None
        rv = self.running_var if (not self.training or self.track_running_stats) else None
        return use_batch, eaf, rm, rv

    def forward(self, x, relu=False):
        self._check_input_dim(x)
        use_batch, eaf, rm, rv = self.bookkeeping()
        return HF.norm_act(x, self.weight, self.bias, rm, rv, HF.NORM_BATCH, 1, use_batch, eaf, self.eps, relu)


class HipSyncBatchNorm(HipBatchNorm3d):
    """"sbn" (parts.py:32-33: nn.SyncBatchNorm).  In training mode with an initialised torch.distributed group of
    more than one rank the batch statistics span all ranks (two small exchanges per layer and direction, see
    functional.SyncBatchNormFn); otherwise -- like nn.SyncBatchNorm in the reference's single-process runs -- it is
    plain BatchNorm.  `process_group` as in nn.SyncBatchNorm (None = the default group)."""

    process_group = None

    def forward(self, x, relu=False):
        import torch.distributed as dist
        sync = self.training and dist.is_available() and dist.is_initialized() and dist.get_world_size(self.process_group) > 1
        if not sync:
            return super().forward(x, relu)
        self._check_input_dim(x)
        eaf = 0.0 if self.momentum is None else self.momentum
        if self.track_running_stats and self.num_batches_tracked is not None:
            keep = 1.0
            for _ in range(max(1, int(self.stat_updates))):
                self.num_batches_tracked.add_(1)
                m = 1.0 / float(self.num_batches_tracked) if self.momentum is None else self.momentum
                keep *= (1.0 - m)
            eaf = 1.0 - keep
        rm = self.running_mean if self.track_running_stats else None
        rv = self.running_var if self.track_running_stats else None
        return HF.sync_batch_norm(x, self.weight, self.bias, rm, rv, eaf, self.eps, relu, self.process_group)


class HipGroupNorm(nn.GroupNorm, _NormMixin):
    """nn.GroupNorm ("ln", "lnna", "in" of normal_wrapper, parts.py:26-31)."""

    def forward(self, x, relu=False):
        return HF.norm_act(x, self.weight, self.bias, None, None, HF.NORM_GROUP, self.num_groups, True, 0.0,
                           self.eps, relu)


class HipReLU(nn.ReLU):
    """nn.ReLU (act_wrapper "relu", parts.py:49-50).  Returns a new tensor; `inplace` only
    matters for memory in the reference and is ignored."""

    def forward(self, x):
        return HF.relu(x)


class HipPReLU(nn.PReLU):
    """nn.PReLU(num_parameters, init) (act_wrapper "prelu", parts.py:51-52)."""

    def forward(self, x):
        return HF.prelu(x, self.weight)


class HipMaxPool3d(nn.MaxPool3d):
    """nn.MaxPool3d(2, 2, 0) (parts.py:191)."""

    def forward(self, x):
        def _t(v):
            return tuple(v) if isinstance(v, (tuple, list)) else (v,) * 3
        if _t(self.kernel_size) != (2, 2, 2) or _t(self.stride) != (2, 2, 2) or _t(self.padding) != (0, 0, 0) \
                or _t(self.dilation) != (1, 1, 1) or self.ceil_mode or self.return_indices:
            raise NotImplementedError("HipMaxPool3d: only kernel 2 / stride 2 / padding 0 is implemented")
        return HF.max_pool3d_2(x)


class HipUpsample(nn.Upsample):
    """nn.Upsample(mode='trilinear', align_corners=True) (parts.py:149, models.py:146)."""

    def forward(self, x):
        if self.mode != "trilinear" or not self.align_corners:
            raise NotImplementedError("HipUpsample: only mode='trilinear', align_corners=True is implemented")
        return HF.upsample_trilinear_ac(x, size=self.size, scale_factor=self.scale_factor)


def run_conv_stack(conv_blocks, x, skip=None):
    """Evaluate a `conv_blocks` Sequential-of-Sequentials ([conv, norm, act(, dropout)] per entry,
    parts.py:102-110) with the norm and ReLU fused into one pass.  `skip` is fed to the first
    conv as the second half of a never-materialised crop_concat_5d."""
    for j, seq in enumerate(conv_blocks):
        mods = list(seq)
        conv = mods[0]
        x = conv(x, skip) if (j == 0 and skip is not None) else conv(x)
        i = 1
        while i < len(mods):
            m = mods[i]
            if isinstance(m, _NormMixin) and i + 1 < len(mods) and isinstance(mods[i + 1], HipReLU):
                x = m(x, relu=True)
                i += 2
            else:
                x = m(x)
                i += 1
    return x
