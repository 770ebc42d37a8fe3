"""Fused execution of DC3D's conv -> norm -> ReLU chains.

The per-op path (`functional.py`, one autograd Function per ATen op of the reference) writes, for every
conv -> norm -> ReLU stage (reference dram/parts.py:177-187), the raw conv output, reads it for the norm
statistics, and reads it again to write the activated tensor that the next layer consumes -- and keeps both
tensors for backward.  Here the whole network (reference dram/models.py:120-147) is ONE autograd Function whose
forward and backward are explicit sequences of C-ABI calls over "lazy" tensors:

  * a conv writes its raw output y once; its epilogue leaves the BatchNorm / GroupNorm moments of y as partials
    (`dram_conv3d_k3_fwd_fused`), `dram_norm_finalize_parts` turns them into per-row coefficients {a, b};
  * the activated tensor act(a*y + b) is never written: every consumer -- the next conv (forward and
    backward-weights), the max-pool, the trilinear upsample, the 1x1x1 head -- applies it while loading;
  * backward keeps the reference's arithmetic (norm backward = two reductions + one apply, in place on the
    incoming gradient; conv backward-data / backward-weights on the matrix cores) and recomputes the upsampled
    tensor of an UpsampleConvBlock5d for its backward-weights instead of keeping it.

Saved for backward per stage: the raw output and {a, b, mean, rstd} -- about 0.4x of what the per-op path keeps
(DESIGN.md section 5), with values bit-identical to that path's (same fmaf / fmaxf on the same inputs).

Not an oracle and not a fallback: every step is a kernel of libdram_hip.so.  Networks the engine does not cover
(PReLU, dropout, `lite` blocks, conv kernels other than 3x3x3, checkpoint_mode="recompute") run the per-op path.
"sbn" (cross-rank BatchNorm, parts.py:32-33) is covered: the per-rank statistics of the conv epilogue are combined over the
process group (_sync_forward_stats), and so are the two sums of the norm's backward.
"""
import os as _os

import torch
import torch.nn as nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib
from . import functional as HF
from ._lib import call
from .functional import NORM_BATCH, NORM_GROUP, _p, _stream, _ws, crop_offsets
from .modules import HipBatchNorm3d, HipConv3d, HipGroupNorm, HipMaxPool3d, HipReLU, HipSyncBatchNorm, HipUpsample


class Lazy:
    """The tensor act(coef[row][0] * raw + coef[row][1]) (ReLU if relu), or `raw` itself when coef is None."""
    __slots__ = ("raw", "coef", "relu")

    def __init__(self, raw, coef=None, relu=False):
        self.raw, self.coef, self.relu = raw, coef, bool(relu)

    @property
    def shape(self):
        return self.raw.shape

    def materialise(self):
        if self.coef is None:
            return self.raw
        N, C = self.raw.shape[:2]
        S = self.raw.numel() // (N * C)
        y = torch.empty_like(self.raw)
        call("dram_row_affine_act", _p(self.raw), _p(self.coef), _p(y), int(self.relu), N * C, S, _stream())
        return y


# An UpsampleConvBlock5d's upsampled tensor is the largest activation of the network (128 channels at full
# resolution in us_modules.2: 22 % of everything the per-op path saves).  Backward needs it once, as the x operand
# of the first conv's backward-weights.  It is kept when small, and produced again from the low-resolution lazy
# tensor (one bandwidth pass, ~0.3 % of a step) when it is larger than this fraction of the device memory.
KEEP_UPSAMPLED_BELOW = float(_os.environ.get("DRAM_KEEP_UPSAMPLED_FRAC", "0.08"))


# Lazy tensors trade time for memory: applying the norm on load costs the consumer kernels 3-6 % (forward conv) and
# 4-5 % (backward-weights) of their time (scripts/bench_fused.py), a materialising pass costs one read + one write of
# the tensor and 4 bytes per element of HBM until backward.  So a stage's activated output is written when it is
# small (below this fraction of the device memory) and stays lazy when it is large: with the 64 x 128^3 benchmark at
# micro-batch 16 everything is written (8.6 GB per full-resolution tensor = 3 %), at micro-batch 32 the
# full-resolution stages stay lazy and the step fits 288 GB.
MATERIALISE_BELOW = float(_os.environ.get("DRAM_MATERIALISE_FRAC", "0.04"))


class Upsampled:
    """The x`scale` trilinear (align_corners=True) upsampling of a Lazy, as a recipe (see KEEP_UPSAMPLED_BELOW)."""
    __slots__ = ("src", "size", "kept")

    def __init__(self, src, size):
        self.src, self.size, self.kept = src, tuple(int(v) for v in size), None

    def produce(self, lo=0, hi=None, keep_below=0):
        """The upsampled tensor of samples [lo, hi) (default: all); the whole tensor is kept for backward when it is
        smaller than `keep_below` bytes."""
        N, C, D, H, W = self.src.raw.shape
        hi = N if hi is None else hi
        if self.kept is not None:
            return self.kept[lo:hi]
        n = hi - lo
        y = torch.empty((n, C) + self.size, dtype=torch.float32, device=self.src.raw.device)
        call("dram_upsample_trilinear_ac_fwd_lazy", _p(self.src.raw[lo:hi]), _p(_rows(self.src.coef, lo, hi, C)),
             int(self.src.relu), _p(y), n, C, D, H, W, *self.size, _stream())
        if (lo, hi) == (0, N) and y.numel() * 4 < keep_below:
            self.kept = y
        return y


# ------------------------------------------------------------------------------------------------ applicability
def _stage_modules(seq):
    """[conv, norm, relu] of one entry of `conv_blocks` (parts.py:102-110), or None if the stage is anything else."""
    mods = list(seq)
    if len(mods) != 3:
        return None
    conv, norm, act = mods
    if not isinstance(conv, HipConv3d) or conv.kernel_size != (3, 3, 3) or conv.padding != (1, 1, 1) \
            or conv.stride != (1, 1, 1) or conv.dilation != (1, 1, 1) or conv.groups != 1 or conv.bias is not None:
        return None
    if not isinstance(norm, (HipBatchNorm3d, HipGroupNorm)):        # (HipSyncBatchNorm is a HipBatchNorm3d: _sync_group)
        return None
    if not isinstance(act, HipReLU):
        return None
    return conv, norm, act


def supports(model):
    """True if `model` (a models.DC3D) is made of the standard blocks only."""
    try:
        blocks = list(model.ds_modules) + [model.bg] + (list(model.us_modules) if model.us_modules is not None else [])
        for blk in blocks:
            if any(_stage_modules(seq) is None for seq in blk.conv_blocks):
                return False
        for ds in model.ds_modules:
            if not isinstance(ds.maxpool, HipMaxPool3d):
                return False
        for us in (model.us_modules or []):
            if not isinstance(us.upsample, HipUpsample) or us.upsample.mode != "trilinear" or not us.upsample.align_corners:
                return False
        top = model.top_layer
        return isinstance(top, HipConv3d) and top.kernel_size == (1, 1, 1) and top.padding == (0, 0, 0)
    except AttributeError:
        return False


def parameters_of(model):
    """The parameters the engine differentiates, in a fixed order (the autograd Function's tensor inputs)."""
    return [p for p in model.parameters()]


# ------------------------------------------------------------------------------------------------ forward pieces
class _Stage:
    """What backward needs of one conv -> norm -> ReLU stage."""
    __slots__ = ("conv", "norm", "inp", "skip", "geom", "y", "coef", "mean", "rstd", "kind", "groups", "batch_stats",
                 "out", "need_input_grad", "ranges", "sync")


def _norm_plan(norm, training):
    """(kind, groups, use_batch_stats, running_mean, running_var, exponential average factor): the bookkeeping of
    HipBatchNorm3d.forward / HipGroupNorm.forward (torch.nn.modules.batchnorm._BatchNorm.forward)."""
    if isinstance(norm, HipGroupNorm):
        return NORM_GROUP, norm.num_groups, True, None, None, 0.0
    use_batch, eaf, rm, rv = norm.bookkeeping()
    return NORM_BATCH, 1, use_batch, rm, rv, eaf


def _sync_group(norm, use_batch):
    """(True, group) if this norm is "sbn" (parts.py:32-33) in a run whose batch statistics must span the ranks of its
    process group: training-mode HipSyncBatchNorm with torch.distributed initialised on more than one rank -- the
    condition of HipSyncBatchNorm.forward.  Otherwise (False, None): plain BatchNorm, as nn.SyncBatchNorm is in the
    reference's single-process runs."""
    if not use_batch or not isinstance(norm, HipSyncBatchNorm):
        return False, None
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(norm.process_group) > 1):
        return False, None
    return True, norm.process_group


def _sync_forward_stats(norm, group, parts, nparts, ws, mean, rstd, coef, count, rm, rv, eaf, N, Co, S, st):
    """Cross-rank combine of a stage's batch statistics (what functional.SyncBatchNormFn.forward does for the per-op
    path): this rank's per-channel {mean, M2} in fp64 straight from the conv epilogue's partials (dram_bn_parts_stats -- not
    from save_rstd, whose inversion rstd^-2 - eps cancels for channels with variance far below eps), all-gather, Chan's
    formula in fp64, running statistics from the GLOBAL moments, then mean / rstd / per-row {a, b} written from them.
    Returns the global element count per channel."""
    import torch.distributed as dist
    eps = float(norm.eps)
    dev = mean.device
    local = torch.empty(2 * Co + 1, dtype=torch.float64, device=dev)
    call("dram_bn_parts_stats", _p(parts), nparts, _p(local), N, Co, S, _p(ws), ws.numel(), st)
    local[2 * Co] = float(count)
    world = dist.get_world_size(group)
    allst = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(allst, local, group=group)
    allst = torch.stack(allst)                               # [world, 2 Co + 1]
    cnt = allst[:, 2 * Co].view(world, 1)
    means, m2s = allst[:, 0:2 * Co:2], allst[:, 1:2 * Co:2]
    total = cnt.sum()
    gmean = (means * cnt).sum(0) / total
    m2 = (m2s + cnt * (means - gmean) ** 2).sum(0)
    gvar = m2 / total
    gmean_f, gvar_f = gmean.float(), gvar.float()
    if rm is not None:
        with torch.no_grad():
            unb = (m2 / (total - 1.0)).float() if float(total) > 1.0 else gvar_f
            rm.mul_(1.0 - eaf).add_(gmean_f, alpha=eaf)
            rv.mul_(1.0 - eaf).add_(unb, alpha=eaf)
    call("dram_bn_eval_coef", _p(norm.weight), _p(norm.bias), _p(gmean_f), _p(gvar_f), _p(mean), _p(rstd), _p(coef), eps, N, Co, st)
    return float(total)


# A stage whose first input is an Upsampled recipe runs in slices of samples when the upsampled tensor (and, in
# backward, its gradient) would exceed this fraction of the device memory: every kernel of the stage is independent
# per sample, only the norm statistics span the batch -- and they are finalised once, over all slices' partials.
# That is what lets 64 x 128^3 chunks go through as ONE batch (whole-batch BatchNorm statistics, like the reference).
SLICE_UPSAMPLED_ABOVE = float(_os.environ.get("DRAM_SLICE_UPSAMPLED_FRAC", "0.08"))


def _rows(coef, lo, hi, C):
    return None if coef is None else coef[2 * lo * C:2 * hi * C]


def _lazy_slice(lz, lo, hi):
    return Lazy(lz.raw[lo:hi], _rows(lz.coef, lo, hi, lz.raw.shape[1]), lz.relu)


def _slices(inp, N, budget):
    """Sample ranges in which a stage with input `inp` is executed: slices whose upsampled tensor stays below `budget`
    bytes.  Decided ONCE, in forward; backward walks the ranges the tape recorded."""
    if not isinstance(inp, Upsampled):
        return [(0, N)]
    C = inp.src.raw.shape[1]
    per_sample = 4 * C * inp.size[0] * inp.size[1] * inp.size[2]
    n = max(1, min(N, int(budget // per_sample)))
    return [(lo, min(N, lo + n)) for lo in range(0, N, n)]


def _conv_stage(conv, norm, inp, skip, training, record, plan):
    """y = conv(inp ++ crop(skip)) with the moments of y from the epilogue -> Lazy(y, coef, relu).  `inp` is a Lazy or
    an Upsampled recipe (then produced here -- whole, or slice by slice -- used, and dropped unless it is small)."""
    up = isinstance(inp, Upsampled)
    N, C1 = inp.src.raw.shape[:2] if up else inp.raw.shape[:2]
    D, H, W = inp.size if up else inp.raw.shape[2:]
    w = conv.weight
    Co, Ci = w.shape[0], w.shape[1]
    dev = w.device
    if skip is not None:
        C2, D2, H2, W2 = skip.raw.shape[1:]
        if skip.raw.shape[0] != N or not (D <= D2 and H <= H2 and W <= W2):
            raise ValueError("fused conv stage: the skip tensor must have the same batch and be at least as large")
        oz, oy, ox = crop_offsets((D, H, W), (D2, H2, W2))
    else:
        C2 = D2 = H2 = W2 = oz = oy = ox = 0
    if C1 + C2 != Ci:
        raise ValueError(f"fused conv stage: input has {C1 + C2} channels, weight expects {Ci}")
    kind, groups, use_batch, rm, rv, eaf = _norm_plan(norm, training)
    S = D * H * W
    st = _stream()
    wt = HF._pack(w, 0)
    y = torch.empty((N, Co, D, H, W), dtype=torch.float32, device=dev)
    nstat = Co if kind == NORM_BATCH else N * groups
    mean = torch.empty(nstat, dtype=torch.float32, device=dev)
    rstd = torch.empty(nstat, dtype=torch.float32, device=dev)
    coef = torch.empty(2 * N * Co, dtype=torch.float32, device=dev)
    gamma, beta = norm.weight, norm.bias
    nparts = _lib.lib.dram_conv3d_k3_stats_parts(Ci, Co, D, H, W) if use_batch else 0
    parts = torch.empty(N * Co * nparts * 3, dtype=torch.float32, device=dev) if use_batch else None
    ranges = _slices(inp, N, plan.slice_up) if up else [(0, N)]
    plan.sliced_stages += len(ranges) > 1
    for lo, hi in ranges:
        n = hi - lo
        if up:
            x1 = Lazy(inp.produce(lo, hi, keep_below=plan.keep_up if record is not None and len(ranges) == 1 else 0))
        else:
            x1 = inp if (lo, hi) == (0, N) else _lazy_slice(inp, lo, hi)
        sk = None if skip is None else (skip if (lo, hi) == (0, N) else _lazy_slice(skip, lo, hi))
        vox = n * S
        name = HF.conv_fwd_kernel_name((D, H, W), Co, Ci, fused=True, src=(x1.raw, sk.raw if sk is not None else None, ox))
        HF._timed_call(name, 54.0 * Ci * Co * vox, 4.0 * (Ci + Co) * vox,
                       "dram_conv3d_k3_fwd_fused", _p(x1.raw), C1, _p(x1.coef), int(x1.relu),
                       _p(sk.raw) if sk is not None else None, C2, _p(sk.coef) if sk is not None else None,
                       int(sk.relu) if sk is not None else 0, D2, H2, W2, oz, oy, ox, _p(wt), None, _p(y[lo:hi]),
                       _p(parts[lo * Co * nparts * 3:]) if use_batch else None, nparts, n, Co, D, H, W, st)
        del x1
    sync, group = _sync_group(norm, training and use_batch and kind == NORM_BATCH)
    total = None
    if use_batch:
        ws = _ws(_lib.lib.dram_norm_parts_ws_bytes(N, Co, nparts), dev)
        if sync:        # "sbn": the statistics span the ranks (two small exchanges per stage and direction)
            total = _sync_forward_stats(norm, group, parts, nparts, ws, mean, rstd, coef, N * S, rm, rv, float(eaf), N, Co, S, st)
        else:
            call("dram_norm_finalize_parts", _p(parts), nparts, _p(gamma), _p(beta), _p(mean), _p(rstd), _p(coef), _p(rm), _p(rv),
                 float(eaf), float(norm.eps), kind, groups, N, Co, S, _p(ws), ws.numel(), st)
    else:   # eval-mode BatchNorm: coefficients from the running statistics
        call("dram_bn_eval_coef", _p(gamma), _p(beta), _p(rm), _p(rv), _p(mean), _p(rstd), _p(coef), float(norm.eps), N, Co, st)
    out = Lazy(y, coef, relu=True)
    if record is not None and y.numel() * 4 < plan.materialise:
        out = Lazy(out.materialise())        # (inference keeps everything lazy: nothing is kept there anyway)
    if record is not None:
        s = _Stage()
        s.conv, s.norm, s.inp, s.skip = conv, norm, inp, skip
        s.geom = (C1, C2, D2, H2, W2, oz, oy, ox)
        s.y, s.coef, s.mean, s.rstd = y, coef, mean, rstd
        s.kind, s.groups, s.batch_stats, s.out = kind, groups, bool(use_batch), out
        s.sync = (group, total) if sync else None
        s.ranges = ranges
        record.append(("conv", s))
    return out


def _conv_stack(conv_blocks, inp, skip, training, record, plan):
    cur = inp
    for j, seq in enumerate(conv_blocks):
        conv, norm, _ = _stage_modules(seq)
        cur = _conv_stage(conv, norm, cur, skip if j == 0 else None, training, record, plan)
    return cur


def _pool(lz, record):
    N, C, D, H, W = lz.raw.shape
    out = torch.empty((N, C, D // 2, H // 2, W // 2), dtype=torch.float32, device=lz.raw.device)
    idx = torch.empty(out.shape, dtype=torch.uint8, device=lz.raw.device)
    call("dram_maxpool3d_2_fwd_lazy", _p(lz.raw), _p(lz.coef), int(lz.relu), _p(out), _p(idx), N, C, D, H, W, _stream())
    res = Lazy(out)
    if record is not None:
        record.append(("pool", lz, idx, res))
    return res


def _bn_modules(block):
    return [m for m in block.modules() if isinstance(m, nn.BatchNorm3d)]


# ------------------------------------------------------------------------------------------------ memory modes
# "speed": activated outputs written (below MATERIALISE_BELOW), upsampled tensors kept (below KEEP_UPSAMPLED_BELOW);
# "tight": the three thresholds at 2 % of the device memory -- everything large stays lazy, upsampled-input stages run in
# slices.  "auto" (default) picks "tight" when writing everything would not fit: measured on the 64 x 128^3 benchmark,
# "speed" peaks at 123 GB (micro-batch 16) / 195 GB (32), "tight" at 251 GB for all 64 chunks as ONE batch.
MEMORY_MODE = _os.environ.get("DRAM_ENGINE_MEMORY", "auto")
_SPEED = (MATERIALISE_BELOW, KEEP_UPSAMPLED_BELOW, SLICE_UPSAMPLED_ABOVE)
_TIGHT = (0.02, 0.02, 0.02)


def _raw_output_bytes(model, x):
    """Bytes of all raw conv outputs of one forward (what the engine keeps at least)."""
    vox = 4.0 * x.shape[0] * x.shape[2] * x.shape[3] * x.shape[4]
    total, scale = 0.0, 1.0
    for ds in model.ds_modules:
        total += scale * sum(seq[0].out_channels for seq in ds.conv_blocks)
        scale /= 8.0
    total += scale * sum(seq[0].out_channels for seq in model.bg.conv_blocks)
    for i, us in enumerate(model.us_modules or []):
        if model.stacking == i:
            break
        scale *= 8.0
        total += scale * sum(seq[0].out_channels for seq in us.conv_blocks)
    return total * vox


class _Plan:
    """The memory decisions of ONE forward call, in bytes; kept on its tape, so that the matching backward -- whatever ran in
    between (an eval pass, another batch size, another model, another thread) -- releases and slices what forward planned."""
    __slots__ = ("mode", "materialise", "keep_up", "slice_up", "sliced_stages")

    def __init__(self, mode, fracs, cap):
        self.mode = mode
        self.materialise, self.keep_up, self.slice_up = (f * cap for f in fracs)
        self.sliced_stages = 0          # (diagnostics) stages that ran in more than one slice of samples


def _memory_plan(model, x, training):
    cap = torch.cuda.get_device_properties(x.device).total_memory
    mode = MEMORY_MODE
    if mode == "auto":
        # the written activations and kept upsampled tensors add ~1.2x the raw outputs on top of them
        mode = "tight" if training and 2.2 * _raw_output_bytes(model, x) > 0.62 * cap else "speed"
    if mode == "tight":
        fracs = _TIGHT
    elif mode == "speed":
        fracs = _SPEED
    else:       # "manual": the three module-level thresholds as the caller set them
        fracs = (MATERIALISE_BELOW, KEEP_UPSAMPLED_BELOW, SLICE_UPSAMPLED_ABOVE)
    return _Plan(mode, fracs, cap)


LAST_PLAN = None        # (diagnostics only, never read by the engine) the plan of the most recent forward call
MAX_PLANES = 65535      # (n, c) planes per launch of the row / plane kernels (norm, pool, resize: planes ride on grid.y)


def forward(model, x, record, taps=()):
    """DC3D.forward (models.py:120-147) / the backbone of DC3DATGeneric.forward (models.py:543-590) on lazy tensors.
    `record`: list that receives the tape for backward, or None (inference).  `taps`: layer ids (0 .. n_layers-1: the skip
    feature of that ConvPoolBlock5d, n_layers: the bottleneck output, n_layers+1+i: the output of up-block i) whose
    activated feature maps are written out as well -- DC3DATGeneric feeds them, detached, to its attention module
    (models.py:556,566,578).  Returns (dense output [N, out_ch, D, H, W], {layer id: feature map})."""
    global LAST_PLAN
    training = model.training
    plan = LAST_PLAN = _memory_plan(model, x, record is not None)
    L = model.n_layers
    x = HF._chk(x, "DC3D input", 5)
    widest = max(seq[0].out_channels for blk in list(model.ds_modules) + [model.bg] + list(model.us_modules or [])
                 for seq in blk.conv_blocks)
    if x.shape[0] * widest > MAX_PLANES:
        raise ValueError(f"DC3D: {x.shape[0]} chunks x {widest} channels = {x.shape[0] * widest} (n, c) planes exceed the "
                         f"{MAX_PLANES} a kernel launch addresses: run micro-batches of at most {MAX_PLANES // widest} chunks "
                         f"(DataParallelTrainer.step(batch, micro_batch=...))")
    grad_flows = record is not None          # the reference re-runs a checkpointed block in backward only then
    taps = set(taps)
    tapped = {}
    # DC3D indexes the checkpoint flag of up-block i with n_layers + i, DC3DATGeneric with n_layers + 1 + i (models.py:140,573)
    us_flag = L + int(getattr(model, "us_flag_offset", 0))
    cur = Lazy(x)
    if record is not None:
        record.append(("input", cur, plan))
    skips = []

    def run_block(flag, block, fn):
        # `checkpoint_layers` in 'stats' mode: a flagged block's BatchNorm running statistics absorb the batch twice
        # (models.DC3D._run)
        twice = flag > 0 and training and grad_flows
        norms = _bn_modules(block) if twice else []
        for m in norms:
            m.stat_updates = 2
        try:
            return fn()
        finally:
            for m in norms:
                m.stat_updates = 1

    for i, ds in enumerate(model.ds_modules):
        feat = run_block(model.checkpoint_layers[i], ds, lambda: _conv_stack(ds.conv_blocks, cur, None, training, record, plan))
        skips.append(feat)
        if i in taps:
            tapped[i] = feat.materialise()
        cur = _pool(feat, record)
    cur = run_block(model.checkpoint_layers[L], model.bg, lambda: _conv_stack(model.bg.conv_blocks, cur, None, training, record, plan))
    if L in taps:
        tapped[L] = cur.materialise()
    if model.us_modules is not None:
        for i, (us, skip) in enumerate(zip(model.us_modules, reversed(skips))):
            if model.stacking == i:
                break
            D, H, W = cur.raw.shape[2:]
            sf = us.scale_factor if isinstance(us.scale_factor, (tuple, list)) else (us.scale_factor,) * 3
            size = tuple(int(float(d) * float(s)) for d, s in zip((D, H, W), sf))   # torch: floor(input * scale_factor)
            if not (size[2] <= skip.raw.shape[-1]):
                raise AssertionError("UpsampleConvBlock5d: upsampled tensor larger than the skip tensor")
            up = Upsampled(cur, size)
            if record is not None:
                record.append(("up", cur, size))
            cur = run_block(model.checkpoint_layers[us_flag + i], us, lambda: _conv_stack(us.conv_blocks, up, skip, training, record, plan))
            if L + 1 + i in taps:
                tapped[L + 1 + i] = cur.materialise()
    top = model.top_layer
    N, C, D, H, W = cur.raw.shape
    Co = top.weight.shape[0]
    dense = torch.empty((N, Co, D, H, W), dtype=torch.float32, device=x.device)
    call("dram_conv3d_k1_fwd_lazy", _p(cur.raw), _p(cur.coef), int(cur.relu), _p(top.weight), _p(top.bias), _p(dense),
         N, C, Co, D * H * W, _stream())
    if record is not None:
        record.append(("head", cur))
    if tuple(dense.shape[-3:]) != tuple(x.shape[-3:]):
        small = dense
        dense = torch.empty((N, Co) + tuple(x.shape[-3:]), dtype=torch.float32, device=x.device)
        call("dram_upsample_trilinear_ac_fwd", _p(small), _p(dense), N, Co, D, H, W, *x.shape[-3:], _stream())
        if record is not None:
            record.append(("resize", tuple(small.shape)))
    return dense, tapped


# ------------------------------------------------------------------------------------------------ backward
def _trilinear_bwd(dy, in_shape):
    N, C, D, H, W = in_shape
    Do, Ho, Wo = dy.shape[-3:]
    dx = torch.empty(in_shape, dtype=torch.float32, device=dy.device)
    full = _lib.lib.dram_upsample_trilinear_ac_bwd_ws_bytes(N, C, D, H, W, Do, Ho, Wo)
    ws = _ws(min(full, HF.TRI_BWD_WS_CAP), dy.device) if full else None
    call("dram_upsample_trilinear_ac_bwd_ws", _p(dy), _p(dx), _p(ws), ws.numel() if ws is not None else 0,
         N, C, D, H, W, Do, Ho, Wo, _stream())
    return dx


def backward(model, record, gout, need_dx, sink=None):
    """Returns ({parameter: gradient}, dx or None).  `gout`: gradient w.r.t. the dense output.
    `sink(parameter, gradient) -> bool`: called as soon as a parameter's gradient is final -- the head first, then stage by
    stage from the last up-block to the first down-block, each right after its backward-weights launch -- so that a
    data-parallel trainer can start that gradient's all-reduce while the rest of backward still runs
    (train_step.DataParallelTrainer); a gradient the sink takes (True) is left out of the returned dict."""
    st = _stream()

    class _Grads(dict):         # parameter -> gradient (what the sink does not take)
        def __setitem__(self, p, g):
            if sink is None or not sink(p, g):
                dict.__setitem__(self, p, g)
    grads = _Grads()
    gact = {}           # id(Lazy) -> dense gradient w.r.t. the ACTIVATED tensor (accumulated over its consumers)
    g = HF._chk(gout, "DC3D grad_output", 5)
    root = record[0][1]
    for item in reversed(record):
        tag = item[0]
        if tag == "input":
            continue
        if tag == "resize":
            g = _trilinear_bwd(g, item[1])
        elif tag == "head":
            lz = item[1]
            top = model.top_layer
            N, C, D, H, W = lz.raw.shape
            Co, S = top.weight.shape[0], D * H * W
            dxa = torch.empty_like(lz.raw)
            dw = torch.empty_like(top.weight)
            db = torch.empty(Co, dtype=torch.float32, device=g.device) if top.bias is not None else None
            ws = _ws(_lib.lib.dram_conv3d_k1_bwd_ws_bytes(N, C, Co, S), g.device)
            call("dram_conv3d_k1_bwd_lazy", _p(g), _p(lz.raw), _p(lz.coef), int(lz.relu), _p(top.weight), _p(dxa), _p(dw),
                 _p(db), _p(ws), ws.numel(), N, C, Co, S, st)
            grads[top.weight] = dw
            if db is not None:
                grads[top.bias] = db
            gact[id(lz)] = dxa
            g = None
        elif tag == "conv":
            s = item[1]
            g = gact.pop(id(s.out))
            N, Co, D, H, W = s.y.shape
            S = D * H * W
            C1, C2, D2, H2, W2, oz, oy, ox = s.geom
            Ci = C1 + C2
            w = s.conv.weight
            # norm (+ReLU) backward, in place: g <- d(raw conv output)
            gamma = s.norm.weight
            dgamma = torch.empty(Co, dtype=torch.float32, device=g.device) if gamma is not None else None
            dbeta = torch.empty(Co, dtype=torch.float32, device=g.device) if s.norm.bias is not None else None
            ws = _ws(_lib.lib.dram_norm_ws_bytes(N, Co, S), g.device)
            g_in = g
            if _os.environ.get("DRAM_ENGINE_NO_INPLACE"):
                g = torch.empty_like(g_in)
            if s.sync is not None:      # "sbn": the two backward sums span the ranks; the parameter gradients stay local sums
                import torch.distributed as dist
                group, total = s.sync
                sums = torch.empty(2 * Co, dtype=torch.float64, device=g.device)
                call("dram_bn_bwd_sums", _p(g_in), _p(s.y), _p(s.mean), _p(s.rstd), _p(s.coef), _p(sums), 1, N, Co, S,
                     _p(ws), ws.numel(), st)
                if dbeta is not None:
                    dbeta.copy_(sums[0::2])
                if dgamma is not None:
                    dgamma.copy_(sums[1::2])
                dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
                call("dram_bn_bwd_apply_sums", _p(g_in), _p(s.y), _p(gamma), _p(s.mean), _p(s.rstd), _p(s.coef), _p(sums),
                     float(total), _p(g), 1, N, Co, S, _p(ws), ws.numel(), st)
            else:
                call("dram_norm_bwd", _p(g_in), _p(s.y), _p(gamma), _p(s.mean), _p(s.rstd), _p(s.coef), _p(g), _p(dgamma), _p(dbeta),
                     s.kind, s.groups, 1, int(s.batch_stats), N, Co, S, _p(ws), ws.numel(), st)
            del g_in
            if dgamma is not None:
                grads[gamma] = dgamma
            if dbeta is not None:
                grads[s.norm.bias] = dbeta
            inp, skip = s.inp, s.skip
            up = isinstance(inp, Upsampled)
            is_root = (not up) and inp is root
            need_dgrad = skip is not None or not is_root or need_dx
            ranges = s.ranges                  # as forward ran the stage (decided once, kept on the tape)
            whole = len(ranges) == 1
            lazy_ok = bool(_lib.lib.dram_conv3d_k3_wgrad_lazy_ok(N, C1, C2, Co, D, H, W)) and not _os.environ.get("DRAM_ENGINE_NO_LAZY_WGRAD")
            dw = None
            wt = HF._pack(w, 1) if need_dgrad else None
            dx2 = None
            if need_dgrad and skip is not None:
                if id(skip) in gact:
                    raise RuntimeError("fused backward: a skip tensor received a gradient before its up-path consumer")
                full = (D2, H2, W2) == (D, H, W)
                dx2 = torch.empty_like(skip.raw) if full else torch.zeros_like(skip.raw)
            g_low = None        # gradient w.r.t. the low-resolution source of an upsampled input, filled slice by slice
            if up and need_dgrad and not whole:
                g_low = torch.empty_like(inp.src.raw)
            for lo, hi in ranges:
                n = hi - lo
                vox = n * S
                gs = g if whole else g[lo:hi]
                # backward-weights: the x operand is the stage's lazy input(s); an upsampled input is produced again
                if up:
                    x1 = Lazy(inp.produce(lo, hi))
                else:
                    x1 = inp
                sk = None if skip is None else (skip if whole else _lazy_slice(skip, lo, hi))
                if not lazy_ok:     # kernels without the on-load path (odd widths, first layer): plain operands
                    x1 = Lazy(x1.materialise())
                    sk = Lazy(sk.materialise()) if sk is not None else None
                dws = torch.empty_like(w)
                wsb = _ws(_lib.lib.dram_conv3d_k3_wgrad_ws_bytes(n, Ci, Co, D, H, W), g.device)
                name = HF.conv_wgrad_kernel_name(n, (D, H, W), Co, C1, C2,
                                                 lazy=(x1.coef is not None or (sk is not None and sk.coef is not None)))
                HF._timed_call(name, 54.0 * Ci * Co * vox, 4.0 * (Ci + Co) * vox,
                               "dram_conv3d_k3_wgrad_fused", _p(x1.raw), C1, _p(x1.coef), int(x1.relu),
                               _p(sk.raw) if sk is not None else None, C2,
                               _p(sk.coef) if sk is not None else None, int(sk.relu) if sk is not None else 0,
                               D2, H2, W2, oz, oy, ox, _p(gs), _p(dws), _p(wsb), wsb.numel(), n, Co, D, H, W, st)
                dw = dws if dw is None else dw.add_(dws)
                del x1, sk, dws, wsb
                # backward-data: gradient w.r.t. the activated input(s)
                if need_dgrad:
                    dx1 = torch.empty((n, C1, D, H, W), dtype=torch.float32, device=g.device)
                    HF._timed_call(HF.conv_fwd_kernel_name((D, H, W), Ci, Co, dst_split=(C1, C2, D2, H2, W2) if dx2 is not None else None,
                                                           src=(gs, None, 0)),
                                   54.0 * Ci * Co * vox, 4.0 * (Ci + Co) * vox,
                                   "dram_conv3d_k3_fwd_ex", _p(gs), Co, None, 0, 0, 0, 0, 0, 0, 0, _p(wt), None,
                                   _p(dx1), C1, _p(dx2[lo:hi]) if dx2 is not None else None, C2, D2, H2, W2, oz, oy, ox,
                                   n, D, H, W, st)
                    if up and not whole:       # straight on to the low-resolution gradient: d(upsampled) is never whole
                        g_low[lo:hi] = _trilinear_bwd(dx1, (n,) + tuple(inp.src.raw.shape[1:]))
                    elif up:
                        gact[("up", id(inp.src))] = dx1
                    else:
                        _accumulate(gact, id(inp), dx1)
                    del dx1
            if up:
                inp.kept = None                  # last use of a kept upsampled tensor
            if g_low is not None:
                gact[("low", id(inp.src))] = g_low
            if dx2 is not None:
                gact[id(skip)] = dx2
            grads[w] = dw
            # this stage's tensors are dead from here on (its consumers ran their backward before it did)
            s.y = s.coef = s.mean = s.rstd = None
            s.out.raw = s.out.coef = None
            g = None
        elif tag == "up":
            lz, size = item[1], item[2]
            if ("low", id(lz)) in gact:      # the consumer stage ran in slices and already went through the resize
                _accumulate(gact, id(lz), gact.pop(("low", id(lz))))
            else:
                gup = gact.pop(("up", id(lz)))
                _accumulate(gact, id(lz), _trilinear_bwd(gup, tuple(lz.raw.shape)))
        elif tag == "pool":
            lz, idx, res = item[1], item[2], item[3]
            gp = gact.pop(id(res), None)
            if gp is None:
                continue
            N, C, D, H, W = lz.raw.shape
            if id(lz) in gact:      # the skip branch's gradient is already there: add the pooled branch's onto it
                call("dram_maxpool3d_2_bwd_acc", _p(gp), _p(idx), _p(gact[id(lz)]), N, C, D, H, W, st)
            else:
                dxp = torch.empty_like(lz.raw)
                call("dram_maxpool3d_2_bwd", _p(gp), _p(idx), _p(dxp), N, C, D, H, W, st)
                gact[id(lz)] = dxp
        else:   # pragma: no cover
            raise RuntimeError(f"fused backward: unknown tape entry {tag!r}")
    return grads, gact.pop(id(root), None)


def _accumulate(gact, key, t):
    if key in gact:
        gact[key].add_(t)
    else:
        gact[key] = t


# ------------------------------------------------------------------------------------------------ autograd wrapper
class DC3DFusedFn(Function):
    """(dense, *tapped feature maps) = DC3D(x) through the fused engine; `dense` is differentiable w.r.t. x and every
    parameter, the tapped maps are outputs without a gradient (the reference detaches them, models.py:556,566,578)."""

    @staticmethod
    def forward(ctx, model, taps, x, *params):
        # (torch runs Function.forward under no_grad: whether a gradient is wanted comes from needs_input_grad)
        record = [] if ctx.needs_input_grad[2] or any(ctx.needs_input_grad[3:]) else None
        out, tapped = forward(model, x, record, taps)
        ctx.model, ctx.record, ctx.params = model, record, params
        extra = tuple(tapped[t] for t in taps)
        ctx.mark_non_differentiable(*extra)
        return (out,) + extra

    @staticmethod
    @once_differentiable
    def backward(ctx, gout, *_):
        model, record, params = ctx.model, ctx.record, ctx.params
        if record is None:
            raise RuntimeError("DC3DFusedFn: backward without a recorded forward")
        try:
            grads, dx = backward(model, record, gout, ctx.needs_input_grad[2], sink=getattr(model, "grad_sink", None))
        finally:
            ctx.record = None        # free the saved activations now, not when the graph dies
        out = [None, None, dx if ctx.needs_input_grad[2] else None]
        for p, need in zip(params, ctx.needs_input_grad[3:]):
            out.append(grads.get(p) if need else None)
        return tuple(out)


def backbone_parameters(model):
    """The parameters of the U-Net proper (DC3DATGeneric carries more: its reshape convs and attention module run outside)."""
    mods = list(model.ds_modules) + [model.bg] + list(model.us_modules or []) + [model.top_layer]
    return [p for m in mods for p in m.parameters()]


def run(model, x, taps=()):
    """DC3D forward through the engine (autograd-connected when gradients are enabled).  Returns the dense output, or
    (dense, {layer id: feature map}) when `taps` are asked for."""
    taps = tuple(taps)
    params = backbone_parameters(model)
    if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params)):
        res = DC3DFusedFn.apply(model, taps, x, *params)
        dense, tapped = res[0], dict(zip(taps, res[1:]))
    else:
        with HF.cached_packs():         # inference: unchanged filters are packed once, not per call
            dense, tapped = forward(model, x, None, taps)
    return (dense, tapped) if taps else dense
