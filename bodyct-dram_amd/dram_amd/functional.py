"""torch.autograd Functions over the libdram_hip.so C ABI.

PyTorch is plumbing here: it owns the HBM buffers (caching allocator), the HIP
stream and the autograd tape; every forward / backward computation is a call
into the hand-written gfx950 kernels.  All tensors are fp32 NCDHW contiguous.
There is no fallback: CPU tensors raise.
"""
import ctypes
import math
import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib
from ._lib import call

NORM_BATCH, NORM_GROUP = 0, 1


def _stream():
    return torch.cuda.current_stream().cuda_stream


class KernelTimer:
    """Optional HIP-event timing of individual kernel launches on the stream they run on
    (used by bench.py for the roofline line).  Disabled (None) by default: zero overhead."""

    def __init__(self):
        self.records = []   # (key, flops, bytes, start_event, end_event)

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for key, flops, nbytes, e0, e1 in self.records:
            d = out.setdefault(key, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
            d["launches"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["flops"] += flops
            d["bytes"] += nbytes
        return out


TIMER = None   # set to a KernelTimer() to record


def _timed_call(key, flops, nbytes, name, *args):
    if TIMER is None:
        call(name, *args)
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    call(name, *args)
    e1.record()
    TIMER.records.append((key, flops, nbytes, e0, e1))


# kernel families of the 3x3x3 conv (include/dram_hip.h DRAM_K3_*)
K3_FWD_DIRECT, K3_FWD_WZ, K3_FWD_WZY, K3_WGRAD_DIRECT, K3_WGRAD_VEC, K3_WGRAD_WZ, K3_WGRAD_WZ_LAZY, K3_WGRAD_C1, \
    K3_FWD_C1, K3_WGRAD_WZY, K3_KINDS = range(11)


def conv_fwd_kernel_name(dhw, Cout, Cin, fused=False, dst_split=None, src=None):
    """Name of the forward / backward-data kernel instantiation the library launches for this shape, as rocprofv3
    prints it -- asked of the library itself (dram_conv3d_k3_fwd_choice_src: the same fwd_choice the launch goes through).
    `fused`: the variant with lazy operands / the statistics epilogue; `dst_split` = (C1, C2, D2, H2, W2) when the
    output is written to two tensors (backward-data of a conv whose input was a virtual concat); `src` = (x1, x2, ox): the
    source tensors of the launch (x2 the cropped second one or None, ox its window's x offset) -- a source off 16-byte
    alignment, or a crop window that is, sends a (z,y)-shaped launch to the z-only kernel."""
    c1, c2, d2, h2, w2 = dst_split if dst_split is not None else (Cout, 0, 0, 0, 0)
    sc2 = sd2 = sh2 = sw2 = sox = mis = 0
    if src is not None:
        x1, x2, sox = src
        mis = int(x1.data_ptr() % 16 != 0 or (x2 is not None and x2.data_ptr() % 16 != 0))
        if x2 is not None:
            sc2, sd2, sh2, sw2 = (int(v) for v in x2.shape[1:])
    buf = ctypes.create_string_buffer(96)
    kind = _lib.lib.dram_conv3d_k3_fwd_choice_src(Cin, Cout, dhw[0], dhw[1], dhw[2], c1, c2, d2, h2, w2, int(bool(fused)),
                                                  sc2, sd2, sh2, sw2, int(sox), mis, buf, len(buf))
    if kind < 0:
        raise _lib.DramHipError(f"dram_conv3d_k3_fwd_choice_src: {_lib.lib.dram_last_error().decode()}")
    return buf.value.decode()


def conv_wgrad_kernel_name(N, dhw, Cout, C1, C2=0, lazy=False):
    """Name of the backward-weights kernel instantiation for x = x1[N, C1] ++ crop(x2[N, C2]) (dram_conv3d_k3_wgrad_choice:
    the same wgrad_plan the launch goes through)."""
    buf = ctypes.create_string_buffer(96)
    kind = _lib.lib.dram_conv3d_k3_wgrad_choice(N, C1, C2, Cout, dhw[0], dhw[1], dhw[2], int(bool(lazy)), buf, len(buf))
    if kind < 0:
        raise _lib.DramHipError(f"dram_conv3d_k3_wgrad_choice: {_lib.lib.dram_last_error().decode()}")
    return buf.value.decode()


def conv_launch_counts():
    """Launches per 3x3x3 conv kernel family (index = K3_*) since the library was loaded."""
    arr = (ctypes.c_ulonglong * K3_KINDS)()
    call("dram_conv3d_k3_launch_counts", ctypes.cast(arr, ctypes.c_void_p), K3_KINDS)
    return list(arr)


def _p(t):
    return None if t is None else t.data_ptr()


def _chk(t, name, ndim=None):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: tensor is on {t.device}; the DRAM HIP path only runs on a ROCm device "
                           f"(there is no CPU fallback)")
    if t.dtype != torch.float32:
        raise TypeError(f"{name}: expected float32, got {t.dtype}")
    if ndim is not None and t.dim() != ndim:
        raise ValueError(f"{name}: expected {ndim}-d tensor, got shape {tuple(t.shape)}")
    return t.contiguous()


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


def crop_offsets(small, big):
    """Start offsets of crop_concat_5d's centre crop: ceil((b - a) / 2) (reference parts.py:42-44)."""
    return tuple(int(math.ceil((b - a) / 2)) for a, b in zip(small, big))


# --------------------------------------------------------------------------- conv 3x3x3
# Packed filters of inference calls: inside `with cached_packs():` (the fused engine's no-gradient branch, LobeInference) a
# filter that has not changed since it was packed (same tensor, same storage, same in-place version counter) is not packed
# again -- whole-scan inference repacked all 14 filters of the network for every scan (1.1 ms of a 27 ms scan).  Training
# never takes this path (torch runs every autograd Function's forward with gradients disabled, so the grad mode cannot be
# the switch): the optimiser bumps the version every step, and a cache of packed filters (4x the parameters) would only
# cost memory there.
import contextlib as _contextlib
import weakref as _weakref
_PACK_CACHE_ON = 0


@_contextlib.contextmanager
def cached_packs():
    global _PACK_CACHE_ON
    _PACK_CACHE_ON += 1
    try:
        yield
    finally:
        _PACK_CACHE_ON -= 1


_PACK_CACHE = {}      # id(weight tensor) -> (weak reference to it, {mode: (data_ptr, version, packed)}); (a WeakKeyDictionary would
                      # compare tensors with ==, i.e. elementwise)


def _pack_cached(w):
    """The cache entry of tensor `w` ({mode: ...}) or None."""
    ent = _PACK_CACHE.get(id(w))
    return ent[1] if ent is not None and ent[0]() is w else None


def _pack(w, mode):
    co, ci = w.shape[0], w.shape[1]
    cacheable = _PACK_CACHE_ON > 0
    if cacheable:
        modes = _pack_cached(w)
        hit = modes.get(mode) if modes is not None else None
        if hit is not None and hit[0] == w.data_ptr() and hit[1] == w._version and hit[2].device == w.device:
            return hit[2]
    wt = torch.empty(_lib.lib.dram_conv3d_k3_packed_floats(co, ci), dtype=torch.float32, device=w.device)
    call("dram_conv3d_k3_pack_weights", _p(w), _p(wt), co, ci, mode, _stream())
    if cacheable:
        modes = _pack_cached(w)
        if modes is None:
            key = id(w)
            try:
                ref = _weakref.ref(w, lambda _r, key=key: _PACK_CACHE.pop(key, None))
            except TypeError:       # (a tensor subclass without weak references: no cache)
                return wt
            modes = {}
            _PACK_CACHE[key] = (ref, modes)
        modes[mode] = (w.data_ptr(), w._version, wt)
    elif _pack_cached(w) is not None:
        del _PACK_CACHE[id(w)]      # a training call: drop what an earlier evaluation pass left
    return wt


class Conv3dK3Fn(Function):
    """y = conv3d(cat(x1, crop(x2)), w, bias), k=3, stride 1, zero pad 1.  x2 may be None.
    (nn.Conv3d of reference parts.py:95,105,133,142,177,185; crop_concat_5d of parts.py:153 fused.)"""

    @staticmethod
    def forward(ctx, x1, x2, w, bias):
        x1 = _chk(x1, "conv3d input", 5)
        w = _chk(w, "conv3d weight", 5)
        N, C1, D, H, W = x1.shape
        Co, Ci = w.shape[0], w.shape[1]
        if tuple(w.shape[2:]) != (3, 3, 3):
            raise ValueError(f"Conv3dK3Fn: kernel {tuple(w.shape[2:])} is not 3x3x3")
        if x2 is not None:
            x2 = _chk(x2, "conv3d second input", 5)
            C2, D2, H2, W2 = x2.shape[1:]
            if x2.shape[0] != N:
                raise ValueError("conv3d: batch sizes of the two inputs differ")
            if not (D <= D2 and H <= H2 and W <= W2):
                raise ValueError("conv3d: the second (skip) tensor must be at least as large as the first")
            oz, oy, ox = crop_offsets((D, H, W), (D2, H2, W2))
        else:
            C2 = D2 = H2 = W2 = oz = oy = ox = 0
        if C1 + C2 != Ci:
            raise ValueError(f"conv3d: input has {C1 + C2} channels, weight expects {Ci}")
        if bias is not None:
            bias = _chk(bias, "conv3d bias", 1)
        wt = _pack(w, 0)
        y = torch.empty((N, Co, D, H, W), dtype=torch.float32, device=x1.device)
        vox = N * D * H * W
        _timed_call(conv_fwd_kernel_name((D, H, W), Co, Ci, src=(x1, x2, ox)), 54.0 * Ci * Co * vox, 4.0 * (Ci + Co) * vox,
                    "dram_conv3d_k3_fwd_ex", _p(x1), C1, _p(x2), C2, D2, H2, W2, oz, oy, ox, _p(wt), _p(bias),
                    _p(y), Co, None, 0, 0, 0, 0, 0, 0, 0, N, D, H, W, _stream())
        ctx.save_for_backward(x1, x2, w)
        ctx.has_bias = bias is not None
        ctx.geom = (C2, D2, H2, W2, oz, oy, ox)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x1, x2, w = ctx.saved_tensors
        dy = _chk(dy, "conv3d grad_output", 5)
        N, C1, D, H, W = x1.shape
        Co, Ci = w.shape[0], w.shape[1]
        C2, D2, H2, W2, oz, oy, ox = ctx.geom
        st = _stream()
        dx1 = dx2 = dw = db = None
        need1 = ctx.needs_input_grad[0]
        need2 = x2 is not None and ctx.needs_input_grad[1]
        if need1 or need2:
            wt = _pack(w, 1)   # filter of the transposed conv: [27][Co][Ci]
            dx1 = torch.empty_like(x1)
            if x2 is not None:
                full = (D2, H2, W2) == (D, H, W)
                dx2 = torch.empty_like(x2) if full else torch.zeros_like(x2)
            vox = N * D * H * W
            _timed_call(conv_fwd_kernel_name((D, H, W), Ci, Co, dst_split=(C1, C2, D2, H2, W2) if x2 is not None else None,
                                             src=(dy, None, 0)), 54.0 * Ci * Co * vox, 4.0 * (Ci + Co) * vox,
                        "dram_conv3d_k3_fwd_ex", _p(dy), Co, None, 0, 0, 0, 0, 0, 0, 0, _p(wt), None,
                        _p(dx1), C1, _p(dx2), C2, D2, H2, W2, oz, oy, ox, N, D, H, W, st)
            if not need1:
                dx1 = None
            if not need2:
                dx2 = None
        if ctx.needs_input_grad[2]:
            dw = torch.empty_like(w)
            nbytes = _lib.lib.dram_conv3d_k3_wgrad_ws_bytes(N, Ci, Co, D, H, W)
            ws = _ws(nbytes, dy.device)
            vox = N * D * H * W
            _timed_call(conv_wgrad_kernel_name(N, (D, H, W), Co, C1, C2), 54.0 * Ci * Co * vox, 4.0 * (Ci + Co) * vox,
                        "dram_conv3d_k3_wgrad_ex", _p(x1), C1, _p(x2), C2, D2, H2, W2, oz, oy, ox, _p(dy), _p(dw),
                        _p(ws), ws.numel(), N, Co, D, H, W, st)
        if ctx.has_bias and ctx.needs_input_grad[3]:
            S = D * H * W
            db = torch.empty(Co, dtype=torch.float32, device=dy.device)
            ws = _ws(_lib.lib.dram_channel_sum_ws_bytes(N, Co, S), dy.device)
            call("dram_channel_sum", _p(dy), _p(db), _p(ws), ws.numel(), N, Co, S, st)
        return dx1, dx2, dw, db


def conv3d_k3(x, w, bias=None, skip=None):
    return Conv3dK3Fn.apply(x, skip, w, bias)


# --------------------------------------------------------------------------- conv 1x1x1 (+bias)
class Conv3dK1Fn(Function):
    """The regression head top_layer = nn.Conv3d(C, out_ch, 1) (reference models.py:109-110,145)."""

    @staticmethod
    def forward(ctx, x, w, bias):
        x = _chk(x, "conv1x1 input", 5)
        w = _chk(w, "conv1x1 weight", 5)
        N, Ci, D, H, W = x.shape
        Co = w.shape[0]
        if w.shape[1] != Ci or tuple(w.shape[2:]) != (1, 1, 1):
            raise ValueError(f"conv1x1: weight {tuple(w.shape)} does not match input channels {Ci}")
        if bias is not None:
            bias = _chk(bias, "conv1x1 bias", 1)
        y = torch.empty((N, Co, D, H, W), dtype=torch.float32, device=x.device)
        call("dram_conv3d_k1_fwd", _p(x), _p(w), _p(bias), _p(y), N, Ci, Co, D * H * W, _stream())
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = _chk(dy, "conv1x1 grad_output", 5)
        N, Ci, D, H, W = x.shape
        Co, S = w.shape[0], D * H * W
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw = torch.empty_like(w) if ctx.needs_input_grad[1] else None
        db = torch.empty(Co, dtype=torch.float32, device=x.device) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        ws = _ws(_lib.lib.dram_conv3d_k1_bwd_ws_bytes(N, Ci, Co, S), x.device)
        call("dram_conv3d_k1_bwd", _p(dy), _p(x), _p(w), _p(dx), _p(dw), _p(db), _p(ws), ws.numel(),
             N, Ci, Co, S, _stream())
        return dx, dw, db


def conv3d_k1(x, w, bias=None):
    return Conv3dK1Fn.apply(x, w, bias)


# --------------------------------------------------------------------------- norm (+ReLU)
class NormActFn(Function):
    """BatchNorm3d / GroupNorm with optional fused ReLU (normal_wrapper + act_wrapper,
    reference parts.py:17-35,48-54).  `running_mean/var` are updated in place when given
    (training BatchNorm with track_running_stats)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, kind, groups, use_batch_stats, momentum, eps, relu):
        x = _chk(x, "norm input")
        if x.dim() < 3:
            raise ValueError("norm: expected (N, C, *spatial)")
        N, C = x.shape[0], x.shape[1]
        S = x.numel() // (N * C)
        dev = x.device
        nstat = C if kind == NORM_BATCH else N * groups
        y = torch.empty_like(x)
        save_mean = torch.empty(nstat, dtype=torch.float32, device=dev)
        save_rstd = torch.empty(nstat, dtype=torch.float32, device=dev)
        rowcoef = torch.empty(2 * N * C, dtype=torch.float32, device=dev)
        st = _stream()
        if gamma is not None:
            gamma = _chk(gamma, "norm weight", 1)
        if beta is not None:
            beta = _chk(beta, "norm bias", 1)
        if use_batch_stats:
            ws = _ws(_lib.lib.dram_norm_ws_bytes(N, C, S), dev)
            call("dram_norm_fwd_train", _p(x), _p(gamma), _p(beta), _p(y), _p(save_mean), _p(save_rstd), _p(rowcoef),
                 _p(running_mean), _p(running_var), float(momentum), float(eps), kind, groups, int(relu),
                 N, C, S, _p(ws), ws.numel(), st)
        else:
            call("dram_bn_fwd_eval", _p(x), _p(gamma), _p(beta), _p(running_mean), _p(running_var), _p(y),
                 _p(save_mean), _p(save_rstd), _p(rowcoef), float(eps), int(relu), N, C, S, st)
        ctx.save_for_backward(x, gamma, save_mean, save_rstd, rowcoef)
        ctx.cfg = (kind, groups, bool(use_batch_stats), bool(relu), beta is not None)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, gamma, save_mean, save_rstd, rowcoef = ctx.saved_tensors
        kind, groups, batch_stats, relu, has_beta = ctx.cfg
        dy = _chk(dy, "norm grad_output")
        N, C = x.shape[0], x.shape[1]
        S = x.numel() // (N * C)
        dev = x.device
        dx = torch.empty_like(x)
        dgamma = torch.empty(C, dtype=torch.float32, device=dev) if gamma is not None else None
        dbeta = torch.empty(C, dtype=torch.float32, device=dev) if has_beta else None
        ws = _ws(_lib.lib.dram_norm_ws_bytes(N, C, S), dev)
        call("dram_norm_bwd", _p(dy), _p(x), _p(gamma), _p(save_mean), _p(save_rstd), _p(rowcoef), _p(dx),
             _p(dgamma), _p(dbeta), kind, groups, int(relu), int(batch_stats), N, C, S, _p(ws), ws.numel(), _stream())
        return dx, dgamma, dbeta, None, None, None, None, None, None, None, None


def norm_act(x, gamma, beta, running_mean, running_var, kind, groups, use_batch_stats, momentum, eps, relu):
    return NormActFn.apply(x, gamma, beta, running_mean, running_var, kind, groups, use_batch_stats, momentum, eps, relu)


class SyncBatchNormFn(Function):
    """Training-mode BatchNorm whose statistics span all ranks of a process group (nn.SyncBatchNorm, the
    reference's normal_wrapper "sbn", parts.py:32-33, under data parallelism).  Two exchanges per layer and
    direction, 2*C doubles each: all-gather of {mean, M2, count} in forward (combined with Chan's formula in
    fp64), all-reduce of {sum dy', sum dy'*xhat} in backward -- like torch's SyncBatchNorm, the parameter
    gradients stay local sums and are averaged by the data-parallel gradient all-reduce."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, momentum, eps, relu, group):
        import torch.distributed as dist
        x = _chk(x, "sync-bn input")
        N, C = x.shape[0], x.shape[1]
        S = x.numel() // (N * C)
        dev = x.device
        st = _stream()
        ws = _ws(_lib.lib.dram_norm_ws_bytes(N, C, S), dev)
        local = torch.empty(2 * C + 1, dtype=torch.float64, device=dev)
        call("dram_bn_stats", _p(x), _p(local), N, C, S, _p(ws), ws.numel(), st)
        local[2 * C] = float(N * S)
        world = dist.get_world_size(group)
        allst = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(allst, local, group=group)
        allst = torch.stack(allst)                                   # [world, 2C+1]
        cnt = allst[:, 2 * C].view(world, 1)
        means, m2s = allst[:, 0:2 * C:2], allst[:, 1:2 * C:2]
        total = cnt.sum()
        mean = (means * cnt).sum(0) / total
        m2 = (m2s + cnt * (means - mean) ** 2).sum(0)                # Chan's parallel combine
        var = m2 / total                                             # biased
        mean_f, var_f = mean.float(), var.float()
        if running_mean is not None:
            with torch.no_grad():
                unb = (m2 / (total - 1.0)).float() if float(total) > 1.0 else var_f
                running_mean.mul_(1.0 - momentum).add_(mean_f, alpha=momentum)
                running_var.mul_(1.0 - momentum).add_(unb, alpha=momentum)
        y = torch.empty_like(x)
        save_mean = torch.empty(C, dtype=torch.float32, device=dev)
        save_rstd = torch.empty(C, dtype=torch.float32, device=dev)
        rowcoef = torch.empty(2 * N * C, dtype=torch.float32, device=dev)
        if gamma is not None:
            gamma = _chk(gamma, "sync-bn weight", 1)
        if beta is not None:
            beta = _chk(beta, "sync-bn bias", 1)
        call("dram_bn_fwd_eval", _p(x), _p(gamma), _p(beta), _p(mean_f), _p(var_f), _p(y), _p(save_mean), _p(save_rstd),
             _p(rowcoef), float(eps), int(relu), N, C, S, st)
        ctx.save_for_backward(x, gamma, save_mean, save_rstd, rowcoef)
        ctx.cfg = (bool(relu), beta is not None, float(total), group)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        import torch.distributed as dist
        x, gamma, save_mean, save_rstd, rowcoef = ctx.saved_tensors
        relu, has_beta, total, group = ctx.cfg
        dy = _chk(dy, "sync-bn grad_output")
        N, C = x.shape[0], x.shape[1]
        S = x.numel() // (N * C)
        dev = x.device
        st = _stream()
        ws = _ws(_lib.lib.dram_norm_ws_bytes(N, C, S), dev)
        sums = torch.empty(2 * C, dtype=torch.float64, device=dev)
        call("dram_bn_bwd_sums", _p(dy), _p(x), _p(save_mean), _p(save_rstd), _p(rowcoef), _p(sums), int(relu), N, C, S,
             _p(ws), ws.numel(), st)
        dbeta = sums[0::2].float() if has_beta else None
        dgamma = sums[1::2].float() if gamma is not None else None
        gsums = sums.clone()
        dist.all_reduce(gsums, op=dist.ReduceOp.SUM, group=group)
        dx = torch.empty_like(x)
        call("dram_bn_bwd_apply_sums", _p(dy), _p(x), _p(gamma), _p(save_mean), _p(save_rstd), _p(rowcoef), _p(gsums),
             float(total), _p(dx), int(relu), N, C, S, _p(ws), ws.numel(), st)
        return dx, dgamma, dbeta, None, None, None, None, None, None


def sync_batch_norm(x, gamma, beta, running_mean, running_var, momentum, eps, relu, group=None):
    return SyncBatchNormFn.apply(x, gamma, beta, running_mean, running_var, momentum, eps, relu, group)


class ReLUFn(Function):
    @staticmethod
    def forward(ctx, x):
        x = _chk(x, "relu input")
        y = torch.empty_like(x)
        call("dram_relu_fwd", _p(x), _p(y), x.numel(), _stream())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = _chk(dy, "relu grad_output")
        dx = torch.empty_like(dy)
        call("dram_relu_bwd", _p(dy), _p(y), _p(dx), dy.numel(), _stream())
        return dx


def relu(x):
    return ReLUFn.apply(x)


# --------------------------------------------------------------------------- max pool 2x2x2
class PReLUFn(Function):
    """nn.PReLU (act_wrapper "prelu", reference parts.py:51-52)."""

    @staticmethod
    def forward(ctx, x, a):
        x, a = _chk(x, "prelu input"), _chk(a, "prelu weight", 1)
        if x.dim() < 2:
            raise ValueError("prelu: expected (N, C, *spatial)")
        N, C = x.shape[0], x.shape[1]
        if a.numel() not in (1, C):
            raise ValueError(f"prelu: {a.numel()} parameters for {C} channels")
        y = torch.empty_like(x)
        call("dram_prelu_fwd", _p(x), _p(a), _p(y), N, C, a.numel(), x.numel() // (N * C), _stream())
        ctx.save_for_backward(x, a)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, a = ctx.saved_tensors
        dy = _chk(dy, "prelu grad_output")
        N, C = x.shape[0], x.shape[1]
        S = x.numel() // (N * C)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        da = torch.empty_like(a)
        ws = _ws(_lib.lib.dram_prelu_bwd_ws_bytes(N, C, S), x.device)
        call("dram_prelu_bwd", _p(dy), _p(x), _p(a), _p(dx), _p(da), _p(ws), ws.numel(), N, C, a.numel(), S, _stream())
        return dx, da


def prelu(x, a):
    return PReLUFn.apply(x, a)


class GlobalMaxFn(Function):
    """F.adaptive_max_pool3d(x, 1).view(B, C) (reference models.py:41-42)."""

    @staticmethod
    def forward(ctx, x):
        x = _chk(x, "global max input")
        B, C = x.shape[0], x.shape[1]
        S = x.numel() // (B * C)
        out = torch.empty((B, C), dtype=torch.float32, device=x.device)
        idx = torch.empty((B, C), dtype=torch.int64, device=x.device)
        call("dram_global_max_fwd", _p(x), _p(out), _p(idx), B * C, S, _stream())
        ctx.save_for_backward(idx)
        ctx.shape = tuple(x.shape)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        dout = _chk(dout, "global max grad_output")
        dx = torch.empty(ctx.shape, dtype=torch.float32, device=dout.device)
        B, C = ctx.shape[0], ctx.shape[1]
        call("dram_global_max_bwd", _p(dout), _p(idx), _p(dx), B * C, dx.numel() // (B * C), _stream())
        return dx


def global_max(x):
    return GlobalMaxFn.apply(x)


class MaxPool2Fn(Function):
    """nn.MaxPool3d(2, 2, 0) (reference parts.py:191)."""

    @staticmethod
    def forward(ctx, x):
        x = _chk(x, "maxpool input", 5)
        N, C, D, H, W = x.shape
        out = torch.empty((N, C, D // 2, H // 2, W // 2), dtype=torch.float32, device=x.device)
        idx = torch.empty(out.shape, dtype=torch.uint8, device=x.device)
        call("dram_maxpool3d_2_fwd", _p(x), _p(out), _p(idx), N, C, D, H, W, _stream())
        ctx.save_for_backward(idx)
        ctx.in_shape = tuple(x.shape)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        dout = _chk(dout, "maxpool grad_output", 5)
        N, C, D, H, W = ctx.in_shape
        dx = torch.empty(ctx.in_shape, dtype=torch.float32, device=dout.device)
        call("dram_maxpool3d_2_bwd", _p(dout), _p(idx), _p(dx), N, C, D, H, W, _stream())
        return dx


def max_pool3d_2(x):
    return MaxPool2Fn.apply(x)


# --------------------------------------------------------------------------- trilinear resize
# workspace cap of the two-stage trilinear backward (the z-reduced intermediate of a group of planes)
TRI_BWD_WS_CAP = int(os.environ.get("DRAM_TRI_WS_MB", "1024")) << 20


class TrilinearACFn(Function):
    """nn.Upsample(mode='trilinear', align_corners=True) (reference parts.py:149, models.py:146)."""

    @staticmethod
    def forward(ctx, x, size):
        x = _chk(x, "upsample input", 5)
        N, C, D, H, W = x.shape
        Do, Ho, Wo = (int(s) for s in size)
        y = torch.empty((N, C, Do, Ho, Wo), dtype=torch.float32, device=x.device)
        call("dram_upsample_trilinear_ac_fwd", _p(x), _p(y), N, C, D, H, W, Do, Ho, Wo, _stream())
        ctx.shapes = (N, C, D, H, W, Do, Ho, Wo)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        N, C, D, H, W, Do, Ho, Wo = ctx.shapes
        dy = _chk(dy, "upsample grad_output", 5)
        dx = torch.empty((N, C, D, H, W), dtype=torch.float32, device=dy.device)
        full = _lib.lib.dram_upsample_trilinear_ac_bwd_ws_bytes(N, C, D, H, W, Do, Ho, Wo)
        ws = _ws(min(full, TRI_BWD_WS_CAP), dy.device) if full else None
        call("dram_upsample_trilinear_ac_bwd_ws", _p(dy), _p(dx), _p(ws), ws.numel() if ws is not None else 0,
             N, C, D, H, W, Do, Ho, Wo, _stream())
        return dx, None


def upsample_trilinear_ac(x, size=None, scale_factor=None):
    if size is None:
        if scale_factor is None:
            raise ValueError("either size or scale_factor should be defined")
        sf = scale_factor if isinstance(scale_factor, (tuple, list)) else (scale_factor,) * 3
        # torch: output = floor(input * scale_factor)
        size = tuple(int(math.floor(float(d) * float(s))) for d, s in zip(x.shape[2:], sf))
    elif not isinstance(size, (tuple, list, torch.Size)):
        size = (size,) * 3
    return TrilinearACFn.apply(x, tuple(size))


# --------------------------------------------------------------------------- crop + concat
class CropConcatFn(Function):
    """crop_concat_5d (reference parts.py:37-46)."""

    @staticmethod
    def forward(ctx, t1, t2):
        t1 = _chk(t1, "crop_concat t1", 5)
        t2 = _chk(t2, "crop_concat t2", 5)
        N, C1, D, H, W = t1.shape
        C2, D2, H2, W2 = t2.shape[1:]
        oz, oy, ox = crop_offsets((D, H, W), (D2, H2, W2))
        out = torch.empty((N, C1 + C2, D, H, W), dtype=torch.float32, device=t1.device)
        call("dram_crop_concat_fwd", _p(t1), _p(t2), _p(out), N, C1, C2, D, H, W, D2, H2, W2, oz, oy, ox, _stream())
        ctx.geom = (N, C1, C2, D, H, W, D2, H2, W2, oz, oy, ox)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        N, C1, C2, D, H, W, D2, H2, W2, oz, oy, ox = ctx.geom
        dout = _chk(dout, "crop_concat grad_output", 5)
        dt1 = torch.empty((N, C1, D, H, W), dtype=torch.float32, device=dout.device) if ctx.needs_input_grad[0] else None
        dt2 = torch.empty((N, C2, D2, H2, W2), dtype=torch.float32, device=dout.device) if ctx.needs_input_grad[1] else None
        call("dram_crop_concat_bwd", _p(dout), _p(dt1), _p(dt2), N, C1, C2, D, H, W, D2, H2, W2, oz, oy, ox, _stream())
        return dt1, dt2


def crop_concat(t1, t2):
    return CropConcatFn.apply(t1, t2)


# --------------------------------------------------------------------------- lobe-masked mean
class MaskedMeanFn(Function):
    """sum(x*m)/sum(m) per (n, c) (pooling_dense_features default branch, reference models.py:45-47)."""

    @staticmethod
    def forward(ctx, x, mask):
        x = _chk(x, "masked_mean input", 5)
        mask = _chk(mask, "masked_mean mask", 5)
        N, C = x.shape[:2]
        S = x.numel() // (N * C)
        if mask.shape[0] != N or mask.shape[1] != 1 or mask.numel() != N * S:
            raise ValueError(f"masked_mean: mask {tuple(mask.shape)} does not broadcast over {tuple(x.shape)}")
        out = torch.empty((N, C), dtype=torch.float32, device=x.device)
        msum = torch.empty(N, dtype=torch.float32, device=x.device)
        ws = _ws(_lib.lib.dram_masked_mean_ws_bytes(N, C, S), x.device)
        call("dram_masked_mean_fwd", _p(x), _p(mask), _p(out), _p(msum), _p(ws), ws.numel(), N, C, S, _stream())
        ctx.save_for_backward(mask, msum)
        ctx.shape = tuple(x.shape)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        mask, msum = ctx.saved_tensors
        dout = _chk(dout, "masked_mean grad_output", 2)
        N, C = ctx.shape[:2]
        S = mask.numel() // N
        dx = torch.empty(ctx.shape, dtype=torch.float32, device=dout.device)
        call("dram_masked_mean_bwd", _p(dout), _p(mask), _p(msum), _p(dx), N, C, S, _stream())
        return dx, None


def masked_mean(x, mask):
    return MaskedMeanFn.apply(x, mask)


# --------------------------------------------------------------------------- fused IntRegRefineLoss
class IntRegRefineLossFn(Function):
    """(reg_loss, seg_loss) of IntRegRefineLoss.__call__ (reference dram/metrics.py:360-373): one streaming
    pass for every sum, one for the gradient.  `refined` is the model's second output (None when the
    model returns the same tensor twice, as DC3D does)."""

    @staticmethod
    def forward(ctx, dense, refined, lobes, lesions, keep, targets, weight, smoothing):
        dense = _chk(dense, "loss dense", 5)
        lobes = _chk(lobes, "loss lobes", 5)
        lesions = _chk(lesions, "loss lesions", 5)
        if refined is not None:
            refined = _chk(refined, "loss refined", 5)
        N = dense.shape[0]
        S = dense.numel() // N
        if dense.shape[1] != 1 or lobes.shape != dense.shape or lesions.shape != dense.shape \
                or (refined is not None and refined.shape != dense.shape):
            raise ValueError("IntRegRefineLossFn: dense / refined / lobes / lesions must all be [N,1,D,H,W]")
        keep, targets, weight = _chk(keep, "loss keep").reshape(-1), _chk(targets, "loss targets"), _chk(weight, "loss weight")
        if keep.numel() != N or targets.numel() != 2 * N or weight.numel() != N:
            raise ValueError("IntRegRefineLossFn: keep[N], targets[N,2], weight[N] expected")
        dev = dense.device
        out = torch.empty(2, dtype=torch.float32, device=dev)
        state = torch.empty(_lib.lib.dram_intreg_loss_state_floats(N), dtype=torch.float32, device=dev)
        ws = _ws(_lib.lib.dram_intreg_loss_ws_bytes(N, S), dev)
        call("dram_intreg_loss_fwd", _p(dense), _p(refined), _p(lobes), _p(lesions), _p(keep), _p(targets), _p(weight),
             float(smoothing), _p(out), _p(state), _p(ws), ws.numel(), N, S, _stream())
        ctx.save_for_backward(dense, lobes, lesions, keep, targets, weight, state, *(() if refined is None else (refined,)))
        ctx.smoothing = float(smoothing)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        dense, lobes, lesions, keep, targets, weight, state, *rest = ctx.saved_tensors
        refined = rest[0] if rest else None
        gout = _chk(gout, "loss grad_output", 1)
        N = dense.shape[0]
        S = dense.numel() // N
        ddense = torch.empty_like(dense)
        drefined = torch.empty_like(refined) if refined is not None else None
        call("dram_intreg_loss_bwd", _p(dense), _p(refined), _p(lobes), _p(lesions), _p(keep), _p(targets), _p(weight),
             _p(state), _p(gout), ctx.smoothing, _p(ddense), _p(drefined), N, S, _stream())
        return ddense, drefined, None, None, None, None, None, None


def intreg_refine_loss(dense, lobes, lesions, keep, targets, weight, smoothing=0.1, refined=None):
    """Returns a [2] tensor: (reg_loss, seg_loss)."""
    if refined is dense:
        refined = None
    return IntRegRefineLossFn.apply(dense, refined, lobes, lesions, keep, targets, weight, smoothing)


class SigmoidFn(Function):
    """F.sigmoid as a differentiable device op (the affine-consistency term compares probabilities,
    reference dram/metrics.py:434,445)."""

    @staticmethod
    def forward(ctx, x):
        x = _chk(x, "sigmoid input")
        y = torch.empty_like(x)
        call("dram_sigmoid_fwd", _p(x), _p(y), x.numel(), _stream())
        ctx.save_for_backward(x)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = _chk(dy, "sigmoid grad_output")
        dx = torch.empty_like(x)
        call("dram_sigmoid_bwd", _p(dy), _p(x), _p(dx), x.numel(), _stream())
        return dx


def sigmoid(x):
    return SigmoidFn.apply(x)


class MaskedSmoothL1Fn(Function):
    """F.smooth_l1_loss(a[m > 0], b[m > 0]) with m [N,1,D,H,W] expanded over the channels of a, b [N,C,D,H,W]
    (reference dram/metrics.py:448-452)."""

    @staticmethod
    def forward(ctx, a, b, mask):
        a, b, mask = _chk(a, "smooth_l1 input", 5), _chk(b, "smooth_l1 target", 5), _chk(mask, "smooth_l1 mask", 5)
        N, C = a.shape[:2]
        S = a.numel() // (N * C)
        if b.shape != a.shape or mask.shape[0] != N or mask.shape[1] != 1 or mask.numel() != N * S:
            raise ValueError(f"masked smooth-L1: shapes {tuple(a.shape)}, {tuple(b.shape)}, mask {tuple(mask.shape)}")
        out = torch.empty(2, dtype=torch.float32, device=a.device)
        ws = _ws(_lib.lib.dram_masked_smooth_l1_ws_bytes(N, C, S), a.device)
        call("dram_masked_smooth_l1_fwd", _p(a), _p(b), _p(mask), _p(out), _p(ws), ws.numel(), N, C, S, _stream())
        ctx.save_for_backward(a, b, mask, out)
        return out[0]

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        a, b, mask, out = ctx.saved_tensors
        N, C = a.shape[:2]
        S = a.numel() // (N * C)
        g = g.reshape(1).contiguous()
        da = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        db = torch.empty_like(b) if ctx.needs_input_grad[1] else None
        if da is not None or db is not None:
            call("dram_masked_smooth_l1_bwd", _p(a), _p(b), _p(mask), _p(out), _p(g), _p(da), _p(db), N, C, S, _stream())
        return da, db, None


def masked_smooth_l1(a, b, mask):
    return MaskedSmoothL1Fn.apply(a, b, mask)


# --------------------------------------------------------------------------- PCM local attention
PCM_RELU, PCM_L2NORM = 1, 2
# merge_type -> (flags, scale_mode); reference models.py:259-331 (dot-product family)
PCM_MERGE_MODES = {
    "sm": (0, 0), "scaled_dot_product": (0, 1), "scaled_dot_product_relu": (PCM_RELU, 1), "smrelu": (PCM_RELU, 0),
    "smscaled": (0, 2), "l2sm": (PCM_L2NORM, 0), "l2smrelu": (PCM_RELU | PCM_L2NORM, 0),
}


def _offsets_arg(offsets):
    flat = [int(v) for o in offsets for v in o]
    return (ctypes.c_int * len(flat))(*flat), len(flat) // 3


class PcmAttentionFn(Function):
    """attn[b,e,i] = softmax over node i's in-grid neighbours e of the merged theta_i . phi_(i+o_e)
    (reference models.py: merge_func 259-331 inside compute_cross_x 365-397)."""

    @staticmethod
    def forward(ctx, theta, phi, offsets, flags, scale_mode, f_relu=None):
        theta, phi = _chk(theta, "pcm theta", 5), _chk(phi, "pcm phi", 5)
        if theta.shape != phi.shape:
            raise ValueError(f"pcm attention: theta {tuple(theta.shape)} and phi {tuple(phi.shape)} differ")
        B, Fd, D, H, W = theta.shape
        f_relu = Fd if f_relu is None else int(f_relu)      # feature planes inside the activation (the rest is added raw)
        arr, E = _offsets_arg(offsets)
        attn = torch.empty((B, E, D, H, W), dtype=torch.float32, device=theta.device)
        call("dram_pcm_attention_split_fwd", _p(theta), _p(phi), arr, E, flags, scale_mode, f_relu, _p(attn), B, Fd, D, H, W,
             _stream())
        ctx.save_for_backward(theta, phi, attn)
        ctx.cfg = (offsets, flags, scale_mode, f_relu)
        return attn

    @staticmethod
    @once_differentiable
    def backward(ctx, dattn):
        theta, phi, attn = ctx.saved_tensors
        offsets, flags, scale_mode, f_relu = ctx.cfg
        dattn = _chk(dattn, "pcm attention grad_output", 5)
        B, Fd, D, H, W = theta.shape
        arr, E = _offsets_arg(offsets)
        ds = torch.empty_like(attn)
        ds2 = torch.empty_like(attn) if (f_relu < Fd and (flags & PCM_RELU)) else None
        dtheta, dphi = torch.empty_like(theta), torch.empty_like(phi)
        call("dram_pcm_attention_split_bwd", _p(theta), _p(phi), _p(attn), _p(dattn), arr, E, flags, scale_mode, f_relu, _p(ds),
             _p(ds2), _p(dtheta), _p(dphi), B, Fd, D, H, W, _stream())
        return dtheta, dphi, None, None, None, None


# merge types normalised by the sum over a node's edges instead of a softmax (reference models.py:300-302, 307-320)
PCM_SUM_MERGES = {"cosine": 0, "heu1": 1, "heu2": 2}


class PcmAttentionSumFn(Function):
    """attn[b,e,i] = v_e / (eps + sum_k v_k) over node i's in-grid neighbours with v = the cosine similarity of
    (theta_i, phi_(i+o_e)) / the heuristic similarity theta.phi / (1 + |theta - phi|_1), masked below 0.03 (heu1: formed
    under no_grad in the reference, so that attention carries no gradient at all) or rectified (heu2)."""

    @staticmethod
    def forward(ctx, theta, phi, offsets, mode):
        theta, phi = _chk(theta, "pcm theta", 5), _chk(phi, "pcm phi", 5)
        if theta.shape != phi.shape:
            raise ValueError(f"pcm attention: theta {tuple(theta.shape)} and phi {tuple(phi.shape)} differ")
        B, Fd, D, H, W = theta.shape
        arr, E = _offsets_arg(offsets)
        attn = torch.empty((B, E, D, H, W), dtype=torch.float32, device=theta.device)
        call("dram_pcm_attention_sum_fwd", _p(theta), _p(phi), arr, E, mode, _p(attn), B, Fd, D, H, W, _stream())
        ctx.save_for_backward(theta, phi, attn)
        ctx.cfg = (offsets, mode)
        return attn

    @staticmethod
    @once_differentiable
    def backward(ctx, dattn):
        theta, phi, attn = ctx.saved_tensors
        offsets, mode = ctx.cfg
        if mode == PCM_SUM_MERGES["heu1"]:
            # reference models.py:311-314: `f = f * mask_f` sits inside `with torch.no_grad():`, so the masked similarities --
            # and the whole heu1 attention -- are constants of the graph: no gradient reaches theta or phi
            return None, None, None, None
        dattn = _chk(dattn, "pcm attention grad_output", 5)
        B, Fd, D, H, W = theta.shape
        arr, E = _offsets_arg(offsets)
        ds = torch.empty_like(attn)
        dtheta, dphi = torch.empty_like(theta), torch.empty_like(phi)
        call("dram_pcm_attention_sum_bwd", _p(theta), _p(phi), _p(attn), _p(dattn), arr, E, mode, _p(ds), _p(dtheta), _p(dphi),
             B, Fd, D, H, W, _stream())
        return dtheta, dphi, None, None


class PcmAggregateFn(Function):
    """out[b,c,i] = sum_e attn[b,e,i] * v[b,c,i+o_e]  (torch.matmul(f_sm, x_g), reference models.py:394)."""

    @staticmethod
    def forward(ctx, attn, v, offsets):
        attn, v = _chk(attn, "pcm attn", 5), _chk(v, "pcm values", 5)
        B, C, D, H, W = v.shape
        arr, E = _offsets_arg(offsets)
        if tuple(attn.shape) != (B, E, D, H, W):
            raise ValueError(f"pcm aggregate: attn {tuple(attn.shape)} does not match values {tuple(v.shape)} x {E} offsets")
        out = torch.empty_like(v)
        call("dram_pcm_aggregate_fwd", _p(attn), _p(v), arr, E, _p(out), B, C, D, H, W, _stream())
        ctx.save_for_backward(attn, v)
        ctx.offsets = offsets
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        attn, v = ctx.saved_tensors
        dout = _chk(dout, "pcm aggregate grad_output", 5)
        B, C, D, H, W = v.shape
        arr, E = _offsets_arg(ctx.offsets)
        dattn, dv = torch.empty_like(attn), torch.empty_like(v)
        call("dram_pcm_aggregate_bwd", _p(attn), _p(v), _p(dout), arr, E, _p(dattn), _p(dv), B, C, D, H, W, _stream())
        return dattn, dv, None


# merge types with a positional-encoding term (reference models.py:287-299): the same kernels on combined feature planes
PCM_GEO_MERGES = ("scaled_dot_product_geo", "scaled_dot_product_geo_relu", "att_is_all")


def pcm_attention(theta, phi, offsets, merge_type, geo_theta=None, geo_phi=None):
    """merge_func (reference models.py:259-331) on the voxel grid.  geo_theta / geo_phi: the projected positional
    encodings [B, geo_f_dim, D, H, W] that the geo merge types add to the appearance term."""
    offsets = tuple(map(tuple, offsets))
    if merge_type in PCM_GEO_MERGES:
        if geo_theta is None or geo_phi is None:
            raise ValueError(f"PCM merge_type {merge_type!r} needs the positional encodings (p_enc_dim > 0)")
        if merge_type == "att_is_all":                  # (theta + geo_theta) . (phi + geo_phi), models.py:297-299
            if theta.shape != geo_theta.shape:
                raise ValueError(f"att_is_all: f_dim {theta.shape[1]} and geo_f_dim {geo_theta.shape[1]} must agree")
            return PcmAttentionFn.apply(theta + geo_theta, phi + geo_phi, offsets, 0, 1)
        th, ph = torch.cat([theta, geo_theta], 1), torch.cat([phi, geo_phi], 1)
        if merge_type == "scaled_dot_product_geo":      # theta.phi + geo_theta.geo_phi: one dot over the stacked planes
            return PcmAttentionFn.apply(th, ph, offsets, 0, 1)
        return PcmAttentionFn.apply(th, ph, offsets, PCM_RELU, 1, theta.shape[1])     # relu(theta.phi) + geo_theta.geo_phi
    if merge_type in PCM_SUM_MERGES:
        return PcmAttentionSumFn.apply(theta, phi, offsets, PCM_SUM_MERGES[merge_type])
    if merge_type == "l2":
        # models.py:262-264: f = exp(-5 (theta - phi)^2), f / f.sum over the edges.  The reference subtracts [.., 1, f_dim] and
        # [.., f_dim, edges] tensors and later reshapes the aggregate to one row per node (models.py:396): both only work for
        # f_dim == 1.  There it is a softmax over the edges of -5 (theta - phi_e)^2 = -5 theta^2 + 10 theta phi_e - 5 phi_e^2,
        # whose first term is common to a node's edges and cancels: the 'sm' kernel on the feature pairs
        # (10 theta, -5) . (phi, phi^2).
        if theta.shape[1] != 1:
            raise ValueError(f"PCM merge_type 'l2' is only defined for f_dim == 1 (got {theta.shape[1]}): the reference "
                             f"broadcasts [.., 1, f_dim] against [.., f_dim, edges] (models.py:263)")
        th2 = torch.cat([10.0 * theta, torch.full_like(theta, -5.0)], 1)
        ph2 = torch.cat([phi, phi * phi], 1)
        return PcmAttentionFn.apply(th2, ph2, offsets, 0, 0)
    if merge_type not in PCM_MERGE_MODES:
        raise NotImplementedError(f"PCM merge_type {merge_type!r}: the dot-product family {sorted(PCM_MERGE_MODES)}, the "
                                  f"geo family {list(PCM_GEO_MERGES)}, {sorted(PCM_SUM_MERGES)} and 'l2' are implemented on the device")
    flags, scale_mode = PCM_MERGE_MODES[merge_type]
    return PcmAttentionFn.apply(theta, phi, offsets, flags, scale_mode)


def pcm_aggregate(attn, v, offsets):
    return PcmAggregateFn.apply(attn, v, tuple(map(tuple, offsets)))


# --------------------------------------------------------------------------- OneShot transforms
class ResizeTrilinearFn(Function):
    """F.interpolate(x, size / scale_factor, mode='trilinear') with align_corners=False
    (Rescale3DOneShot on "#image" tensors, reference data_transforms.py:1202-1239)."""

    @staticmethod
    def forward(ctx, x, size, scales):
        x = _chk(x, "resize input", 5)
        N, C, D, H, W = x.shape
        Do, Ho, Wo = (int(v) for v in size)
        y = torch.empty((N, C, Do, Ho, Wo), dtype=torch.float32, device=x.device)
        call("dram_resize_trilinear_fwd", _p(x), _p(y), N, C, D, H, W, Do, Ho, Wo, *scales, _stream())
        ctx.geom = (N, C, D, H, W, Do, Ho, Wo, scales)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        N, C, D, H, W, Do, Ho, Wo, scales = ctx.geom
        dy = _chk(dy, "resize grad_output", 5)
        dx = torch.empty((N, C, D, H, W), dtype=torch.float32, device=dy.device)
        call("dram_resize_trilinear_bwd", _p(dy), _p(dx), N, C, D, H, W, Do, Ho, Wo, *scales, _stream())
        return dx, None, None


def _resize_geometry(x, size, scale_factor):
    """Output size and the per-axis source scale ATen uses (0 = the default in/out)."""
    if (size is None) == (scale_factor is None):
        raise ValueError("only one of size or scale_factor should be defined")
    if size is not None:
        size = tuple(int(v) for v in (size if isinstance(size, (tuple, list, torch.Size)) else (size,) * 3))
        return size, (0.0, 0.0, 0.0)
    sf = tuple(float(v) for v in (scale_factor if isinstance(scale_factor, (tuple, list)) else (scale_factor,) * 3))
    size = tuple(int(math.floor(float(d) * s)) for d, s in zip(x.shape[2:], sf))
    return size, tuple(1.0 / s for s in sf)      # recompute_scale_factor=None: the given factor maps coordinates


def interpolate_trilinear(x, size=None, scale_factor=None):
    size, scales = _resize_geometry(x, size, scale_factor)
    return ResizeTrilinearFn.apply(x, size, scales)


def interpolate_nearest(x, size=None, scale_factor=None):
    """F.interpolate(mode='nearest'); label maps: no gradient."""
    size, scales = _resize_geometry(x, size, scale_factor)
    xc = _chk(x.detach(), "nearest input", 5)
    N, C, D, H, W = xc.shape
    y = torch.empty((N, C) + size, dtype=torch.float32, device=xc.device)
    call("dram_resize_nearest", _p(xc), _p(y), N, C, D, H, W, *size, *scales, _stream())
    return y


class AffineSampleFn(Function):
    """F.grid_sample(x, F.affine_grid(theta, x.size())) for one [3,4] theta shared by the batch
    (Rotate3DXOneShot, reference data_transforms.py:1186-1208)."""

    @staticmethod
    def forward(ctx, x, theta):
        x = _chk(x, "affine sample input", 5)
        N, C, D, H, W = x.shape
        arr = (ctypes.c_float * 12)(*[float(v) for v in theta])
        y = torch.empty_like(x)
        call("dram_affine_sample_fwd", _p(x), _p(y), arr, N, C, D, H, W, _stream())
        ctx.theta = tuple(float(v) for v in theta)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        dy = _chk(dy, "affine sample grad_output", 5)
        N, C, D, H, W = dy.shape
        arr = (ctypes.c_float * 12)(*ctx.theta)
        dx = torch.empty_like(dy)
        call("dram_affine_sample_bwd", _p(dy), _p(dx), arr, N, C, D, H, W, _stream())
        return dx, None


def affine_sample(x, theta12):
    return AffineSampleFn.apply(x, tuple(theta12))


class SpatialPermuteFlipFn(Function):
    """A signed permutation of the spatial axes of [N,C,D,H,W] (torch.flip / torch.rot90 / transpose chains)."""

    @staticmethod
    def forward(ctx, x, perm, flip):
        x = _chk(x, "permute/flip input", 5)
        N, C, D, H, W = x.shape
        ind = (D, H, W)
        y = torch.empty((N, C) + tuple(ind[p] for p in perm), dtype=torch.float32, device=x.device)
        call("dram_spatial_permute_flip", _p(x), _p(y), N, C, D, H, W, (ctypes.c_int * 3)(*perm), (ctypes.c_int * 3)(*flip),
             _stream())
        ctx.pf = (perm, flip)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        perm, flip = ctx.pf
        dy = _chk(dy, "permute/flip grad_output", 5)
        inv_perm, inv_flip = [0, 0, 0], [0, 0, 0]
        for k in range(3):                      # output axis k came from input axis perm[k] (flipped or not)
            inv_perm[perm[k]], inv_flip[perm[k]] = k, flip[k]
        N, C, D, H, W = dy.shape
        ind = (D, H, W)
        dx = torch.empty((N, C) + tuple(ind[p] for p in inv_perm), dtype=torch.float32, device=dy.device)
        call("dram_spatial_permute_flip", _p(dy), _p(dx), N, C, D, H, W, (ctypes.c_int * 3)(*inv_perm),
             (ctypes.c_int * 3)(*inv_flip), _stream())
        return dx, None, None


def signed_permutation(ops):
    """Compose torch-style ops on the spatial axes into (perm, flip): ("flip", dims) and ("transpose", a, b)
    with dims in 5-D numbering (2, 3, 4), applied in order."""
    perm, flip = [0, 1, 2], [0, 0, 0]
    for op in ops:
        if op[0] == "flip":
            for d in op[1]:
                flip[d - 2] ^= 1
        else:
            a, b = op[1] - 2, op[2] - 2
            perm[a], perm[b] = perm[b], perm[a]
            flip[a], flip[b] = flip[b], flip[a]
    return tuple(perm), tuple(flip)


def rot90_ops(k, dims):
    """torch.rot90(x, k, dims) as flips and a transpose."""
    a, b = dims
    k %= 4
    if k == 0:
        return []
    if k == 1:
        return [("flip", (b,)), ("transpose", a, b)]
    if k == 2:
        return [("flip", (a, b))]
    return [("flip", (a,)), ("transpose", a, b)]


def spatial_permute_flip(x, perm, flip):
    if tuple(perm) == (0, 1, 2) and not any(flip):
        return x
    return SpatialPermuteFlipFn.apply(x, tuple(perm), tuple(flip))
