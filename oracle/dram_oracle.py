"""CPU oracle for the DRAM DC3D hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain torch-CPU / numpy restatement of the algorithm in the
reference's ``dram/parts.py`` and ``dram/models.py`` (class ``DC3D``).  It is
imported only by ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- never by the product path
(``bodyct-dram_amd/``), which fails loudly when the HIP library is missing.

Parity pin: every function here is checked against golden vectors produced by
importing the reference itself on CPU (``oracle/make_golden.py`` ->
``tests/golden/*.npz``, checked by ``tests/test_oracle_golden.py``).

Everything is fp32, NCDHW.  Parameters travel in a flat dict keyed by the
reference's state-dict names (``ds_modules.0.conv_blocks.0.0.weight`` ...), so
that a reference checkpoint, the oracle and the HIP modules are interchangeable.

Two layers of restatement:
  * ``np_*``  - slow, loop/numpy definitions of each op from first principles
                (used on tiny shapes to pin the op semantics themselves);
  * the rest  - torch-CPU functional code with the reference's wiring, used at
                sizes the tests / the CPU baseline need.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F
from torch.utils.checkpoint import checkpoint

EPS = 1e-5          # torch default for BatchNorm3d / GroupNorm (reference parts.py:19-31 passes none)
BN_MOMENTUM = 0.1   # torch default

# The shipped benchmark model: reference dram/exp_settings/st_dram_ref.py:54-71.
ST_DRAM_REF_MODEL = {
    "n_layers": 3,
    "in_ch_list": [1, 64, 128, 256, 768, 384, 192],
    "base_ch_list": [32, 64, 128, 256, 256, 128, 64],
    "end_ch_list": [64, 128, 256, 512, 256, 128, 64],
    "kernel_sizes": [(3, 3)] * 7,
    "stacking": 3,
    "padding_list": [(1, 1)] * 7,
    "checkpoint_layers": [0, 1, 0, 1, 0, 1, 0],
    "dropout": 0.0,
    "upsample_ksize": (3, 3, 3),
    "upsample_sf": (2, 2, 2),
    "out_ch": 1,
}


# --------------------------------------------------------------------------
# first-principles numpy definitions (tiny shapes only)
# --------------------------------------------------------------------------
def np_conv3d(x, w, bias=None, pad=1):
    """y[n,o,z,y,x] = sum_{c,dz,dy,dx} w[o,c,dz,dy,dx] * xpad[n,c,z+dz,y+dy,x+dx]
    (cross-correlation, stride 1, zero padding) -- what nn.Conv3d in
    reference parts.py:177 computes."""
    N, C, D, H, W = x.shape
    O, _, k, _, _ = w.shape
    xp = np.zeros((N, C, D + 2 * pad, H + 2 * pad, W + 2 * pad), dtype=np.float64)
    xp[:, :, pad:pad + D, pad:pad + H, pad:pad + W] = x
    Do, Ho, Wo = D + 2 * pad - k + 1, H + 2 * pad - k + 1, W + 2 * pad - k + 1
    y = np.zeros((N, O, Do, Ho, Wo), dtype=np.float64)
    for dz in range(k):
        for dy in range(k):
            for dx in range(k):
                patch = xp[:, :, dz:dz + Do, dy:dy + Ho, dx:dx + Wo]
                y += np.einsum("nczyx,oc->nozyx", patch, w[:, :, dz, dy, dx].astype(np.float64))
    if bias is not None:
        y += bias.reshape(1, -1, 1, 1, 1)
    return y.astype(np.float32)


def np_maxpool2(x):
    """2x2x2 / stride 2 / no padding max pool with floor sizes (parts.py:191).
    Returns (out, idx) with idx in 0..7 = first maximum in (z,y,x) scan order
    (the element ATen's CPU kernel routes the gradient to)."""
    N, C, D, H, W = x.shape
    Do, Ho, Wo = D // 2, H // 2, W // 2
    out = np.empty((N, C, Do, Ho, Wo), dtype=x.dtype)
    idx = np.empty((N, C, Do, Ho, Wo), dtype=np.uint8)
    for z in range(Do):
        for y in range(Ho):
            for xx in range(Wo):
                win = x[:, :, 2 * z:2 * z + 2, 2 * y:2 * y + 2, 2 * xx:2 * xx + 2].reshape(N, C, 8)
                out[:, :, z, y, xx] = win.max(-1)
                idx[:, :, z, y, xx] = win.argmax(-1)  # numpy argmax = first occurrence
    return out, idx


def _ac_axis(n_in, n_out):
    """align_corners=True source coordinates for one axis, computed like ATen
    (area_pixel_compute_scale / compute_source_index): scale=(in-1)/(out-1) in
    fp32, src = scale*dst, i0=int(src), lam1=src-i0, i1=i0+(i0<in-1)."""
    scale = np.float32(0.0) if n_out <= 1 else np.float32(n_in - 1) / np.float32(n_out - 1)
    dst = np.arange(n_out, dtype=np.float32)
    src = scale * dst
    i0 = src.astype(np.int64)
    i0 = np.minimum(i0, n_in - 1)
    i1 = i0 + (i0 < n_in - 1)
    l1 = (src - i0.astype(np.float32)).astype(np.float32)
    l0 = (np.float32(1.0) - l1).astype(np.float32)
    return i0, i1, l0, l1


def np_trilinear_ac(x, size):
    """nn.Upsample(mode='trilinear', align_corners=True) to `size`
    (parts.py:149, models.py:146)."""
    N, C, D, H, W = x.shape
    Do, Ho, Wo = size
    z0, z1, a0, a1 = _ac_axis(D, Do)
    y0, y1, b0, b1 = _ac_axis(H, Ho)
    x0, x1, c0, c1 = _ac_axis(W, Wo)
    xf = x.astype(np.float32)
    out = np.zeros((N, C, Do, Ho, Wo), dtype=np.float32)
    for (zi, za) in ((z0, a0), (z1, a1)):
        for (yi, yb) in ((y0, b0), (y1, b1)):
            for (xi, xc) in ((x0, c0), (x1, c1)):
                wgt = (za[:, None, None] * yb[None, :, None] * xc[None, None, :]).astype(np.float32)
                out += wgt * xf[:, :, zi[:, None, None], yi[None, :, None], xi[None, None, :]]
    return out


def np_batchnorm_train(x, gamma, beta, eps=EPS):
    """Training-mode BatchNorm3d: per-channel mean / biased variance over
    (N,D,H,W) (parts.py:19).  Returns y, mean, biased var."""
    xd = x.astype(np.float64)
    mean = xd.mean(axis=(0, 2, 3, 4))
    var = xd.var(axis=(0, 2, 3, 4))
    sh = (1, -1, 1, 1, 1)
    y = (xd - mean.reshape(sh)) / np.sqrt(var.reshape(sh) + eps)
    if gamma is not None:
        y = y * gamma.reshape(sh) + beta.reshape(sh)
    return y.astype(np.float32), mean.astype(np.float32), var.astype(np.float32)


def np_groupnorm(x, groups, gamma, beta, eps=EPS):
    """GroupNorm(groups, C): per-(sample, group) mean / biased variance over
    (C/G, D, H, W) (parts.py:26-31; 'ln' = 1 group, 'in' = C groups)."""
    N, C = x.shape[:2]
    xd = x.astype(np.float64).reshape(N, groups, -1)
    mean = xd.mean(-1, keepdims=True)
    var = xd.var(-1, keepdims=True)
    y = ((xd - mean) / np.sqrt(var + eps)).reshape(x.shape)
    if gamma is not None:
        sh = (1, -1, 1, 1, 1)
        y = y * gamma.reshape(sh) + beta.reshape(sh)
    return y.astype(np.float32)


# --------------------------------------------------------------------------
# op-level torch-CPU restatement
# --------------------------------------------------------------------------
def conv3d(x, w, bias=None, pad=1):
    """nn.Conv3d(stride=1) of parts.py:177 / models.py:109."""
    return F.conv3d(x, w, bias, stride=1, padding=pad)


def crop_offsets(small, big):
    """Start offsets of the centre crop in crop_concat_5d: ceil((b-a)/2) per
    spatial axis (parts.py:42-44)."""
    return tuple(int(math.ceil((b - a) / 2)) for a, b in zip(small, big))


def crop_concat_5d(t1, t2):
    """cat([t1, centre_crop(t2 -> t1 spatial)], dim=1) (parts.py:37-46):
    t1 (the upsampled tensor) comes FIRST."""
    assert t1.dim() == t2.dim() == 5
    assert t1.shape[-1] <= t2.shape[-1]
    oz, oy, ox = crop_offsets(t1.shape[2:], t2.shape[2:])
    d, h, w = t1.shape[2:]
    return torch.cat([t1, t2[:, :, oz:oz + d, oy:oy + h, ox:ox + w]], dim=1)


def upsample_trilinear_ac(x, scale_factor=None, size=None):
    """nn.Upsample(mode='trilinear', align_corners=True) (parts.py:149,
    models.py:146)."""
    return F.interpolate(x, size=size, scale_factor=scale_factor, mode="trilinear", align_corners=True)


def max_pool3d_2(x):
    """nn.MaxPool3d(2, 2, 0) (parts.py:191)."""
    return F.max_pool3d(x, kernel_size=2, stride=2, padding=0)


def pooling_dense_features(dense_outs, lungs, pooling_method="avg"):
    """models.py:37-49."""
    B, C = dense_outs.shape[0], dense_outs.shape[1]
    if pooling_method == "global_avg":
        return F.adaptive_avg_pool3d(dense_outs, 1).view(B, C)
    if pooling_method == "global_max":
        return F.adaptive_max_pool3d(dense_outs, 1).view(B, C)
    le = lungs.expand_as(dense_outs)
    return (dense_outs * le).view(B, C, -1).sum(dim=-1) / le.view(B, C, -1).sum(dim=-1)


def _norm(method, prefix, params, buffers, x, training):
    """normal_wrapper (parts.py:17-35) applied functionally.

    `buffers` (running_mean / running_var / num_batches_tracked) is updated in
    place in training mode exactly like nn.BatchNorm3d (momentum 0.1, unbiased
    variance into running_var)."""
    if method in ("bn", "sbn"):
        rm, rv = buffers[prefix + ".running_mean"], buffers[prefix + ".running_var"]
        if training:
            buffers[prefix + ".num_batches_tracked"] += 1
        return F.batch_norm(x, rm, rv, params[prefix + ".weight"], params[prefix + ".bias"],
                            training, BN_MOMENTUM, EPS)
    if method == "bnt":
        return F.batch_norm(x, None, None, params[prefix + ".weight"], params[prefix + ".bias"],
                            True, BN_MOMENTUM, EPS)
    if method == "bntna":
        return F.batch_norm(x, None, None, None, None, True, BN_MOMENTUM, EPS)
    if method == "ln":
        return F.group_norm(x, 1, params[prefix + ".weight"], params[prefix + ".bias"], EPS)
    if method == "lnna":
        return F.group_norm(x, 1, None, None, EPS)
    if method == "in":
        return F.group_norm(x, x.shape[1], params[prefix + ".weight"], params[prefix + ".bias"], EPS)
    return x  # Identity (parts.py:35)


def conv_norm_act_stack(prefix, n_convs, params, buffers, x, norm_method, training, pads):
    """The `conv_blocks` Sequential of all three block types
    (parts.py:102-110, 138-146, 183-190): [Conv3d -> norm -> ReLU] * n."""
    for j in range(n_convs):
        p = f"{prefix}.conv_blocks.{j}"
        x = conv3d(x, params[p + ".0.weight"], params.get(p + ".0.bias"), pads[j])
        x = _norm(norm_method, p + ".1", params, buffers, x, training)
        x = F.relu(x)
    return x


def _pads(cfg, n):
    p = cfg["padding_list"][n]
    return list(p) if isinstance(p, (tuple, list)) else [p, p]


def dc3d_forward(cfg, params, buffers, x, training=False, norm_method="bn", use_checkpoint=True):
    """DC3D.forward (models.py:120-147) for `cfg` (a MODEL dict without
    'method').  Returns dense_outs [N, out_ch, D, H, W] (the reference returns
    this same tensor twice).

    With `use_checkpoint` the blocks flagged in cfg['checkpoint_layers'] go
    through torch.utils.checkpoint like the reference does, which re-runs their
    forward during backward and therefore updates their BatchNorm running
    statistics twice per training step."""
    L = cfg["n_layers"]
    ck = cfg["checkpoint_layers"]
    grad_mode = torch.is_grad_enabled()

    def run(fn, flag, *args):
        if use_checkpoint and flag > 0 and grad_mode:
            return checkpoint(fn, *args, use_reentrant=True)
        return fn(*args)

    feats = []
    cur = x
    for n in range(L):
        def ds(inp, n=n):
            y = conv_norm_act_stack(f"ds_modules.{n}", 2, params, buffers, inp, norm_method, training, _pads(cfg, n))
            return y, max_pool3d_2(y)
        if ck[n] > 0 and n == 0:
            # models.py:125 passes a dummy requires-grad tensor so the reentrant
            # checkpoint of the first block still records a graph.
            dummy = torch.ones(1, requires_grad=True)
            y, cur = run(lambda inp, _d: ds(inp), ck[n], cur, dummy)
        else:
            y, cur = run(ds, ck[n], cur)
        feats.append(y)

    def bg(inp):
        return conv_norm_act_stack("bg", 2, params, buffers, inp, norm_method, training, _pads(cfg, L))
    cur = run(bg, ck[L], cur)

    if (L + 1) < len(cfg["in_ch_list"]):
        for idx, skip in enumerate(reversed(feats)):
            if cfg.get("stacking", 0) == idx:
                break

            def us(inp, cat, idx=idx):
                up = upsample_trilinear_ac(inp, scale_factor=tuple(cfg["upsample_sf"])
                                           if isinstance(cfg["upsample_sf"], (tuple, list)) else cfg["upsample_sf"])
                z = crop_concat_5d(up, cat)
                return conv_norm_act_stack(f"us_modules.{idx}", 2, params, buffers, z, norm_method, training,
                                           _pads(cfg, L + 1 + idx))
            # NB models.py:140 indexes the flag with n_layers + idx (not +1).
            cur = run(us, ck[L + idx], cur, skip)

    dense = conv3d(cur, params["top_layer.weight"], params["top_layer.bias"], 0)
    dense = upsample_trilinear_ac(dense, size=tuple(x.shape[-3:]))
    return dense


def init_params(cfg, norm_method="bn", seed=0):
    """Random-init parameters with the reference's HeNorm(mode='fan_in')
    statistics (models.py:17-35): conv weight ~ kaiming_normal(fan_in), conv
    bias 0.01, norm weight 1 / bias 0.  (Not the same random stream as
    constructing the reference modules -- use a golden state dict for that.)"""
    g = torch.Generator().manual_seed(seed)
    L = cfg["n_layers"]
    params, buffers = {}, {}
    conv_bias = norm_method is None

    def add_stack(prefix, cins, couts):
        for j, (ci, co) in enumerate(zip(cins, couts)):
            std = math.sqrt(2.0 / (ci * 27))
            params[f"{prefix}.conv_blocks.{j}.0.weight"] = torch.randn(co, ci, 3, 3, 3, generator=g) * std
            if conv_bias:
                params[f"{prefix}.conv_blocks.{j}.0.bias"] = torch.full((co,), 0.01)
            if norm_method in ("bn", "sbn", "bnt", "ln", "in"):
                params[f"{prefix}.conv_blocks.{j}.1.weight"] = torch.ones(co)
                params[f"{prefix}.conv_blocks.{j}.1.bias"] = torch.zeros(co)
            if norm_method in ("bn", "sbn"):
                buffers[f"{prefix}.conv_blocks.{j}.1.running_mean"] = torch.zeros(co)
                buffers[f"{prefix}.conv_blocks.{j}.1.running_var"] = torch.ones(co)
                buffers[f"{prefix}.conv_blocks.{j}.1.num_batches_tracked"] = torch.tensor(0, dtype=torch.long)

    names = [f"ds_modules.{n}" for n in range(L)] + ["bg"] + [f"us_modules.{n}" for n in range(L)]
    for n, name in enumerate(names):
        if n >= len(cfg["in_ch_list"]):
            break
        add_stack(name, [cfg["in_ch_list"][n], cfg["base_ch_list"][n]],
                  [cfg["base_ch_list"][n], cfg["end_ch_list"][n]])
    cin_top = cfg["end_ch_list"][L + cfg.get("stacking", 0)]
    params["top_layer.weight"] = torch.randn(cfg["out_ch"], cin_top, 1, 1, 1, generator=g) * math.sqrt(2.0 / cin_top)
    params["top_layer.bias"] = torch.full((cfg["out_ch"],), 0.01)
    return params, buffers


def split_state_dict(sd):
    """Reference state dict -> (params, buffers) float32/long CPU tensors."""
    params, buffers = {}, {}
    for k, v in sd.items():
        t = torch.as_tensor(v).clone()
        if k.endswith(("running_mean", "running_var", "num_batches_tracked")):
            buffers[k] = t
        else:
            params[k] = t
    return params, buffers


# --------------------------------------------------------------------------
# loss restatement (reference dram/metrics.py)
# --------------------------------------------------------------------------
CTSS_RATIO_MAP = {0: (0.0, 0.001), 1: (0.001, 0.01), 2: (0.01, 0.05),
                  3: (0.05, 0.35), 4: (0.35, 0.5), 5: (0.5, 1.00001)}   # metrics.py:76-83


def get_labels(ctsses, lesion_ps, band_width):
    """IntRegLoss.get_labels (metrics.py:122-138)."""
    labels = []
    for ctss, lesion_p in zip(ctsses, lesion_ps):
        lp = float(lesion_p)
        lb, ub = max(0.0, lp - band_width), min(1.0, lp + band_width)
        c_lb, c_ub = CTSS_RATIO_MAP[int(float(ctss))]
        band = (max(c_lb, lb), min(c_ub, ub))
        if band[1] < band[0]:
            if ub <= c_lb:
                band = (lb, ub)
            elif lb >= c_ub:
                band = (c_lb, c_ub)
            else:
                raise RuntimeError("cannot reach here!")
        labels.append(band)
    return torch.tensor(labels, dtype=torch.float32)


def reg_loss_with_probs(probs, lobes, lesions, ctsses, freq_map, band_width):
    """IntRegLoss.compute_reg_loss_with_probs (metrics.py:158-177)."""
    B = probs.shape[0]
    ratio_ub = (lesions * lobes).view(B, 1, -1).sum(-1) / lobes.view(B, 1, -1).sum(-1)
    m = (lobes > 0).to(probs.dtype)
    pred_ratio = (probs * m).view(B, -1).sum(-1) / m.view(B, -1).sum(-1)   # == mean of probs[lobes>0] per sample
    tgt = get_labels(ctsses, ratio_ub.view(-1), band_width).to(probs.device)
    K = (0.5 * (tgt[:, 1] - tgt[:, 0])) ** 2
    unh = (pred_ratio - (tgt[:, 1] + tgt[:, 0]) / 2.0) ** 2 - K
    unw = torch.clamp(unh, min=0.0)
    wf = torch.tensor([freq_map[int(float(c))] for c in ctsses], dtype=torch.float32, device=probs.device)
    wf = torch.clamp(wf, 0.2, 0.8)
    return (unw / wf).sum()


def boot_bce(p, t, voi, smoothing=0.1, eps=1e-7):
    """BootBinCrossEntropy.__call__ (metrics.py:17-51)."""
    t = t.to(p.dtype)
    tb = voi < 1e-7 if voi.dtype != torch.bool else ~voi
    po, to = p[tb], t[tb]
    pto = (po * to + (1.0 - po) * (1.0 - to)).clamp(eps, 1.0 - eps)
    bceo = (-torch.log(pto)).mean()
    tf = voi > 0
    if tf.sum() > 0:
        pi, ti = p[tf], t[tf]
        alpha = (1.0 - ti.sum() / tf.sum()).clamp(0.25, 0.75)
        pti = (pi * ti + (1.0 - pi) * (1.0 - ti)).clamp(eps, 1.0 - eps)
        w = alpha * ti + (1.0 - alpha) * (1.0 - ti)
        bce = (-torch.log(pti) * w).sum() / w.sum()
        th = (pi > 0.5).to(p.dtype)
        pth = (pi * th + (1.0 - pi) * (1.0 - th)).clamp(eps, 1.0 - eps)
        boot = (-torch.log(pth)).mean()
        return bceo + (1.0 - smoothing) * bce + smoothing * boot
    return bceo


def int_reg_refine_loss(dense, lobes, lesions, ctsses, freq_map, band_width=1e-2, smoothing=0.1):
    """IntRegRefineLoss.__call__ (metrics.py:360-373) given the model output
    (`dense` is both dense_outs and refined_dense_outs for DC3D)."""
    probs = torch.sigmoid(dense)
    reg = reg_loss_with_probs(probs, lobes, lesions, ctsses, freq_map, band_width)
    # compute_seg_loss (metrics.py:331-358): pseudo label = (p>0.5 inside lobe) & lesion, zero if ctss==0
    with torch.no_grad():
        pd = probs.detach().clone()
        pd[lobes == 0] = 0.0
        pseudo = ((pd > 0.5) & (lesions > 0)).to(dense.dtype)
        keep = torch.tensor([0.0 if float(c) < 1e-7 else 1.0 for c in ctsses], dtype=dense.dtype,
                            device=dense.device).view(-1, 1, 1, 1, 1)
        pseudo = pseudo * keep
    seg = boot_bce(probs, pseudo, lobes > 0, smoothing)
    return reg, seg


# --------------------------------------------------------------------------
# whole-scan inference restatement (reference dram/job_runner.py:720-779, 1003-1005)
# --------------------------------------------------------------------------
def windowing(image, from_span, to_span=(0.0, 1.0)):
    """dram/utils.py:189-198."""
    image = np.clip(image, a_min=from_span[0], a_max=from_span[1])
    return ((image - from_span[0]) / float(from_span[1] - from_span[0])) * (to_span[1] - to_span[0]) + to_span[0]


def find_crops(mask, spacing, border):
    """dram/utils.py:244-254 (scipy.ndimage.find_objects of the whole mask = its bounding box)."""
    idx = np.nonzero(mask)
    sl = tuple(slice(int(i.min()), int(i.max()) + 1) for i in idx)
    if border > 0:
        sl = tuple(slice(max(0, o.start - int(math.ceil(border / sp))), min(ss, o.stop + int(math.ceil(border / sp))))
                   for o, ss, sp in zip(sl, mask.shape, spacing))
    return sl


def threshold_otsu_u8(img_u8):
    """skimage.filters.threshold_otsu for a uint8 array, restated from the published algorithm:
    histogram over the occupied integer range, maximise the between-class variance.  (skimage is
    not installed here: parity with the library itself is unpinned.)"""
    v = np.asarray(img_u8).ravel()
    lo, hi = int(v.min()), int(v.max())
    hist = np.bincount(v, minlength=256)[lo:hi + 1].astype(np.float64)
    centers = np.arange(lo, hi + 1, dtype=np.float64)
    w1 = np.cumsum(hist)
    w2 = np.cumsum(hist[::-1])[::-1]
    m1 = np.cumsum(hist * centers) / w1
    m2 = (np.cumsum((hist * centers)[::-1]) / w2[::-1])[::-1]
    var12 = w1[:-1] * w2[1:] * (m1[:-1] - m2[1:]) ** 2
    return float(centers[int(np.argmax(var12))])


def binary_cam(cam_np, scaler=1.0, from_span=(0, 1)):
    """dram/utils.py:226-242: returns (mask, th/255)."""
    w = windowing(cam_np, from_span=from_span, to_span=(0, 255)).astype(np.uint8)
    u = np.unique(w)
    if len(u) < 2:
        return np.ones_like(w).astype(bool), u[0] / 255.0
    th = min(threshold_otsu_u8(w) * scaler, 255.0)
    return w >= th, th / 255.0


def resample_itk_linear(img, size):
    """Resample('fixed_size') of a 3-D float image (data_transforms.py:170-175 -> utils.resample -> utils.py:371-381:
    sitk.ResampleImageFilter.Execute(image, new_size, identity transform, sitkLinear, the image's origin and direction,
    new_spacing = spacing * size_in / size_out, default value 0)).  SimpleITK 1.1.0 (requirements.in:33) is not installed
    (PARITY UNPINNED); restated from ITK's published semantics of that call: output voxel o sits at continuous input index
    c = o * size_in / size_out (same origin, identity transform); inside the buffer while c < size_in - 0.5
    (ImageFunction::IsInsideBuffer), else the default value; linear interpolation between floor(c) and floor(c) + 1 with the
    upper neighbour clamped to the last voxel (LinearInterpolateImageFunction).  fp64 here, fp32 on the device."""
    out = np.asarray(img, dtype=np.float64)
    inside = []
    for ax, n_out in enumerate(size):
        n_in = out.shape[ax]
        c = np.arange(n_out, dtype=np.float64) * float(n_in) / float(n_out)
        inside.append(c < n_in - 0.5)
        b = np.minimum(c.astype(np.int64), n_in - 1)
        u = np.minimum(b + 1, n_in - 1)
        w1 = np.where(u == b, 0.0, c - b)
        shape = [1, 1, 1]
        shape[ax] = n_out
        out = np.take(out, b, axis=ax) * (1.0 - w1).reshape(shape) + np.take(out, u, axis=ax) * w1.reshape(shape)
    ok = inside[0][:, None, None] & inside[1][None, :, None] & inside[2][None, None, :]
    return np.where(ok, out, 0.0).astype(np.float32)


def resample_itk(narray, spacing, required_spacing, new_size, interpolator="linear"):
    """utils.resample(narray, spacing, factor=2, required_spacing=..., new_size=..., interpolator=...) (utils.py:414-434 ->
    resample_sitk_image, utils.py:299-381) as LesionSegTest.run calls it for the way back to the scan's original grid
    (job_runner.py:1016-1032): sitk.ResampleImageFilter.Execute(image, new_size, identity transform, nearest / linear, the
    image's origin and direction, required_spacing, default value 0, the image's own pixel type).  SimpleITK 1.1.0 is not
    installed (PARITY UNPINNED); restated from ITK's published semantics: output voxel o sits at continuous input index
    c = o * required_spacing / spacing per axis, inside the buffer while c < size_in - 0.5; nearest neighbour = floor(c + 0.5)
    (Math::RoundHalfIntegerUp); linear = lerps along x, then y, then z in double with the upper neighbour clamped to the last
    voxel; integer pixel types by clamp + truncation (ResampleImageFilter::CastPixelWithBoundsChecking).  Axes in (z, y, x)
    order on both sides, as the reference's numpy side has them."""
    a = np.asarray(narray)
    idx, inside, frac = [], [], []
    for ax in range(3):
        n_in = a.shape[ax]
        c = np.arange(int(new_size[ax]), dtype=np.float64) * (float(required_spacing[ax]) / float(spacing[ax]))
        inside.append(c < n_in - 0.5)
        if interpolator == "nearest":
            idx.append((np.minimum((c + 0.5).astype(np.int64), n_in - 1),))
        else:
            b = np.minimum(c.astype(np.int64), n_in - 1)
            u = np.minimum(b + 1, n_in - 1)
            idx.append((b, u))
            frac.append(np.where(u == b, 0.0, c - b))
    ok = inside[0][:, None, None] & inside[1][None, :, None] & inside[2][None, None, :]
    if interpolator == "nearest":
        out = a[np.ix_(idx[0][0], idx[1][0], idx[2][0])]
        return np.where(ok, out, 0).astype(a.dtype)
    assert interpolator == "linear"
    f = a.astype(np.float64)
    tz, ty, tx = frac[0][:, None, None], frac[1][None, :, None], frac[2][None, None, :]
    g = lambda i, j, k: f[np.ix_(idx[0][i], idx[1][j], idx[2][k])]
    lerp = lambda p, q, t: p + (q - p) * t
    v = lerp(lerp(lerp(g(0, 0, 0), g(0, 0, 1), tx), lerp(g(0, 1, 0), g(0, 1, 1), tx), ty),
             lerp(lerp(g(1, 0, 0), g(1, 0, 1), tx), lerp(g(1, 1, 0), g(1, 1, 1), tx), ty), tz)
    v = np.where(ok, v, 0.0)
    if np.issubdtype(a.dtype, np.integer):
        info = np.iinfo(a.dtype)
        return np.trunc(np.clip(v, info.min, info.max)).astype(a.dtype)
    return v.astype(a.dtype)


def evaluate_scan(cfg, params, buffers, scan, lobe, spacing, norm_method="bn", resample=80,
                  window=(-1000.0, -300.0), border=5.0, forward=None):
    """evaluate_scan (job_runner.py:729-770) + the thresholding of LesionSegTest.run
    (job_runner.py:1003-1005), one lobe at a time like the reference.  The crop -> resample^3 step
    restates the grid of the reference's SimpleITK call from ITK's published semantics (resample_itk_linear: unpinned).
    `forward(t) -> logits` replaces the DC3D forward (e.g. the refined output of dc3dat_forward: the
    reference takes the model's *second* output, job_runner.py:764)."""
    htp = np.zeros(scan.shape, dtype=np.float32)
    with torch.no_grad():
        for label in np.unique(lobe)[1:]:
            lobe_binary = lobe == label
            sl = find_crops(lobe_binary, spacing, border)
            lobe_chunk = lobe_binary[sl]
            scan_chunk = scan[sl].astype(np.float32).copy()
            crop_size = lobe_chunk.shape
            scan_chunk[lobe_chunk == 0] = -2048
            img = windowing(scan_chunk, from_span=window, to_span=(0.0, 1.0)).astype(np.float32)
            t = torch.from_numpy(resample_itk_linear(img, (resample,) * 3))[None, None]
            dense = forward(t) if forward is not None else \
                dc3d_forward(cfg, params, buffers, t, training=False, norm_method=norm_method)
            probs = torch.sigmoid(dense)
            probs = upsample_trilinear_ac(probs, size=tuple(crop_size))[0, 0].numpy()
            view = htp[sl]
            view[lobe_chunk] = probs[lobe_chunk]
    lung = lobe > 0
    _, th = binary_cam(htp[lung])
    mask = htp > th
    ratio = float((htp * lung).sum() / lung.sum())
    return htp, mask, th, ratio


def lesion_post_process(htp, scan, lobe, vessel, th, post_window=(-1150, 350), scaler=0.75):
    """The tail of LesionSegTest.run (dram/job_runner.py:1004-1012) on a finished heat map: lesion_pred = htp > th;
    w_scan = windowing(scan, to_span=(0, 1)) with windowing()'s DEFAULT from_span (-1150, 350) (utils.py:189);
    _, th2 = binary_cam(w_scan[lobe > 0], 0.75); lesion_pred_post = lesion_pred & (w_scan > th2) & ~(vessel > 0).
    Returns (lesion_pred uint8, lesion_pred_post uint8, th2)."""
    lesion_pred = htp > th
    w_scan = windowing(scan, from_span=post_window, to_span=(0, 1))
    _, th2 = binary_cam(w_scan[lobe > 0], scaler)
    vessel = np.zeros(scan.shape, dtype=np.uint8) if vessel is None else vessel
    post = np.logical_and(np.logical_and(lesion_pred, w_scan > th2), np.logical_not(vessel > 0)).astype(np.uint8)
    return lesion_pred.astype(np.uint8), post, th2


def iou(predict, target, smooth):
    """dram/utils.py:437-442."""
    intersection = np.sum(np.logical_and(predict, target))
    union = np.sum(np.logical_or(predict, target))
    return (intersection + smooth) / (union + smooth)


def dice(predict, target, smooth):
    """dram/utils.py:444-446."""
    intersection = np.sum(np.logical_and(predict, target))
    return (2.0 * intersection.sum() + smooth) / (predict.sum() + target.sum() + smooth)


# --------------------------------------------------------------------------
# PCM local attention + DC3DATGeneric (reference dram/models.py:150-597)
#
# PARITY UNPINNED at the PCM boundary: PCM.forward needs the real DGL (dgl.DGLGraph / update_all,
# models.py:256-258,340), a third-party package that is absent from /root/reference and from this
# image ("install 0.6.x", README.md:12; the Dockerfile builds git master).  The reference holds no
# test or fixture for it.  What follows restates models.py:221-411 from reading it; `pcm_forward`
# (dense, shifted views) and `pcm_forward_literal` (node by node, mailbox by mailbox, written to
# mirror reduce_func / compute_cross_x line by line) are two independent restatements that are checked
# against each other.  Everything of DC3DATGeneric around the PCM call *is* pinned by a golden vector
# produced from the reference with `attention_module` swapped for a pass-through
# (oracle/make_golden.py:gen_att).
# --------------------------------------------------------------------------
ST_DRAM_REF_ATT_MODEL = dict(ST_DRAM_REF_MODEL, **{     # dram/exp_settings/st_dram_ref_att.py:56-82
    "at_spatial_size": (64, 64, 64), "at_f_dim": 8, "at_g_dim": 8, "at_g_iter": 1, "at_k_size": 3,
    "at_merge_type": "scaled_dot_product_relu", "at_self_loop": False, "at_layers": [-1, 0, 1],
    "at_p_enc_dim": 0, "at_geo_f_dim": 0,
})

PCM_DOT_MERGES = ("sm", "scaled_dot_product", "scaled_dot_product_relu", "smrelu", "smscaled", "l2sm", "l2smrelu")
PCM_GEO_MERGES = ("scaled_dot_product_geo", "scaled_dot_product_geo_relu", "att_is_all")     # models.py:287-299
PCM_SUM_MERGES = ("cosine", "heu1", "heu2")                                                   # models.py:300-302, 307-320
PCM_L2_MERGE = "l2"      # models.py:262-264: exp(-5 (theta - phi)^2) / sum over the edges; its broadcast and the reshape of
                         # compute_cross_x (models.py:396) are only defined for f_dim == 1


def pcm_geo_feature(p_enc_dim, spatial, dtype=torch.float32):
    """PCM.build_geo_feature (models.py:194-219) for one sample: [p_enc_dim, D, H, W]."""
    spatial = tuple(spatial)
    if p_enc_dim % (2 * len(spatial)) != 0:
        raise ValueError("Cannot use sin/cos positional encoding with odd dimension (got dim={:d})".format(p_enc_dim))
    pos = torch.ones(spatial).nonzero().to(dtype).view(*spatial, len(spatial))
    pe = torch.zeros(p_enc_dim, *spatial, dtype=dtype)
    d_model = int(p_enc_dim / len(spatial))
    div = torch.pow(1e-4, torch.arange(0., d_model, 2) / d_model).to(dtype)
    for d in range(len(spatial)):
        start, end = d * d_model, (d + 1) * d_model
        arg = pos[..., d].expand(len(div), *spatial) * div.view(len(div), *([1] * len(spatial)))
        pe[start:end:2] = torch.sin(arg)
        pe[(start + 1):end:2] = torch.cos(arg)
    return pe


def pcm_offsets(k_size=3, connectivity=2, self_loop=True):
    """Neighbour offsets of PCM.init_graph (models.py:230-232): the `connectivity` structuring element,
    nearest-neighbour zoomed to k_size, relative to its centre; the centre is dropped when
    self_loop is False (dgl.transform.remove_self_loop, models.py:257-258)."""
    from scipy import ndimage
    base = ndimage.generate_binary_structure(3, connectivity)
    base = ndimage.zoom(base, k_size / 3.0, order=0)
    off = np.asarray(np.where(base > 0)).T - np.asarray([k_size // 2] * 3)
    if not self_loop:
        off = off[np.any(off != 0, axis=1)]
    return off.astype(np.int64)


def _pcm_logits(merge_type, f, valid, deg):
    """merge_func (models.py:259-331), dot-product family, on dense [B, E, ...] logits `f`; `valid`
    marks real edges, `deg` = number of them (= f.shape[-1] of the reference's degree bucket)."""
    ninf = torch.finfo(f.dtype).min
    if merge_type in ("scaled_dot_product_relu", "smrelu", "l2smrelu"):
        f = F.relu(f)
    if merge_type in ("l2sm", "l2smrelu"):          # F.normalize(f, dim=-1): over the node's edges
        nrm = torch.sqrt(((f * valid) ** 2).sum(1, keepdim=True)).clamp_min(1e-12)
        f = f / nrm
    if merge_type in ("scaled_dot_product", "scaled_dot_product_relu"):
        f = f / torch.sqrt(deg)
    if merge_type == "smscaled":
        f = f / 0.01
    f = torch.where(valid.bool().expand_as(f), f, torch.full_like(f, ninf))
    return torch.softmax(f, dim=1)


def _lin(p, name, x):
    """nn.Linear over the channel axis of [B, C, ...]; Identity when the module has no weight."""
    w = p.get(name + ".weight")
    if w is None:
        return x
    y = torch.einsum("bc...,fc->bf...", x, w)
    return y + p[name + ".bias"].view(1, -1, *([1] * (x.dim() - 2)))


def pcm_forward(p, cam, f, k_size=3, connectivity=2, self_loop=True, merge_type="scaled_dot_product_relu",
                non_local_iter=1, residual=False, p_enc_dim=0):
    """PCM.forward (models.py:333-363) as dense tensor algebra.
    p: {"theta.weight", "theta.bias", "phi.*", "G.*", "r.*", "geo_theta.*", "geo_phi.*"} (absent = Identity,
    models.py:169-192).  cam [B, g_ch, D, H, W], f [B, in_ch, D, H, W] -> refined cam [B, g_ch, D, H, W]."""
    if merge_type not in PCM_DOT_MERGES + PCM_GEO_MERGES + PCM_SUM_MERGES + (PCM_L2_MERGE,):
        raise NotImplementedError(merge_type)
    B, _, D, H, W = f.shape
    offs = pcm_offsets(k_size, connectivity, self_loop)
    R = int(np.abs(offs).max()) if len(offs) else 0
    th, ph = _lin(p, "theta", f), _lin(p, "phi", f)
    ones = torch.ones((1, 1, D, H, W), dtype=f.dtype)

    def shifted(t, o):      # t at node + o, zero outside the grid
        tp = F.pad(t, (R, R, R, R, R, R))
        return tp[..., R + o[0]:R + o[0] + D, R + o[1]:R + o[1] + H, R + o[2]:R + o[2] + W]
    valid = torch.cat([shifted(ones, o) for o in offs], dim=1)              # [1, E, D, H, W]
    deg = valid.sum(1, keepdim=True)
    if merge_type in PCM_SUM_MERGES:                                        # no softmax: f / (eps + f.sum over the edges)
        if merge_type == "cosine":
            v = torch.stack([F.cosine_similarity(th, shifted(ph, o), dim=1) for o in offs], dim=1)
            eps = 0.0
        else:
            dot = torch.stack([(th * shifted(ph, o)).sum(1) for o in offs], dim=1)
            l1 = torch.stack([(th - shifted(ph, o)).abs().sum(1) for o in offs], dim=1)
            v = dot / (1.0 + l1)
            if merge_type == "heu1":
                with torch.no_grad():       # models.py:311-314: the masked f is formed INSIDE no_grad -- it (and with it
                    mask = torch.ones_like(v)           # the whole heu1 attention) carries no gradient to theta / phi
                    mask[v < 0.03] = 0.0
                    v = v * mask
            else:
                v = F.relu(v)
            eps = 1e-7
        v = v * valid
        a = v / (eps + v.sum(1, keepdim=True))
    elif merge_type == PCM_L2_MERGE:                                        # models.py:262-264
        if th.shape[1] != 1:
            raise ValueError("PCM merge_type 'l2' is only defined for f_dim == 1 (models.py:263 broadcasts [.., 1, f_dim] "
                             "against [.., f_dim, edges], models.py:396 reshapes the result to one row per node)")
        v = torch.cat([torch.exp(5.0 * (-(th - shifted(ph, o)) ** 2)) for o in offs], dim=1) * valid
        a = v / v.sum(1, keepdim=True)
    elif merge_type in PCM_GEO_MERGES:                                      # models.py:287-299
        geo = pcm_geo_feature(p_enc_dim, (D, H, W), f.dtype).unsqueeze(0).expand(B, -1, D, H, W)
        gth, gph = _lin(p, "geo_theta", geo), _lin(p, "geo_phi", geo)
        if merge_type == "att_is_all":
            logits = torch.stack([((th + gth) * shifted(ph + gph, o)).sum(1) for o in offs], dim=1)
        else:
            app = torch.stack([(th * shifted(ph, o)).sum(1) for o in offs], dim=1)
            if merge_type == "scaled_dot_product_geo_relu":
                app = F.relu(app)
            logits = app + torch.stack([(gth * shifted(gph, o)).sum(1) for o in offs], dim=1)
        a = _pcm_logits("scaled_dot_product", logits, valid, deg)
    else:
        logits = torch.stack([(th * shifted(ph, o)).sum(1) for o in offs], dim=1)   # [B, E, D, H, W]
        a = _pcm_logits(merge_type, logits, valid, deg)
    for _ in range(non_local_iter):
        g = _lin(p, "G", cam)                                                # [B, g_dim, D, H, W]
        y = sum(a[:, e:e + 1] * shifted(g, o) for e, o in enumerate(offs))
        refined = _lin(p, "r", y)
        cam = refined + cam if residual else refined
    return cam


def pcm_forward_literal(p, cam, f, k_size=3, connectivity=2, self_loop=True,
                        merge_type="scaled_dot_product_relu", p_enc_dim=0):
    """The same, written the way DGL executes it: one node at a time, its mailbox = the in-grid
    neighbours (init_graph's interior/side split collapses to "neighbours inside the grid"),
    compute_cross_x's permutes and matmuls spelled out (models.py:365-397).  Tiny grids only."""
    B, C, D, H, W = f.shape
    offs = pcm_offsets(k_size, connectivity, self_loop)
    out = torch.zeros_like(cam)
    lin = lambda name, x: x if (name + ".weight") not in p else F.linear(x, p[name + ".weight"], p[name + ".bias"])
    geo = pcm_geo_feature(p_enc_dim, (D, H, W), f.dtype).unsqueeze(0).expand(B, -1, D, H, W) if p_enc_dim > 0 else None
    for z in range(D):
        for y in range(H):
            for x in range(W):
                nb = [(z + o[0], y + o[1], x + o[2]) for o in offs]
                nb = [q for q in nb if 0 <= q[0] < D and 0 <= q[1] < H and 0 <= q[2] < W]
                f_agg = torch.stack([f[:, :, q[0], q[1], q[2]] for q in nb], 0)          # [E, B, C]
                cam_agg = torch.stack([cam[:, :, q[0], q[1], q[2]] for q in nb], 0)      # [E, B, g_ch]
                x_phi = lin("phi", f_agg).permute(1, 2, 0)                                # [B, F, E]
                x_theta = lin("theta", f[:, :, z, y, x]).unsqueeze(1)                     # [B, 1, F]
                fm = torch.matmul(x_theta, x_phi)                                         # [B, 1, E]
                if merge_type in PCM_GEO_MERGES:                                          # merge_func, models.py:287-299
                    g_agg = torch.stack([geo[:, :, q[0], q[1], q[2]] for q in nb], 0)     # [E, B, p_enc]
                    x_gphi = lin("geo_phi", g_agg).permute(1, 2, 0)                       # [B, Fg, E]
                    x_gtheta = lin("geo_theta", geo[:, :, z, y, x]).unsqueeze(1)          # [B, 1, Fg]
                    if merge_type == "att_is_all":
                        fm = torch.matmul(x_theta + x_gtheta, x_phi + x_gphi)
                    else:
                        if merge_type == "scaled_dot_product_geo_relu":
                            fm = F.relu(fm)
                        fm = fm + torch.matmul(x_gtheta, x_gphi)
                    fm = fm / np.sqrt(fm.shape[-1])
                if merge_type in ("scaled_dot_product_relu", "smrelu", "l2smrelu"):
                    fm = F.relu(fm)
                if merge_type in ("l2sm", "l2smrelu"):
                    fm = F.normalize(fm, dim=-1)
                if merge_type in ("scaled_dot_product", "scaled_dot_product_relu"):
                    fm = fm / np.sqrt(fm.shape[-1])
                if merge_type == "smscaled":
                    fm = fm / 0.01
                if merge_type in PCM_SUM_MERGES:                                          # merge_func, models.py:300-302, 307-320
                    if merge_type == "cosine":
                        fm = F.cosine_similarity(x_theta.transpose(-1, -2), x_phi, dim=-2).unsqueeze(-2)
                        f_sm = fm / fm.sum(dim=-1, keepdim=True)
                    else:
                        fm = fm / (1.0 + torch.abs(x_theta.transpose(-1, -2) - x_phi).sum(dim=-2, keepdim=True))
                        if merge_type == "heu1":
                            with torch.no_grad():
                                mask_f = torch.ones_like(fm)
                                mask_f[fm < 0.03] = 0.0
                                fm = fm * mask_f
                        else:
                            fm = F.relu(fm)
                        f_sm = fm / (1e-7 + fm.sum(dim=-1, keepdim=True))
                elif merge_type == PCM_L2_MERGE:                                           # merge_func, models.py:262-264
                    fm = torch.exp(5.0 * (-(x_theta - x_phi) ** 2))
                    f_sm = fm / fm.sum(dim=-1, keepdim=True)
                else:
                    f_sm = F.softmax(fm, dim=-1)
                x_g = lin("G", cam_agg).permute(1, 0, 2)                                  # [B, E, g_dim]
                yv = torch.matmul(f_sm, x_g).squeeze(1)                                   # [B, g_dim]
                out[:, :, z, y, x] = lin("r", yv)
    return out


def dc3dat_forward(cfg, params, buffers, x, training=False, norm_method="bn", attention=True):
    """DC3DATGeneric.forward (models.py:543-597) + apply_attention (498-506) for `cfg` (a MODEL dict
    of st_dram_ref_att.py without 'method').  Returns (dense_outs, refined_dense_outs, attention_features).  Blocks are run
    without torch.utils.checkpoint (use for values/gradients; the checkpoint double update of
    BatchNorm buffers is a DC3D-level effect covered by dc3d_forward).  attention=False replaces the
    PCM by a pass-through (what the golden vector pins)."""
    L = cfg["n_layers"]
    at_layers = list(cfg["at_layers"])
    at_size = tuple(cfg["at_spatial_size"])
    feats, att = [], ([x] if -1 in at_layers else [])
    nc = 0

    def reshape(t):
        nonlocal nc
        pfx = f"reshape.{nc}"
        nc += 1
        y = conv3d(t.detach(), params[pfx + ".0.weight"], params[pfx + ".0.bias"], 0)
        rm, rv = buffers[pfx + ".1.running_mean"], buffers[pfx + ".1.running_var"]
        if training:
            buffers[pfx + ".1.num_batches_tracked"] += 1
        y = F.batch_norm(y, rm, rv, params[pfx + ".1.weight"], params[pfx + ".1.bias"], training, BN_MOMENTUM, EPS)
        return F.relu(y)

    cur = x
    for n in range(L):
        y = conv_norm_act_stack(f"ds_modules.{n}", 2, params, buffers, cur, norm_method, training, _pads(cfg, n))
        cur = max_pool3d_2(y)
        feats.append(y)
        if n in at_layers:
            att.append(reshape(y))
    cur = conv_norm_act_stack("bg", 2, params, buffers, cur, norm_method, training, _pads(cfg, L))
    if L in at_layers:
        att.append(reshape(cur))
    for idx, skip in enumerate(reversed(feats)):
        if cfg.get("stacking", 0) == idx:
            break
        up = upsample_trilinear_ac(cur, scale_factor=tuple(cfg["upsample_sf"])
                                   if isinstance(cfg["upsample_sf"], (tuple, list)) else cfg["upsample_sf"])
        cur = conv_norm_act_stack(f"us_modules.{idx}", 2, params, buffers, crop_concat_5d(up, skip), norm_method,
                                  training, _pads(cfg, L + 1 + idx))
        if L + idx + 1 in at_layers:
            att.append(reshape(cur))
    dense = conv3d(cur, params["top_layer.weight"], params["top_layer.bias"], 0)
    dense = upsample_trilinear_ac(dense, size=tuple(x.shape[-3:]))
    att = torch.cat([upsample_trilinear_ac(t, size=at_size) for t in att], dim=1)
    cam = upsample_trilinear_ac(dense, size=at_size)
    if attention:
        pp = {k[len("attention_module."):]: v for k, v in params.items() if k.startswith("attention_module.")}
        cam = pcm_forward(pp, cam, att, cfg["at_k_size"], 2, cfg["at_self_loop"], cfg["at_merge_type"],
                          cfg["at_g_iter"], False)
    refined = upsample_trilinear_ac(cam, size=tuple(dense.shape[2:]))
    return dense, refined, att


def int_reg_refine_loss2(dense, refined, lobes, lesions, ctsses, freq_map, band_width=1e-2, smoothing=0.1):
    """IntRegRefineLoss.__call__ (metrics.py:360-373) for a model with distinct outputs
    (DC3DATGeneric): regression term and pseudo label from `dense`, segmentation term on `refined`."""
    probs = torch.sigmoid(dense)
    reg = reg_loss_with_probs(probs, lobes, lesions, ctsses, freq_map, band_width)
    with torch.no_grad():
        pd = probs.detach().clone()
        pd[lobes == 0] = 0.0
        pseudo = ((pd > 0.5) & (lesions > 0)).to(dense.dtype)
        keep = torch.tensor([0.0 if float(c) < 1e-7 else 1.0 for c in ctsses], dtype=dense.dtype,
                            device=dense.device).view(-1, 1, 1, 1, 1)
        pseudo = pseudo * keep
    seg = boot_bce(torch.sigmoid(refined), pseudo, lobes > 0, smoothing)
    return reg, seg


def fused_loss_math(dense, batch, refined=None, smoothing=0.1, eps=1e-7):
    """IntRegRefineLoss (metrics.py:360-373) in the *fused formulation* that csrc/loss.hip implements (sums
    over the batch of per-voxel terms, instead of the reference's per-sample numpy round trips), spelled with
    torch ops on a `dram_amd.train_step.Batch`.  The CPU tests check it against the reference's golden loss
    vectors, the GPU tests check the kernels against the same vectors: the formulation and its
    implementation are pinned separately.  Returns (reg_loss, seg_loss)."""
    pd = torch.sigmoid(dense)
    p = pd if refined is None or refined is dense else torch.sigmoid(refined)
    B = p.shape[0]
    lobes = batch.lobes
    inside = (lobes > 0).to(p.dtype)
    # compute_reg_loss_with_probs (metrics.py:158-177): hinge on the lobe-mean probability of dense_outs
    pred_ratio = (pd * inside).view(B, -1).sum(-1) / inside.view(B, -1).sum(-1)
    lo, hi = batch.targets[:, 0], batch.targets[:, 1]
    K = (0.5 * (hi - lo)) ** 2
    reg = torch.clamp((pred_ratio - (hi + lo) / 2.0) ** 2 - K, min=0.0) / batch.weight
    reg_loss = reg.sum()
    # compute_seg_loss (metrics.py:331-358): pseudo label from dense_outs, then BootBinCrossEntropy
    # (metrics.py:17-51) on the refined probabilities
    with torch.no_grad():
        t = ((pd > 0.5) & (lobes != 0) & (batch.lesions > 0)).to(p.dtype) * batch.keep
    outside = 1.0 - inside
    n_out = outside.sum()
    bceo = -(torch.log((1.0 - p).clamp(eps, 1.0 - eps)) * outside).sum() / n_out      # outside the lobe t == 0
    n_in = inside.sum()
    alpha = (1.0 - (t * inside).sum() / n_in).clamp(0.25, 0.75)
    pt = (p * t + (1.0 - p) * (1.0 - t)).clamp(eps, 1.0 - eps)
    w = (alpha * t + (1.0 - alpha) * (1.0 - t)) * inside
    bce = -(torch.log(pt) * w).sum() / w.sum()
    ph = torch.maximum(p, 1.0 - p).clamp(eps, 1.0 - eps)                               # p if p > 0.5 else 1 - p
    boot = -(torch.log(ph) * inside).sum() / n_in
    seg_loss = bceo + (1.0 - smoothing) * bce + smoothing * boot
    return reg_loss, seg_loss
