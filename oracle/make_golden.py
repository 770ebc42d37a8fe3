"""Generate tests/golden/*.npz by running the REFERENCE itself on CPU.

Runs only in the build container (needs /root/reference); the fixtures it writes
are plain data (inputs + expected outputs) and are committed.  Usage:

    python oracle/make_golden.py            # writes tests/golden/*.npz

How the reference is imported: dram/parts.py imports cleanly.  dram/models.py and
dram/metrics.py import dgl / SimpleITK / cv2 / skimage at module level (never
touched by DC3D or IntRegRefineLoss); those packages are not installed, so this
process -- and only this process -- pre-seeds sys.modules with empty stand-in
modules for them.  metrics.py hard-codes `.cuda()`; inside this process only,
torch.Tensor.cuda is replaced by a no-op so the loss runs on CPU.
"""
import os
import sys
import types
import warnings

import numpy as np
import torch

REF = "/root/reference/dram"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def _import_reference():
    warnings.filterwarnings("ignore", category=SyntaxWarning)
    for name in ("dgl", "cv2", "SimpleITK", "skimage", "skimage.filters", "skimage.filters.thresholding",
                 "skimage.exposure", "skimage.measure", "skimage.morphology", "skimage.transform",
                 "tensorboardX", "seaborn"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = []
            sys.modules[name] = m
    sitk = sys.modules["SimpleITK"]
    for c in ("sitkNearestNeighbor", "sitkLinear", "sitkGaussian", "sitkLabelGaussian", "sitkBSpline",
              "sitkHammingWindowedSinc", "sitkCosineWindowedSinc", "sitkWelchWindowedSinc",
              "sitkLanczosWindowedSinc"):
        setattr(sitk, c, 0)
    sys.modules["skimage.filters.thresholding"].threshold_otsu = None
    sys.modules["skimage.filters"].threshold_otsu = None
    sys.modules["skimage.exposure"].equalize_hist = None
    sys.path.insert(0, REF)
    import parts
    import models
    return parts, models


def _np(t):
    return t.detach().cpu().numpy().copy()


def _save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"wrote {path}: {os.path.getsize(path) / 1024:.1f} KiB")


def _block_case(tag, block, inputs, out_names, arrs, seed):
    """Run a reference block fwd+bwd with fixed upstream grads, record everything."""
    g = torch.Generator().manual_seed(seed)
    for k, v in block.state_dict().items():
        arrs[f"{tag}/sd/{k}"] = _np(v)
    xs = []
    for i, x in enumerate(inputs):
        x = x.clone().requires_grad_(True)
        xs.append(x)
        arrs[f"{tag}/in{i}"] = _np(x)
    outs = block(*xs)
    if not isinstance(outs, tuple):
        outs = (outs,)
    loss = 0
    for nm, o in zip(out_names, outs):
        arrs[f"{tag}/out/{nm}"] = _np(o)
        go = torch.randn(o.shape, generator=g)
        arrs[f"{tag}/gout/{nm}"] = _np(go)
        loss = loss + (o * go).sum()
    loss.backward()
    for i, x in enumerate(xs):
        arrs[f"{tag}/gin{i}"] = _np(x.grad)
    for k, p in block.named_parameters():
        arrs[f"{tag}/gparam/{k}"] = _np(p.grad)
    for k, v in block.state_dict().items():   # BN buffers after the training forward
        if "running" in k or "num_batches" in k:
            arrs[f"{tag}/sd_after/{k}"] = _np(v)
    # eval-mode forward with the updated buffers
    block.eval()
    with torch.no_grad():
        outs = block(*[x.detach() for x in xs])
    if not isinstance(outs, tuple):
        outs = (outs,)
    for nm, o in zip(out_names, outs):
        arrs[f"{tag}/eval/{nm}"] = _np(o)
    block.train()


def gen_blocks(parts):
    arrs = {}
    g = torch.Generator().manual_seed(1234)
    shapes = {"even": (6, 10, 12), "odd": (7, 9, 11)}
    for norm in ("bn", "ln", "in", "bnt", "bntna", "lnna", None):
        for sname, (D, H, W) in shapes.items():
            torch.manual_seed(11)
            blk = parts.ConvPoolBlock5d([3, 4], [4, 6], 0, (3, 3), norm is None, (1, 1), 2, 2, 0,
                                        dropout=0.0, norm_method=norm, act_method="relu")
            # non-trivial affine parameters
            for m in blk.modules():
                if isinstance(m, (torch.nn.BatchNorm3d, torch.nn.GroupNorm)) and m.weight is not None:
                    m.weight.data = torch.rand(m.weight.shape, generator=g) + 0.5
                    m.bias.data = torch.randn(m.bias.shape, generator=g) * 0.3
            x = torch.randn(2, 3, D, H, W, generator=g)
            _block_case(f"convpool/{norm}/{sname}", blk, [x], ["y", "pooled"], arrs, seed=5)
    for norm in ("bn", "ln"):
        torch.manual_seed(12)
        blk = parts.ConvBlock5d([5, 8], [8, 7], 0, 3, False, 1, 0.0, norm_method=norm)
        x = torch.randn(2, 5, 5, 6, 9, generator=g)
        _block_case(f"conv/{norm}", blk, [x], ["y"], arrs, seed=6)
    torch.manual_seed(13)
    blk = parts.ConvBlock5d([5, 8], [8, 7], 0, 3, True, 1, 0.0, lite=True)
    x = torch.randn(2, 5, 5, 6, 9, generator=g)
    _block_case("conv/lite", blk, [x], ["y"], arrs, seed=7)
    for norm in ("bn", "in"):
        torch.manual_seed(14)
        blk = parts.UpsampleConvBlock5d([10, 5], [5, 4], 0, (2, 2, 2), (3, 3), False, (1, 1),
                                        dropout=0.0, norm_method=norm)
        lo = torch.randn(2, 6, 3, 5, 6, generator=g)
        cat = torch.randn(2, 4, 7, 11, 13, generator=g)   # larger than 2x -> centre crop with ceil offset
        _block_case(f"upconv/{norm}", blk, [lo, cat], ["y"], arrs, seed=8)
    # crop_concat_5d alone
    t1 = torch.randn(1, 2, 4, 5, 6, generator=g)
    t2 = torch.randn(1, 3, 7, 6, 9, generator=g)
    arrs["cropcat/t1"], arrs["cropcat/t2"] = _np(t1), _np(t2)
    arrs["cropcat/out"] = _np(parts.crop_concat_5d(t1, t2))
    _save("blocks", **arrs)


SLIM = {  # same topology / flags as st_dram_ref.MODEL, channels / 8
    "n_layers": 3,
    "in_ch_list": [1, 8, 16, 32, 96, 48, 24],
    "base_ch_list": [4, 8, 16, 32, 32, 16, 8],
    "end_ch_list": [8, 16, 32, 64, 32, 16, 8],
    "kernel_sizes": [(3, 3)] * 7,
    "stacking": 3,
    "padding_list": [(1, 1)] * 7,
    "checkpoint_layers": [0, 1, 0, 1, 0, 1, 0],
    "dropout": 0.0,
    "upsample_ksize": (3, 3, 3),
    "upsample_sf": (2, 2, 2),
    "out_ch": 1,
}


def _model_case(models, cfg, norm, shape, tag, arrs, store_sd, n_grad_keys=None):
    torch.manual_seed(0)
    model = models.DC3D(**cfg, norm_method=norm) if norm != "bn" else models.DC3D(**cfg)
    model.init(models.HeNorm(mode="fan_in"))
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    if store_sd:
        for k, v in sd0.items():
            arrs[f"{tag}/sd/{k}"] = _np(v)
    else:   # checksums only: the test re-creates the weights from the same seed
        for k, v in sd0.items():
            vf = v.double()
            arrs[f"{tag}/sdsum/{k}"] = np.array([vf.sum().item(), (vf * vf).sum().item()])
    x = torch.rand(shape, generator=torch.Generator().manual_seed(1))
    arrs[f"{tag}/x"] = _np(x)
    model.eval()
    with torch.no_grad():
        arrs[f"{tag}/eval_out"] = _np(model(x)[0])
    model.train()
    d0, d1 = model(x, None)
    assert d0 is d1
    arrs[f"{tag}/train_out"] = _np(d0)
    gout = torch.randn(d0.shape, generator=torch.Generator().manual_seed(2)) / d0.numel()
    arrs[f"{tag}/gout"] = _np(gout)
    (d0 * gout).sum().backward()
    grads = {k: p.grad for k, p in model.named_parameters()}
    keys = list(grads) if n_grad_keys is None else n_grad_keys
    for k in keys:
        arrs[f"{tag}/grad/{k}"] = _np(grads[k])
    for k, gr in grads.items():
        arrs[f"{tag}/gradnorm/{k}"] = np.array(gr.double().norm().item())
    for k, v in model.state_dict().items():
        if "running" in k or "num_batches" in k:
            arrs[f"{tag}/sd_after/{k}"] = _np(v)


def gen_models(models):
    arrs = {}
    _model_case(models, SLIM, "bn", (2, 1, 32, 32, 32), "slim_bn", arrs, store_sd=True)
    few = ["top_layer.weight", "top_layer.bias", "ds_modules.0.conv_blocks.0.0.weight",
           "ds_modules.1.conv_blocks.1.0.weight", "bg.conv_blocks.0.1.weight", "bg.conv_blocks.0.1.bias",
           "us_modules.0.conv_blocks.0.0.weight", "us_modules.2.conv_blocks.1.0.weight"]
    _model_case(models, SLIM, "ln", (2, 1, 24, 16, 32), "slim_ln", arrs, store_sd=True, n_grad_keys=few)
    _model_case(models, SLIM, "in", (1, 1, 21, 18, 20), "slim_in_odd", arrs, store_sd=True,
                n_grad_keys=few)   # floor-pool / crop / final resize
    _save("dc3d_slim", **arrs)
    # full-width st_dram_ref model: weights are re-created from seed 0 by the test (65 MB otherwise)
    sys.path.insert(0, os.path.join(REF, "exp_settings"))
    import st_dram_ref
    cfg = dict(st_dram_ref.MODEL)
    cfg.pop("method")
    arrs = {}
    small = ["top_layer.weight", "top_layer.bias", "ds_modules.0.conv_blocks.0.0.weight",
             "ds_modules.0.conv_blocks.0.1.weight", "ds_modules.0.conv_blocks.0.1.bias",
             "us_modules.2.conv_blocks.1.1.weight", "bg.conv_blocks.1.1.bias"]
    _model_case(models, cfg, "bn", (1, 1, 32, 32, 32), "full_bn", arrs, store_sd=False, n_grad_keys=small)
    _save("dc3d_full", **arrs)


def gen_loss():
    torch.Tensor.cuda = lambda self, *a, **k: self     # generator process only (metrics.py:136,173)
    import metrics

    class Obj:
        ctss_frequency_map = {k: 1.0 / 6 for k in range(6)}
        debug_path = "/tmp/_dram_golden_dbg"
        epoch_n = 0
    arrs = {}
    g = torch.Generator().manual_seed(77)
    N, S = 6, 12
    zz, yy, xx = np.meshgrid(*[np.arange(S)] * 3, indexing="ij")
    lobe = (((zz - S / 2 + .5) ** 2 + (yy - S / 2 + .5) ** 2 + (xx - S / 2 + .5) ** 2) < (0.45 * S) ** 2)
    lobes = torch.from_numpy(lobe.astype(np.float32))[None, None].repeat(N, 1, 1, 1, 1)
    images = torch.rand(N, 1, S, S, S, generator=g) * lobes
    lesions = ((images > 0.7) & (lobes > 0)).float()
    ctss = [float(n % 6) for n in range(N)]
    dense = (torch.randn(N, 1, S, S, S, generator=g) * 2.0).requires_grad_(True)
    loss_fn = metrics.IntRegRefineLoss(band_width=1e-2, smoothing=0.1)

    def fake_model(imgs, lbs):
        return dense, dense
    fake_model.trace_path = None
    reg, seg = loss_fn(fake_model, images, lobes, lesions, ctss, obj=Obj(), metas=None)
    total = 2.0 * reg + 1.0 * seg
    total.backward()
    arrs.update(images=_np(images), lobes=_np(lobes), lesions=_np(lesions), ctss=np.array(ctss),
                dense=_np(dense), reg=np.array(reg.item()), seg=np.array(seg.item()), gdense=_np(dense.grad))
    _save("loss", **arrs)


def gen_misc(parts, models):
    """The less-travelled factory branches: act_wrapper "prelu" (parts.py:51-52) inside a ConvPoolBlock5d,
    a per-channel nn.PReLU on its own, and pooling_dense_features 'global_avg' / 'global_max' / default
    (models.py:37-49) with ties in the maximum."""
    arrs = {}
    g = torch.Generator().manual_seed(4321)
    torch.manual_seed(21)
    blk = parts.ConvPoolBlock5d([3, 4], [4, 6], 0, (3, 3), False, (1, 1), 2, 2, 0, dropout=0.0, norm_method="bn",
                                act_method="prelu")
    assert any(isinstance(m, torch.nn.PReLU) for m in blk.modules())
    x = torch.randn(2, 3, 7, 9, 11, generator=g)
    _block_case("convpool_prelu", blk, [x], ["y", "pooled"], arrs, seed=9)
    act = parts.act_wrapper("prelu", 5, 0.1)
    act.weight.data = torch.rand(5, generator=g) - 0.3
    x = torch.randn(3, 5, 4, 6, 7, generator=g)
    _block_case("prelu_c", act, [x], ["y"], arrs, seed=10)
    dense = torch.randn(2, 3, 5, 6, 7, generator=g).round()        # rounded: repeated maxima
    lungs = (torch.rand(2, 1, 5, 6, 7, generator=g) > 0.5).float()
    for method in ("global_avg", "global_max", "avg"):
        d = dense.clone().requires_grad_(True)
        out = models.pooling_dense_features(d, lungs, method)
        go = torch.randn(out.shape, generator=g)
        (out * go).sum().backward()
        arrs[f"pool/{method}/out"], arrs[f"pool/{method}/gout"], arrs[f"pool/{method}/gin"] = _np(out), _np(go), _np(d.grad)
    arrs["pool/dense"], arrs["pool/lungs"] = _np(dense), _np(lungs)
    _save("misc", **arrs)


def gen_transforms():
    """The OneShot tensor transforms (data_transforms.py:1140-1239) run by the reference on CPU tensors: every flip
    subset, every rot90 axis pair x times, Rescale3DOneShot by size (up, down, mixed) and by factor, for "#image"
    (trilinear, with the gradient of a random cotangent) and "#reference" (nearest) tensors."""
    import data_transforms as DT
    arrs = {}
    g = torch.Generator().manual_seed(99)
    x = torch.randn(2, 2, 5, 6, 7, generator=g)
    lab = (torch.rand(2, 1, 5, 6, 7, generator=g) * 5).floor()
    arrs["x"], arrs["lab"] = _np(x), _np(lab)
    cases = []
    for n in (1, 2, 3):
        for axes in __import__("itertools").combinations((2, 3, 4), n):
            cases.append(("flip", axes, None))
    for axes in __import__("itertools").permutations((2, 3, 4), 2):
        for k in (1, 2, 3):
            cases.append(("rot", axes, k))
    for i, (kind, axes, k) in enumerate(cases):
        t = DT.Flip3DOneShot(flip_axis=axes) if kind == "flip" else DT.Rotate903DOneShot(rotate_axis=axes, rotate_times=k)
        xi = x.clone().requires_grad_(True)
        out = t({"#image": xi, "meta": 1})["#image"]
        go = torch.randn(out.shape, generator=g)
        (out * go).sum().backward()
        arrs[f"pf/{i}/cfg"] = np.array([0 if kind == "flip" else 1, k or 0] + list(axes) + [0] * (3 - len(axes)))
        arrs[f"pf/{i}/naxes"] = np.array(len(axes))
        arrs[f"pf/{i}/out"], arrs[f"pf/{i}/gout"], arrs[f"pf/{i}/gin"] = _np(out), _np(go), _np(xi.grad)
    rs = [("size", (8, 9, 11)), ("size", (3, 4, 5)), ("size", (7, 3, 12)), ("size", (5, 6, 7)), ("factor", (1.5, 0.75, 2.0)),
          ("factor", (0.5, 0.5, 0.5))]
    for i, (mode, sf) in enumerate(rs):
        t = DT.Rescale3DOneShot(None, sf, mode=mode)
        xi = x.clone().requires_grad_(True)
        res = t({"#image": xi, "#reference": lab})
        out, outl = res["#image"], res["#reference"]
        go = torch.randn(out.shape, generator=g)
        (out * go).sum().backward()
        arrs[f"rs/{i}/mode"], arrs[f"rs/{i}/sf"] = np.array(0 if mode == "size" else 1), np.array(sf, dtype=np.float64)
        arrs[f"rs/{i}/out"], arrs[f"rs/{i}/gout"], arrs[f"rs/{i}/gin"], arrs[f"rs/{i}/lab"] = _np(out), _np(go), _np(xi.grad), _np(outl)
    # Rotate3DXOneShot (affine_grid + grid_sample; commented out of the reference's pool, but the class is there)
    for i, th in enumerate((0.3, 1.1, 2.6)):
        t = DT.Rotate3DXOneShot()
        t.theta = np.array([th])
        xi = x.clone().requires_grad_(True)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out = t({"#image": xi})["#image"]
        go = torch.randn(out.shape, generator=g)
        (out * go).sum().backward()
        arrs[f"rx/{i}/theta"] = np.array(th)
        arrs[f"rx/{i}/out"], arrs[f"rx/{i}/gout"], arrs[f"rx/{i}/gin"] = _np(out), _np(go), _np(xi.grad)
    _save("transforms", **arrs)


API_NAMES = ("Identity", "normal_wrapper", "crop_concat_5d", "act_wrapper", "checkpoint_wrapper", "ConvBlock5d",
             "UpsampleConvBlock5d", "ConvPoolBlock5d", "Initializer", "HeNorm", "pooling_dense_features", "DC3D", "PCM",
             "DC3DATGeneric")
# methods of the DGL message-passing formulation of PCM; the grid-stencil implementation has no counterpart
PCM_DGL_INTERNALS = ("build_geo_feature", "merge_func", "compute_cross_x", "message_func", "reduce_func")


def gen_signatures(parts, models):
    """The call signatures (inspect.signature strings) of the reference's public callables of parts.py / models.py and
    of their public methods: API facts the drop-in modules are checked against (tests/test_host_cpu.py)."""
    import inspect
    import json
    out = {}
    for mod in (parts, models):
        for name in API_NAMES:
            if not hasattr(mod, name):
                continue
            o = getattr(mod, name)
            entry = {"module": mod.__name__, "signature": str(inspect.signature(o.__init__ if inspect.isclass(o) else o))}
            if inspect.isclass(o):
                entry["methods"] = {k: str(inspect.signature(v)) for k, v in vars(o).items()
                                    if inspect.isfunction(v) and not k.startswith("__")}
            out[name] = entry
    path = os.path.join(OUT, "signatures.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(f"wrote {path}")


SLIM_ATT = dict(SLIM, at_spatial_size=(6, 5, 7), at_f_dim=4, at_g_dim=3, at_g_iter=1, at_k_size=3,
                at_merge_type="scaled_dot_product_relu", at_self_loop=False, at_layers=[-1, 0, 1],
                at_p_enc_dim=0, at_geo_f_dim=0)


def gen_att(models):
    """DC3DATGeneric (models.py:413-597) with everything of the reference run as is except the PCM
    (needs DGL, absent): `attention_module` is swapped for a pass-through that records its inputs.
    Pins the wiring around the attention (reshape convs on detached features, resizes, concat order,
    the two outputs) and the state-dict keys / init of the full module incl. PCM's Linear layers."""
    arrs = {}
    tag = "slim_att"
    torch.manual_seed(0)
    model = models.DC3DATGeneric(**SLIM_ATT)
    model.init(models.HeNorm(mode="fan_in"))
    for k, v in model.state_dict().items():
        arrs[f"{tag}/sd/{k}"] = _np(v)

    class PassThrough(torch.nn.Module):
        def forward(self, cam, feats, args=None):
            self.seen = (cam, feats)
            return cam
    model.attention_module = PassThrough()
    x = torch.rand((2, 1, 16, 16, 16), generator=torch.Generator().manual_seed(1))
    arrs[f"{tag}/x"] = _np(x)
    model.train()
    d0, d1 = model(x, None)
    cam, feats = model.attention_module.seen
    arrs[f"{tag}/dense"], arrs[f"{tag}/refined"] = _np(d0), _np(d1)
    arrs[f"{tag}/cam"], arrs[f"{tag}/feats"] = _np(cam), _np(feats)
    g = torch.Generator().manual_seed(2)
    g0 = torch.randn(d0.shape, generator=g) / d0.numel()
    g1 = torch.randn(d1.shape, generator=g) / d1.numel()
    gf = torch.randn(feats.shape, generator=g) / feats.numel()
    arrs[f"{tag}/gout0"], arrs[f"{tag}/gout1"], arrs[f"{tag}/goutf"] = _np(g0), _np(g1), _np(gf)
    ((d0 * g0).sum() + (d1 * g1).sum() + (feats * gf).sum()).backward()
    for k, p in model.named_parameters():
        if p.grad is not None:
            arrs[f"{tag}/grad/{k}"] = _np(p.grad)
    for k, v in model.state_dict().items():
        if "running" in k or "num_batches" in k:
            arrs[f"{tag}/sd_after/{k}"] = _np(v)
    _save("dc3dat_slim", **arrs)


def gen_loss2():
    """IntRegRefineLoss for a model whose two outputs differ (DC3DATGeneric)."""
    torch.Tensor.cuda = lambda self, *a, **k: self     # generator process only (metrics.py:136,173)
    import metrics

    class Obj:
        ctss_frequency_map = {k: 1.0 / 6 for k in range(6)}
        debug_path = "/tmp/_dram_golden_dbg"
        epoch_n = 0
    g = torch.Generator().manual_seed(78)
    N, S = 6, 12
    zz, yy, xx = np.meshgrid(*[np.arange(S)] * 3, indexing="ij")
    lobe = (((zz - S / 2 + .5) ** 2 + (yy - S / 2 + .5) ** 2 + (xx - S / 2 + .5) ** 2) < (0.45 * S) ** 2)
    lobes = torch.from_numpy(lobe.astype(np.float32))[None, None].repeat(N, 1, 1, 1, 1)
    images = torch.rand(N, 1, S, S, S, generator=g) * lobes
    lesions = ((images > 0.7) & (lobes > 0)).float()
    ctss = [float(n % 6) for n in range(N)]
    dense = (torch.randn(N, 1, S, S, S, generator=g) * 2.0).requires_grad_(True)
    refined = (torch.randn(N, 1, S, S, S, generator=g) * 2.0).requires_grad_(True)
    loss_fn = metrics.IntRegRefineLoss(band_width=1e-2, smoothing=0.1)

    def fake_model(imgs, lbs):
        return dense, refined
    fake_model.trace_path = None
    reg, seg = loss_fn(fake_model, images, lobes, lesions, ctss, obj=Obj(), metas=None)
    (2.0 * reg + 1.0 * seg).backward()
    _save("loss2", images=_np(images), lobes=_np(lobes), lesions=_np(lesions), ctss=np.array(ctss),
          dense=_np(dense), refined=_np(refined), reg=np.array(reg.item()), seg=np.array(seg.item()),
          gdense=_np(dense.grad), grefined=_np(refined.grad))


def gen_clean(models):
    """A well-conditioned BatchNorm fixture for model-level gradient parity.

    Gradients of a BatchNorm U-Net's early layers are ill-conditioned in fp32 only through ReLU-mask flips:
    elements whose pre-activation (BatchNorm output) lies within rounding of 0 switch sides between
    implementations, and each flip moves d(beta) by 1/sqrt(M).  Here the weights are CHOSEN so that this cannot
    happen: layer by layer in forward order, every BatchNorm bias gets a small per-channel offset (|delta| <= 0.02)
    such that, in a float64 run of the reference model, no pre-activation of that channel lies within MARGIN
    (1e-4 of the channel's standard deviation, >= 100x the fp32 rounding of these values) of zero.  With that
    state dict the reference's own fp32 run below is the golden: fp32 implementations must then agree to
    rounding on every parameter gradient, early layers included (tests/test_gpu_parity.py)."""
    import copy
    MARGIN = 1e-4
    torch.manual_seed(0)
    model = models.DC3D(**SLIM)
    model.init(models.HeNorm(mode="fan_in"))
    g = torch.Generator().manual_seed(41)
    with torch.no_grad():       # non-trivial affine parameters (HeNorm leaves them at 1 / 0)
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm3d):
                m.weight.copy_(1.0 + 0.2 * torch.randn(m.weight.shape, generator=g))
                m.bias.copy_(0.1 * torch.randn(m.bias.shape, generator=g))
    x = torch.rand((2, 1, 16, 16, 16), generator=torch.Generator().manual_seed(42))
    names = [n for n, m in model.named_modules() if isinstance(m, torch.nn.BatchNorm3d)]   # forward order
    cand = torch.arange(-80, 81, dtype=torch.float64) * 2.5e-4

    def preact(layer):
        m64 = copy.deepcopy(model).double().train()
        got = {}

        def hook(mod, inputs, output):          # (returns None: a returned tensor would replace the module's output)
            got.setdefault("z", output.detach().clone())   # before the in-place ReLU that follows
        h = dict(m64.named_modules())[layer].register_forward_hook(hook)
        with torch.no_grad():
            m64(x.double())
        h.remove()
        return got["z"]

    worst = 1e9
    for name in names:
        z = preact(name)
        zc = z.transpose(0, 1).reshape(z.shape[1], -1)                                   # [C, M]
        sd = zc.std(dim=1, unbiased=False).clamp_min(1e-12)
        gap = (zc[:, None, :] + cand[None, :, None]).abs().min(dim=2).values / sd[:, None]   # [C, ncand]
        score = gap - 1e-3 * cand.abs()[None, :]                                         # prefer small offsets among good ones
        score = torch.where(gap >= 4 * MARGIN, score, gap - 1.0)
        best = score.argmax(dim=1)
        bn = dict(model.named_modules())[name]
        with torch.no_grad():
            bn.bias.add_(cand[best].float())
        z = preact(name)                                                                 # with the fp32-rounded bias
        zc = z.transpose(0, 1).reshape(z.shape[1], -1)
        m = (zc.abs().min(dim=1).values / zc.std(dim=1, unbiased=False).clamp_min(1e-12)).min().item()
        assert m >= MARGIN, (name, m)
        worst = min(worst, m)
    print(f"clean fixture: smallest |pre-activation| / sigma over all BatchNorm layers = {worst:.2e} (margin {MARGIN:.0e})")
    arrs = {"slim_bn_clean/margin": np.array(worst)}
    for k, v in model.state_dict().items():
        arrs[f"slim_bn_clean/sd/{k}"] = _np(v)
    arrs["slim_bn_clean/x"] = _np(x)
    m64 = copy.deepcopy(model).double().train()
    model.train()
    d0, _ = model(x, None)
    arrs["slim_bn_clean/train_out"] = _np(d0)
    gout = torch.randn(d0.shape, generator=torch.Generator().manual_seed(43)) / d0.numel()
    arrs["slim_bn_clean/gout"] = _np(gout)
    (d0 * gout).sum().backward()
    e0, _ = m64(x.double(), None)
    (e0 * gout.double()).sum().backward()
    g64 = dict(m64.named_parameters())
    worst_g = 0.0
    for k, p in model.named_parameters():
        arrs[f"slim_bn_clean/grad/{k}"] = _np(p.grad)
        worst_g = max(worst_g, ((p.grad.double() - g64[k].grad).abs().max() / g64[k].grad.abs().max()).item())
    print(f"clean fixture: the reference's own fp32-vs-fp64 gradient error, worst tensor: {worst_g:.2e}")
    arrs["slim_bn_clean/ref_fp32_vs_fp64"] = np.array(worst_g)
    _save("dc3d_clean", **arrs)


CASES_AFFLOSS = (("all3", 0), ("all3b", 12), ("fliprot", 7), ("rescale", 5), ("none", 1))


def gen_affloss():
    """IntRegAffRefineLoss (metrics.py:376-462) of the reference with a closed-form 3-output stand-in for the model
    (dense / refined / 2-channel cls as smooth functions of the input and three scalar parameters), the random
    affine transform drawn under fixed `random` / `numpy.random` seeds.  Stores inputs, the transform that was
    drawn, the three loss values and the gradients of the parameters."""
    import random
    torch.Tensor.cuda = lambda self, *a, **k: self     # generator process only (metrics.py:136,173)
    import metrics

    class Obj:
        ctss_frequency_map = {k: 1.0 / 6 for k in range(6)}
        debug_path = "/tmp/_dram_golden_dbg"
        epoch_n = 0
    g = torch.Generator().manual_seed(91)
    N, S = 4, 12
    zz, yy, xx = np.meshgrid(*[np.arange(S)] * 3, indexing="ij")
    lobe = (((zz - S / 2 + .5) ** 2 + (yy - S / 2 + .5) ** 2 + (xx - S / 2 + .5) ** 2) < (0.45 * S) ** 2)
    lobes = torch.from_numpy(lobe.astype(np.float32))[None, None].repeat(N, 1, 1, 1, 1)
    images = torch.rand(N, 1, S, S, S, generator=g) * lobes
    lesions = ((images > 0.7) & (lobes > 0)).float()
    ctss = [float(1 + n % 5) for n in range(N)]
    theta = torch.tensor([1.5, -0.4, 0.8], requires_grad=True)

    def fake_model(imgs, lbs):
        a, b, c = theta[0], theta[1], theta[2]
        D, H, W = imgs.shape[-3:]      # position-dependent terms: the stand-in must not commute with flips / rotations
        rz = torch.linspace(0.0, 1.0, D).view(1, 1, D, 1, 1)
        rx = torch.linspace(0.0, 1.0, W).view(1, 1, 1, 1, W)
        dense = a * (imgs - 0.5) * 4.0 + b + 0.6 * c * rx - 0.4 * rz
        refined = 0.7 * dense - c * imgs
        cls = torch.cat([a * imgs + rz, imgs * imgs + b * c * rx], dim=1)
        return dense, refined, cls
    fake_model.trace_path = None
    arrs = {}
    for case, seed in CASES_AFFLOSS:
        random.seed(seed)
        np.random.seed(seed)
        loss_fn = metrics.IntRegAffRefineLoss(rescale_jitter=[8, 10, 12, 14], band_width=5e-2, smoothing=0.05)
        T_holder = {}
        orig = loss_fn.get_affine_transform

        def spy():
            T_holder["T"] = orig()
            return T_holder["T"]
        loss_fn.get_affine_transform = spy
        theta.grad = None
        reg, aff, seg = loss_fn(fake_model, images, lobes, lesions, ctss, obj=Obj(), metas=None)
        (2.0 * reg + 0.5 * aff + 1.0 * seg).backward()
        desc = [f"{type(t).__name__}:" + ",".join(f"{k}={v}" for k, v in sorted(t.__dict__.items()) if k != "rescale_factor_pool")
                for t in T_holder["T"].p]
        print("affloss", case, seed, desc, reg.item(), aff.item(), seg.item(), theta.grad.tolist())
        arrs[f"{case}/seed"] = np.array(seed)
        arrs[f"{case}/T"] = np.array("|".join(desc))
        arrs[f"{case}/out"] = np.array([reg.item(), aff.item(), seg.item()])
        arrs[f"{case}/gtheta"] = _np(theta.grad)
    _save("affloss", images=_np(images), lobes=_np(lobes), lesions=_np(lesions), ctss=np.array(ctss), theta=_np(theta), **arrs)


def gen_infer_tail():
    """The numpy helpers around the model call of LesionSegTest.run that do NOT need SimpleITK / skimage, run from the
    reference's utils.py on small random volumes: windowing (utils.py:189-198, with its default span and to_span=(0, 1) as
    job_runner.py:1006 calls it), binary_cam's 8-bit view of it (utils.py:233: windowing(., from_span=(0, 1)).astype(uint8)),
    find_crops (utils.py:244-254), IOU and Dice (utils.py:437-446).  (binary_cam itself needs skimage's threshold_otsu: the
    Otsu step stays unpinned.)"""
    import utils as U
    rng = np.random.default_rng(77)
    shape = (14, 19, 23)
    scan = rng.integers(-1400, 700, size=shape).astype(np.int16)
    scan.flat[:8] = [-1150, -1151, 350, 351, -1050, -950, 349, -1149]          # window edges and exact 1/15 steps
    lobe = np.zeros(shape, dtype=np.uint8)
    lobe[2:9, 3:12, 4:15] = 1
    lobe[6:13, 8:17, 10:22] = 3
    a = (rng.random(shape) > 0.6).astype(np.uint8)
    b = (rng.random(shape) > 0.5).astype(np.uint8)
    w_scan = U.windowing(scan, to_span=(0, 1))
    view8 = U.windowing(w_scan[lobe > 0], from_span=(0, 1)).astype(np.uint8)
    arrs = dict(scan=scan, lobe=lobe, a=a, b=b, w_scan=w_scan, view8=view8, hist=np.bincount(view8, minlength=256),
                iou=np.array(U.IOU(a > 0, b > 0, 1e-5)), dice=np.array(U.Dice(a > 0, b > 0, 1e-5)),
                iou_empty=np.array(U.IOU(np.zeros(shape, bool), np.zeros(shape, bool), 1e-5)))
    for i, (lab, spacing, border) in enumerate(((1, (1.0, 0.7, 0.7), 5), (3, (2.5, 1.0, 1.0), 5), (3, (1.0, 1.0, 1.0), 0))):
        sl = U.find_crops(lobe == lab, spacing, border)
        arrs[f"crop{i}"] = np.array([lab, *spacing, border, *[s.start for s in sl], *[s.stop for s in sl]], dtype=np.float64)
    _save("infer_tail", **arrs)


if __name__ == "__main__":
    torch.set_num_threads(8)
    parts, models = _import_reference()
    only = set(sys.argv[1:])          # e.g. `python oracle/make_golden.py att loss2`; default: everything
    if not only or "blocks" in only:
        gen_blocks(parts)
    if not only or "models" in only:
        gen_models(models)
    if not only or "clean" in only:
        gen_clean(models)
    if not only or "att" in only:
        gen_att(models)
    if not only or "misc" in only:
        gen_misc(parts, models)
    if not only or "transforms" in only:
        gen_transforms()
    if not only or "signatures" in only:
        gen_signatures(parts, models)
    if not only or "loss" in only:
        gen_loss()
    if not only or "loss2" in only:
        gen_loss2()
    if not only or "affloss" in only:
        gen_affloss()
    if not only or "infer_tail" in only:
        gen_infer_tail()
