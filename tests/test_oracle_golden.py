"""Pins the CPU oracle (oracle/dram_oracle.py) to golden vectors produced by
running the reference itself (oracle/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import dram_oracle as O
from oracle.make_golden import SLIM


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def _t(a):
    return torch.from_numpy(np.asarray(a))


def _sub(z, prefix):
    return {k[len(prefix):]: z[k] for k in z.files if k.startswith(prefix)}


def _close(a, b, rtol=1e-5, atol=1e-6):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = np.asarray(b)
    scale = max(1.0, float(np.abs(b).max())) if b.size else 1.0
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol * scale)


NORMS = ["bn", "ln", "in", "bnt", "bntna", "lnna", "None"]


@pytest.mark.parametrize("norm", NORMS)
@pytest.mark.parametrize("shape", ["even", "odd"])
def test_convpool_block(golden_dir, norm, shape):
    z = _load(golden_dir, "blocks")
    tag = f"convpool/{norm}/{shape}/"
    sd = {"b." + k: _t(v) for k, v in _sub(z, tag + "sd/").items()}
    params, buffers = O.split_state_dict(sd)
    for p in params.values():
        p.requires_grad_(True)
    x = _t(z[tag + "in0"]).requires_grad_(True)
    nm = None if norm == "None" else norm
    y = O.conv_norm_act_stack("b", 2, params, buffers, x, nm, True, [1, 1])
    pooled = O.max_pool3d_2(y)
    _close(y, z[tag + "out/y"])
    _close(pooled, z[tag + "out/pooled"])
    ((y * _t(z[tag + "gout/y"])).sum() + (pooled * _t(z[tag + "gout/pooled"])).sum()).backward()
    _close(x.grad, z[tag + "gin0"], rtol=1e-4, atol=1e-5)
    for k, g in _sub(z, tag + "gparam/").items():
        _close(params["b." + k].grad, g, rtol=1e-4, atol=1e-5)
    for k, v in _sub(z, tag + "sd_after/").items():
        _close(buffers["b." + k].float(), v.astype(np.float32))
    with torch.no_grad():
        ye = O.conv_norm_act_stack("b", 2, params, buffers, x, nm, False, [1, 1])
    _close(ye, z[tag + "eval/y"])


@pytest.mark.parametrize("norm", ["bn", "in"])
def test_upconv_block(golden_dir, norm):
    z = _load(golden_dir, "blocks")
    tag = f"upconv/{norm}/"
    sd = {"b." + k: _t(v) for k, v in _sub(z, tag + "sd/").items()}
    params, buffers = O.split_state_dict(sd)
    lo, cat = _t(z[tag + "in0"]).requires_grad_(True), _t(z[tag + "in1"]).requires_grad_(True)
    up = O.upsample_trilinear_ac(lo, scale_factor=(2, 2, 2))
    y = O.conv_norm_act_stack("b", 2, params, buffers, O.crop_concat_5d(up, cat), norm, True, [1, 1])
    _close(y, z[tag + "out/y"])
    (y * _t(z[tag + "gout/y"])).sum().backward()
    _close(lo.grad, z[tag + "gin0"], rtol=1e-4, atol=1e-5)
    _close(cat.grad, z[tag + "gin1"], rtol=1e-4, atol=1e-5)


def test_crop_concat(golden_dir):
    z = _load(golden_dir, "blocks")
    out = O.crop_concat_5d(_t(z["cropcat/t1"]), _t(z["cropcat/t2"]))
    assert np.array_equal(out.numpy(), z["cropcat/out"])
    assert O.crop_offsets((4, 5, 6), (7, 6, 9)) == (2, 1, 2)   # ceil((b-a)/2), parts.py:42-44


def test_numpy_definitions_agree_with_torch_ops():
    """First-principles numpy definitions == the ATen ops the oracle calls."""
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 3, 5, 6, 7, generator=g)
    w = torch.randn(4, 3, 3, 3, 3, generator=g)
    b = torch.randn(4, generator=g)
    _close(O.conv3d(x, w, b, 1), O.np_conv3d(x.numpy(), w.numpy(), b.numpy(), 1), rtol=1e-4, atol=1e-5)
    xr = torch.relu(x)   # many ties at 0, like post-ReLU feature maps
    out, idx = O.np_maxpool2(xr.numpy())
    tout, tidx = torch.nn.functional.max_pool3d(xr, 2, 2, 0, return_indices=True)
    assert np.array_equal(out, tout.numpy())
    # flat input index -> window-local 0..7 index
    D, H, W = xr.shape[2:]
    ti = tidx.numpy()
    tz, ty, tx = ti // (H * W), (ti // W) % H, ti % W
    local = (tz % 2) * 4 + (ty % 2) * 2 + (tx % 2)
    assert np.array_equal(local.astype(np.uint8), idx)
    for size in [(10, 12, 14), (9, 13, 8), (5, 6, 7)]:
        _close(O.upsample_trilinear_ac(x, size=size), O.np_trilinear_ac(x.numpy(), size), rtol=1e-5, atol=1e-6)
    gam, bet = torch.rand(3) + 0.5, torch.randn(3)
    yb, mean, var = O.np_batchnorm_train(x.numpy(), gam.numpy(), bet.numpy())
    _close(torch.nn.functional.batch_norm(x, None, None, gam, bet, True, 0.1, O.EPS), yb, rtol=1e-4, atol=1e-5)
    for G in (1, 3):
        _close(torch.nn.functional.group_norm(x, G, gam, bet, O.EPS), O.np_groupnorm(x.numpy(), G, gam.numpy(), bet.numpy()),
               rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("tag,norm", [("slim_bn", "bn"), ("slim_ln", "ln"), ("slim_in_odd", "in")])
def test_dc3d_slim(golden_dir, tag, norm):
    z = _load(golden_dir, "dc3d_slim")
    params, buffers = O.split_state_dict({k: _t(v) for k, v in _sub(z, tag + "/sd/").items()})
    x = _t(z[tag + "/x"])
    with torch.no_grad():
        ev = O.dc3d_forward(SLIM, params, buffers, x, training=False, norm_method=norm)
    _close(ev, z[tag + "/eval_out"], rtol=1e-4, atol=1e-5)
    for p in params.values():
        p.requires_grad_(True)
    out = O.dc3d_forward(SLIM, params, buffers, x, training=True, norm_method=norm)
    _close(out, z[tag + "/train_out"], rtol=1e-4, atol=1e-5)
    (out * _t(z[tag + "/gout"])).sum().backward()
    for k, g in _sub(z, tag + "/grad/").items():
        _close(params[k].grad, g, rtol=1e-3, atol=1e-4)
    for k, v in _sub(z, tag + "/sd_after/").items():
        _close(buffers[k].double(), v.astype(np.float64), rtol=1e-5, atol=1e-6)
    if norm == "bn":   # checkpointed blocks update their BN buffers twice per step
        assert int(buffers["ds_modules.1.conv_blocks.0.1.num_batches_tracked"]) == 2
        assert int(buffers["ds_modules.0.conv_blocks.0.1.num_batches_tracked"]) == 1


def test_loss(golden_dir):
    z = _load(golden_dir, "loss")
    dense = _t(z["dense"]).requires_grad_(True)
    freq = {k: 1.0 / 6 for k in range(6)}
    reg, seg = O.int_reg_refine_loss(dense, _t(z["lobes"]), _t(z["lesions"]), list(z["ctss"]), freq, 1e-2, 0.1)
    assert abs(reg.item() - float(z["reg"])) <= 1e-5 * max(1.0, abs(float(z["reg"])))
    assert abs(seg.item() - float(z["seg"])) <= 1e-5 * max(1.0, abs(float(z["seg"])))
    (2.0 * reg + 1.0 * seg).backward()
    _close(dense.grad, z["gdense"], rtol=1e-4, atol=1e-6)


# ---------------------------------------------------------------- N2: DC3DATGeneric / PCM
def test_dc3dat_wiring_matches_reference_golden(golden_dir):
    """Everything of DC3DATGeneric around the PCM call, against the reference run with a pass-through
    attention module (oracle/make_golden.py:gen_att)."""
    from oracle.make_golden import SLIM_ATT
    z = _load(golden_dir, "dc3dat_slim")
    tag = "slim_att"
    params, buffers = O.split_state_dict({k: _t(v) for k, v in _sub(z, tag + "/sd/").items()})
    for p in params.values():
        p.requires_grad_(True)
    x = _t(z[tag + "/x"])
    dense, refined, feats = O.dc3dat_forward(SLIM_ATT, params, buffers, x, training=True, attention=False)
    _close(dense, z[tag + "/dense"], rtol=1e-4, atol=1e-5)
    _close(refined, z[tag + "/refined"], rtol=1e-4, atol=1e-5)
    _close(feats, z[tag + "/feats"], rtol=1e-4, atol=1e-5)
    ((dense * _t(z[tag + "/gout0"])).sum() + (refined * _t(z[tag + "/gout1"])).sum()
     + (feats * _t(z[tag + "/goutf"])).sum()).backward()
    grads = _sub(z, tag + "/grad/")
    assert any(k.startswith("reshape.") for k in grads) and not any(k.startswith("attention_module") for k in grads)
    for k, g in grads.items():
        _close(params[k].grad, g, rtol=2e-3, atol=1e-4 * max(1e-3, float(np.abs(g).max())))
    for k, v in _sub(z, tag + "/sd_after/").items():
        if "num_batches" in k:   # the oracle runs flagged blocks once (no checkpoint re-run)
            continue
        if k.startswith("reshape."):
            _close(buffers[k].double(), v.astype(np.float64), rtol=1e-5, atol=1e-6)
    # PCM's parameters exist under the reference's names and shapes
    assert params["attention_module.theta.weight"].shape == (4, 9) and params["attention_module.G.weight"].shape == (3, 1)
    assert params["attention_module.r.weight"].shape == (1, 3) and params["attention_module.phi.bias"].shape == (4,)


@pytest.mark.parametrize("merge", O.PCM_DOT_MERGES)
@pytest.mark.parametrize("self_loop", [False, True])
def test_pcm_dense_restatement_equals_literal_one(merge, self_loop):
    """PARITY UNPINNED (DGL absent): two independent restatements of PCM agree with each other."""
    g = torch.Generator().manual_seed(5)
    r = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)
    p = {"theta.weight": r(4, 5), "theta.bias": r(4), "phi.weight": r(4, 5), "phi.bias": r(4),
         "G.weight": r(3, 2), "G.bias": r(3), "r.weight": r(2, 3), "r.bias": r(2)}
    for shape in [(4, 3, 5), (1, 2, 3)]:
        cam, f = r(2, 2, *shape), r(2, 5, *shape)
        a = O.pcm_forward(p, cam, f, 3, 2, self_loop, merge)
        b = O.pcm_forward_literal(p, cam, f, 3, 2, self_loop, merge)
        assert (a - b).abs().max().item() < 1e-10
    assert len(O.pcm_offsets(3, 2, False)) == 18 and len(O.pcm_offsets(3, 1, True)) == 7   # st_dram_ref_att: 18


@pytest.mark.parametrize("merge", O.PCM_SUM_MERGES)
def test_pcm_sum_merges_dense_restatement_equals_literal_one(merge):
    """PARITY UNPINNED (DGL absent): cosine / heu1 / heu2 (models.py:300-302, 307-320) in the two restatements."""
    g = torch.Generator().manual_seed(8)
    r = lambda *s: torch.rand(*s, generator=g, dtype=torch.float64) + 0.1        # positive features: sums away from zero
    p = {"theta.weight": r(4, 5), "theta.bias": r(4), "phi.weight": r(4, 5), "phi.bias": r(4),
         "G.weight": r(3, 2), "G.bias": r(3), "r.weight": r(2, 3), "r.bias": r(2)}
    for shape in [(4, 3, 5), (1, 2, 3)]:
        cam, f = r(2, 2, *shape), r(2, 5, *shape)
        a = O.pcm_forward(p, cam, f, 3, 2, False, merge)
        b = O.pcm_forward_literal(p, cam, f, 3, 2, False, merge)
        assert (a - b).abs().max().item() < 1e-10


def test_pcm_l2_and_heu1_gradient_semantics():
    """PARITY UNPINNED (DGL absent).  'l2' (models.py:262-264, f_dim = 1): the dense restatement equals the literal one, which
    spells the reference's expression out.  'heu1' (models.py:307-314): the masked similarities are formed under no_grad, so the
    attention is a constant of the graph -- theta / phi receive no gradient in either restatement."""
    g = torch.Generator().manual_seed(9)
    r = lambda *s: torch.rand(*s, generator=g, dtype=torch.float64) + 0.1
    p = {"theta.weight": r(1, 5), "theta.bias": r(1), "phi.weight": r(1, 5), "phi.bias": r(1),
         "G.weight": r(3, 2), "G.bias": r(3), "r.weight": r(2, 3), "r.bias": r(2)}
    for shape in [(4, 3, 5), (1, 2, 3)]:
        cam, f = r(2, 2, *shape), r(2, 5, *shape)
        a = O.pcm_forward(p, cam, f, 3, 2, False, "l2")
        b = O.pcm_forward_literal(p, cam, f, 3, 2, False, "l2")
        assert (a - b).abs().max().item() < 1e-10
    with pytest.raises(ValueError):
        O.pcm_forward({**p, "theta.weight": r(2, 5), "theta.bias": r(2), "phi.weight": r(2, 5), "phi.bias": r(2)},
                      r(1, 2, 2, 2, 2), r(1, 5, 2, 2, 2), 3, 2, False, "l2")
    p4 = {"theta.weight": r(4, 5), "theta.bias": r(4), "phi.weight": r(4, 5), "phi.bias": r(4),
          "G.weight": r(3, 2), "G.bias": r(3), "r.weight": r(2, 3), "r.bias": r(2)}
    for fwd in (O.pcm_forward, O.pcm_forward_literal):
        q = {k: v.clone().requires_grad_(True) for k, v in p4.items()}
        fwd(q, r(2, 2, 3, 2, 3), r(2, 5, 3, 2, 3), 3, 2, False, "heu1").sum().backward()
        assert q["theta.weight"].grad is None and q["phi.weight"].grad is None and q["G.weight"].grad is not None


@pytest.mark.parametrize("merge", O.PCM_GEO_MERGES)
@pytest.mark.parametrize("geo_f", [4, 0])
def test_pcm_geo_dense_restatement_equals_literal_one(merge, geo_f):
    """PARITY UNPINNED (DGL absent): the geo family of merge_func (models.py:287-299) in the two restatements; the
    positional encoding against the reference's formula evaluated by hand at one voxel."""
    g = torch.Generator().manual_seed(7)
    r = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)
    P = 12
    Fd = 4 if geo_f else (P if merge == "att_is_all" else 5)
    p = {"theta.weight": r(Fd, 5), "theta.bias": r(Fd), "phi.weight": r(Fd, 5), "phi.bias": r(Fd),
         "G.weight": r(3, 2), "G.bias": r(3), "r.weight": r(2, 3), "r.bias": r(2)}
    if geo_f:
        p.update({"geo_theta.weight": r(geo_f, P), "geo_theta.bias": r(geo_f), "geo_phi.weight": r(geo_f, P), "geo_phi.bias": r(geo_f)})
    for shape in [(4, 3, 5), (1, 2, 3)]:
        cam, f = r(2, 2, *shape), r(2, 5, *shape)
        a = O.pcm_forward(p, cam, f, 3, 2, False, merge, p_enc_dim=P)
        b = O.pcm_forward_literal(p, cam, f, 3, 2, False, merge, p_enc_dim=P)
        assert (a - b).abs().max().item() < 1e-10
    pe = O.pcm_geo_feature(P, (4, 3, 5), torch.float64)          # d_model = 4 per axis: sin, cos at frequencies 1, 1e-2
    z, y, x = 3, 1, 4
    want = [fn(c * w) for c in (z, y, x) for w in (1.0, 1e-2) for fn in (np.sin, np.cos)]
    assert np.allclose(pe[:, z, y, x].numpy(), want, rtol=1e-6, atol=1e-7)     # (the frequencies are fp32 powers, as in the reference)


def test_loss_two_outputs(golden_dir):
    z = _load(golden_dir, "loss2")
    dense, refined = _t(z["dense"]).requires_grad_(True), _t(z["refined"]).requires_grad_(True)
    freq = {k: 1.0 / 6 for k in range(6)}
    reg, seg = O.int_reg_refine_loss2(dense, refined, _t(z["lobes"]), _t(z["lesions"]), list(z["ctss"]), freq, 1e-2, 0.1)
    assert abs(reg.item() - float(z["reg"])) <= 1e-5 * max(1.0, abs(float(z["reg"])))
    assert abs(seg.item() - float(z["seg"])) <= 1e-5 * max(1.0, abs(float(z["seg"])))
    (2.0 * reg + 1.0 * seg).backward()
    _close(dense.grad, z["gdense"], rtol=1e-4, atol=1e-6)
    _close(refined.grad, z["grefined"], rtol=1e-4, atol=1e-6)


def test_dc3d_clean_fixture(golden_dir):
    """The well-conditioned BatchNorm fixture (oracle/make_golden.py:gen_clean: no pre-activation within 1e-4 sigma
    of zero, so no ReLU-mask flips between fp32 implementations): the oracle's fp32 gradients equal the reference's
    on EVERY parameter tensor to rounding, early layers included."""
    z = _load(golden_dir, "dc3d_clean")
    tag = "slim_bn_clean"
    assert float(z[tag + "/margin"]) >= 1e-4 and float(z[tag + "/ref_fp32_vs_fp64"]) < 1e-4
    params, buffers = O.split_state_dict({k: _t(v) for k, v in _sub(z, tag + "/sd/").items()})
    for p in params.values():
        p.requires_grad_(True)
    out = O.dc3d_forward(SLIM, params, buffers, _t(z[tag + "/x"]), training=True, norm_method="bn")
    _close(out, z[tag + "/train_out"], rtol=1e-4, atol=1e-5)
    (out * _t(z[tag + "/gout"])).sum().backward()
    grads = _sub(z, tag + "/grad/")
    assert len(grads) == len(params)
    for k, gref in grads.items():
        err = np.abs(params[k].grad.numpy() - gref).max() / np.abs(gref).max()
        assert err <= 1e-4, (k, err)


def test_inference_tail_helpers(golden_dir):
    """The numpy helpers around the model call of LesionSegTest.run (reference utils.py: windowing with its default span,
    binary_cam's 8-bit view, find_crops, IOU, Dice -- oracle/make_golden.py:gen_infer_tail ran the reference's own
    functions): the oracle's restatements are BIT-exact on them (integer / fp64 work).  The Otsu step of binary_cam is not
    covered (skimage absent: unpinned)."""
    z = _load(golden_dir, "infer_tail")
    scan, lobe = z["scan"], z["lobe"]
    w = O.windowing(scan, from_span=(-1150, 350), to_span=(0, 1))
    assert w.dtype == np.float64 and np.array_equal(w, z["w_scan"])
    view8 = O.windowing(w[lobe > 0], from_span=(0, 1), to_span=(0, 255)).astype(np.uint8)
    assert np.array_equal(view8, z["view8"])
    assert float(O.iou(z["a"] > 0, z["b"] > 0, 1e-5)) == float(z["iou"])
    assert float(O.dice(z["a"] > 0, z["b"] > 0, 1e-5)) == float(z["dice"])
    assert float(O.iou(np.zeros(3, bool), np.zeros(3, bool), 1e-5)) == float(z["iou_empty"]) == 1.0
    for i in range(3):
        c = z[f"crop{i}"]
        sl = O.find_crops(lobe == int(c[0]), tuple(c[1:4]), c[4])
        assert [s.start for s in sl] == list(c[5:8].astype(int)) and [s.stop for s in sl] == list(c[8:11].astype(int))
    # the tail itself: with no vessel mask and a brightness threshold below every voxel the post mask equals the plain one
    htp = (z["a"] * 0.7).astype(np.float32)
    pred, post, th2 = O.lesion_post_process(htp, scan, lobe, None, 0.5)
    assert np.array_equal(pred, z["a"]) and 0.0 <= th2 <= 1.0
    assert np.array_equal(post, (z["a"] > 0) & (w > th2))
    ves = z["b"]
    _, post_v, _ = O.lesion_post_process(htp, scan, lobe, ves, 0.5)
    assert np.array_equal(post_v, post & (ves == 0))


def test_resample_itk_restatement_properties():
    """`O.resample_itk` (the sitk.ResampleImageFilter call of utils.resample, restated; SimpleITK absent: unpinned) against what
    that call's published semantics imply, on cases small enough to spell out: the identity on an unchanged grid; nearest
    neighbour at half-integer positions rounds UP; voxels beyond size_in - 0.5 take the default value 0; linear agrees with
    the crop -> R^3 restatement (`resample_itk_linear`: the same call with spacing * in / out); int16 results are truncated."""
    rng = np.random.default_rng(2)
    a = rng.integers(-1000, 1000, size=(5, 6, 7)).astype(np.int16)
    sp = (1.5, 0.7, 0.7)
    for how in ("nearest", "linear"):
        assert np.array_equal(O.resample_itk(a, sp, sp, a.shape, how), a)
    # z spacing halved: c = 0, 0.5, 1, 1.5, ...: nearest takes floor(c + 0.5) = 0, 1, 1, 2, 2, ...; c = 4.5 (o = 9) is outside
    up = O.resample_itk(a, sp, (0.75, 0.7, 0.7), (11, 6, 7), "nearest")
    assert np.array_equal(up[:9], a[[0, 1, 1, 2, 2, 3, 3, 4, 4]]) and (up[9:] == 0).all()
    lin = O.resample_itk(a, sp, (0.75, 0.7, 0.7), (11, 6, 7), "linear")
    mid = (a[0].astype(np.float64) + a[1].astype(np.float64)) / 2.0
    assert np.array_equal(lin[1], np.trunc(mid).astype(np.int16)) and np.array_equal(lin[8], a[4]) and (lin[9:] == 0).all()
    f = rng.random((9, 12, 10)).astype(np.float32)
    size = (16, 7, 10)
    req = tuple(1.0 * n / m for n, m in zip(f.shape, size))
    assert np.abs(O.resample_itk(f, (1.0, 1.0, 1.0), req, size, "linear") - O.resample_itk_linear(f, size)).max() <= 1e-6
