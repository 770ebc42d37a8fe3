"""GPU parity of SURVEY row N2: the PCM local-attention kernels (csrc/pcm.hip) and DC3DATGeneric.

PCM itself is PARITY UNPINNED (the reference needs DGL, absent): the kernels are compared with the
oracle's restatement in fp64 (itself cross-checked against a literal node-by-node restatement in
tests/test_oracle_golden.py).  Everything of DC3DATGeneric around the PCM call is compared with the
reference's own outputs (tests/golden/dc3dat_slim.npz, generated with a pass-through attention)."""
import os

import numpy as np
import pytest
import torch

from oracle import dram_oracle as O
from dram_amd.configs import SLIM_ATT

pytestmark = pytest.mark.gpu
TOL = 1e-4


def rel(got, ref):
    ref = torch.as_tensor(ref).double()
    got = torch.as_tensor(got).detach().cpu().double()
    return ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()


@pytest.mark.parametrize("merge", O.PCM_DOT_MERGES)
@pytest.mark.parametrize("shape,self_loop,iters,residual,conn", [
    ((5, 4, 7), False, 1, False, 2), ((1, 2, 3), True, 1, False, 2), ((8, 8, 8), False, 2, True, 2), ((6, 5, 4), True, 1, False, 1),
    ((4, 6, 5), False, 1, False, 3)])
def test_pcm_module_matches_oracle(merge, shape, self_loop, iters, residual, conn):
    import models
    torch.manual_seed(3)
    B, C, Fd, G, Gd = 2, 9, 4, 2, 3
    m = models.PCM(shape, C, G, Fd, 0, Gd, iters, 3, merge_type=merge, self_loop=self_loop, connectivity=conn,
                   residual=residual, p_enc_dim=0)
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(3.0)         # spread the logits so the softmax is far from uniform
    g = torch.Generator().manual_seed(4)
    cam = torch.randn((B, G) + shape, generator=g)
    f = torch.randn((B, C) + shape, generator=g)
    if merge == "smscaled":     # logits / 0.01: keep the softmax out of saturation (gradients would all be ~1e-12)
        f = f * 0.03
    gout = torch.randn((B, G) + shape, generator=g)
    # oracle, fp64
    p64 = {k: v.detach().double().requires_grad_(True) for k, v in m.state_dict().items()}
    cam64, f64 = cam.double().requires_grad_(True), f.double().requires_grad_(True)
    ref = _oracle_pcm(p64, cam64, f64, conn, self_loop, merge, iters, residual)
    (ref * gout.double()).sum().backward()
    # device
    m = m.cuda()
    camg, fg = cam.cuda().requires_grad_(True), f.cuda().requires_grad_(True)
    out = m(camg, fg)
    (out * gout.cuda()).sum().backward()
    assert rel(out, ref) <= TOL
    assert rel(camg.grad, cam64.grad) <= TOL
    assert rel(fg.grad, f64.grad) <= TOL
    # phi.bias shifts every logit of a node equally: its exact gradient is 0 for the plain softmax merges, so
    # parameter gradients are compared on the scale of the largest one
    scale = max(v.grad.abs().max().item() for v in p64.values())
    for k, p in m.named_parameters():
        err = (p.grad.detach().cpu().double() - p64[k].grad).abs().max().item()
        assert err <= TOL * max(scale, 1e-30), (k, err, scale)


@pytest.mark.parametrize("merge", O.PCM_GEO_MERGES)
@pytest.mark.parametrize("shape,self_loop,iters,residual,conn,geo_f", [
    ((5, 4, 7), False, 1, False, 2, 4), ((1, 2, 3), True, 1, False, 2, 4), ((6, 6, 6), False, 2, True, 2, 4), ((4, 5, 6), True, 1, False, 3, 0)])
def test_pcm_geo_merges_match_oracle(merge, shape, self_loop, iters, residual, conn, geo_f):
    """The geo family of merge_func (models.py:287-299): appearance term + positional-encoding term (sin/cos encodings of
    build_geo_feature projected by geo_theta / geo_phi); geo_f = 0: Identity projections.  PARITY UNPINNED like all of
    PCM; checked against the oracle's fp64 restatement, values and every gradient (geo_theta / geo_phi included)."""
    import models
    torch.manual_seed(5)
    B, C, G, Gd, P = 2, 7, 2, 3, 12
    Fd = 4 if geo_f else P              # att_is_all adds the two feature vectors: f_dim == geo_f_dim
    if merge != "att_is_all" and not geo_f:
        Fd = 5
    m = models.PCM(shape, C, G, Fd, geo_f, Gd, iters, 3, merge_type=merge, self_loop=self_loop, connectivity=conn,
                   residual=residual, p_enc_dim=P)
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(2.0)
    g = torch.Generator().manual_seed(6)
    cam = torch.randn((B, G) + shape, generator=g)
    f = torch.randn((B, C) + shape, generator=g)
    gout = torch.randn((B, G) + shape, generator=g)
    p64 = {k: v.detach().double().requires_grad_(True) for k, v in m.state_dict().items()}
    cam64, f64 = cam.double().requires_grad_(True), f.double().requires_grad_(True)
    ref = O.pcm_forward(p64, cam64, f64, 3, conn, self_loop, merge, iters, residual, p_enc_dim=P)
    (ref * gout.double()).sum().backward()
    m = m.cuda()
    camg, fg = cam.cuda().requires_grad_(True), f.cuda().requires_grad_(True)
    out = m(camg, fg)
    (out * gout.cuda()).sum().backward()
    assert rel(out, ref) <= TOL
    assert rel(camg.grad, cam64.grad) <= TOL
    assert rel(fg.grad, f64.grad) <= TOL
    scale = max(v.grad.abs().max().item() for v in p64.values() if v.grad is not None)
    names = dict(m.named_parameters())
    assert (geo_f == 0) or {"geo_theta.weight", "geo_phi.weight"} <= set(names)
    for k, p in names.items():
        err = (p.grad.detach().cpu().double() - p64[k].grad).abs().max().item()
        assert err <= TOL * max(scale, 1e-30), (k, err, scale)


@pytest.mark.parametrize("merge", O.PCM_SUM_MERGES)
@pytest.mark.parametrize("shape,self_loop,iters,residual,conn", [
    ((5, 4, 7), False, 1, False, 2), ((1, 2, 3), True, 1, False, 2), ((6, 6, 6), False, 2, True, 2), ((4, 5, 6), True, 1, False, 3)])
def test_pcm_sum_merges_match_oracle(merge, shape, self_loop, iters, residual, conn):
    """cosine / heu1 / heu2 (models.py:300-302, 307-320): similarities normalised by their sum over a node's edges.
    PARITY UNPINNED like all of PCM; against the oracle's fp64 restatement, values and every gradient.  Positive features
    keep the sums away from zero (the reference divides by the bare sum for 'cosine'); heu1's 0.03 mask is exercised by the
    scale of the features (about a third of the similarities fall below it)."""
    import models
    torch.manual_seed(9)
    B, C, Fd, G, Gd = 2, 6, 5, 2, 3
    m = models.PCM(shape, C, G, Fd, 0, Gd, iters, 3, merge_type=merge, self_loop=self_loop, connectivity=conn,
                   residual=residual, p_enc_dim=0)
    with torch.no_grad():
        for k, p in m.named_parameters():
            if k.startswith(("theta", "phi")):
                p.abs_()
                if merge == "heu1":
                    p.mul_(0.12)        # similarities of about 0.01 ... 0.07: on both sides of the 0.03 mask
    g = torch.Generator().manual_seed(10)
    cam = torch.randn((B, G) + shape, generator=g)
    f = torch.rand((B, C) + shape, generator=g)
    gout = torch.randn((B, G) + shape, generator=g)
    p64 = {k: v.detach().double().requires_grad_(True) for k, v in m.state_dict().items()}
    cam64, f64 = cam.double().requires_grad_(True), f.double().requires_grad_(True)
    ref = O.pcm_forward(p64, cam64, f64, 3, conn, self_loop, merge, iters, residual)
    (ref * gout.double()).sum().backward()
    if merge == "heu1":                 # the mask really cuts: some, not all, same-node similarities lie below 0.03
        with torch.no_grad():
            th = torch.einsum("bc...,fc->bf...", f64, p64["theta.weight"]) + p64["theta.bias"].view(1, -1, 1, 1, 1)
            ph = torch.einsum("bc...,fc->bf...", f64, p64["phi.weight"]) + p64["phi.bias"].view(1, -1, 1, 1, 1)
            u = (th * ph).sum(1) / (1.0 + (th - ph).abs().sum(1))
            assert 0.02 < (u < 0.03).double().mean().item() < 0.98
    m = m.cuda()
    camg, fg = cam.cuda().requires_grad_(True), f.cuda().requires_grad_(True)
    out = m(camg, fg)
    (out * gout.cuda()).sum().backward()
    assert rel(out, ref) <= TOL
    assert rel(camg.grad, cam64.grad) <= TOL
    if merge == "heu1":     # the features only enter through theta / phi, which this attention does not differentiate (below)
        assert f64.grad is None and (fg.grad is None or fg.grad.abs().max().item() == 0.0)
    else:
        assert rel(fg.grad, f64.grad) <= TOL
    scale = max(v.grad.abs().max().item() for v in p64.values() if v.grad is not None)
    for k, p in m.named_parameters():
        if merge == "heu1" and k.startswith(("theta", "phi")):
            # reference models.py:311-314: the masked similarities are formed under no_grad -- no gradient reaches theta / phi
            assert p64[k].grad is None and (p.grad is None or p.grad.abs().max().item() == 0.0), k
            continue
        err = (p.grad.detach().cpu().double() - p64[k].grad).abs().max().item()
        assert err <= TOL * max(scale, 1e-30), (k, err, scale)


@pytest.mark.parametrize("shape,self_loop,conn", [((5, 4, 7), False, 2), ((1, 2, 3), True, 2), ((4, 5, 6), True, 3)])
def test_pcm_l2_merge_matches_oracle(shape, self_loop, conn):
    """merge_type 'l2' (the constructor default, models.py:238,262-264) for f_dim == 1, the one width for which the reference's
    broadcast and reshape are defined; any other width raises.  PARITY UNPINNED like all of PCM; against the oracle's fp64
    restatement (which spells out the reference's exp / sum form), values and every gradient."""
    import models
    torch.manual_seed(11)
    B, C, G, Gd = 2, 6, 2, 3
    m = models.PCM(shape, C, G, 1, 0, Gd, 1, 3, merge_type="l2", self_loop=self_loop, connectivity=conn, p_enc_dim=0)
    g = torch.Generator().manual_seed(12)
    cam = torch.randn((B, G) + shape, generator=g)
    f = torch.rand((B, C) + shape, generator=g)
    gout = torch.randn((B, G) + shape, generator=g)
    p64 = {k: v.detach().double().requires_grad_(True) for k, v in m.state_dict().items()}
    cam64, f64 = cam.double().requires_grad_(True), f.double().requires_grad_(True)
    ref = O.pcm_forward(p64, cam64, f64, 3, conn, self_loop, "l2", 1, False)
    (ref * gout.double()).sum().backward()
    m = m.cuda()
    camg, fg = cam.cuda().requires_grad_(True), f.cuda().requires_grad_(True)
    out = m(camg, fg)
    (out * gout.cuda()).sum().backward()
    assert rel(out, ref) <= TOL
    assert rel(camg.grad, cam64.grad) <= TOL
    assert rel(fg.grad, f64.grad) <= TOL
    scale = max(v.grad.abs().max().item() for v in p64.values())
    for k, p in m.named_parameters():
        err = (p.grad.detach().cpu().double() - p64[k].grad).abs().max().item()
        assert err <= TOL * max(scale, 1e-30), (k, err, scale)


def test_pcm_l2_merge_saturated_inputs_stay_finite():
    """The one deliberate numerical deviation of merge_type 'l2' (round-3 advisor note): the reference forms
    exp(-5 (theta - phi_e)^2) / sum_e exp(...) (models.py:262-264); when every edge of a node underflows (|theta - phi| > ~4.2 in
    fp32) that is 0 / 0 = NaN.  The device path evaluates the same quantity as a softmax over the edges (the common term
    -5 theta^2 cancels, the maximum is subtracted): finite weights that sum to 1 and equal the fp64 softmax of the logits --
    the limit the reference's expression has just before it underflows.  Pinned here so that the behaviour is a decision."""
    from dram_amd import functional as HF
    shape = (4, 5, 6)
    g = torch.Generator().manual_seed(21)
    theta = torch.full((1, 1) + shape, 30.0) + torch.randn((1, 1) + shape, generator=g)
    phi = torch.randn((1, 1) + shape, generator=g) * 2.0                     # |theta - phi| ~ 30: exp(-4500) = 0 even in fp64
    offs = [tuple(int(v) for v in o) for o in O.pcm_offsets(3, 2, False)]
    attn = HF.pcm_attention(theta.cuda(), phi.cuda(), offs, "l2").cpu().double()          # [B, E, D, H, W]
    assert torch.isfinite(attn).all()
    naive = torch.exp(-5.0 * (theta.double()[0, 0, 2, 2, 3] - phi.double()[0, 0, 2, 2, 4]) ** 2)
    assert float(naive) == 0.0                                                # the reference's numerator underflows ...
    # interior node (2, 2, 3): all 18 edges exist; weights = softmax_e(-5 (theta - phi_e)^2)
    z, y, x = 2, 2, 3
    logits = torch.tensor([-5.0 * (float(theta[0, 0, z, y, x]) - float(phi[0, 0, z + dz, y + dy, x + dx])) ** 2 for dz, dy, dx in offs],
                          dtype=torch.float64)
    want = torch.softmax(logits, 0)
    got = attn[0, :, z, y, x]
    assert abs(float(got.sum()) - 1.0) <= 1e-5
    assert (got - want).abs().max().item() <= 1e-4 * want.max().item() + 1e-7


def _oracle_pcm(p, cam, f, conn, self_loop, merge, iters, residual):
    return O.pcm_forward(p, cam, f, 3, conn, self_loop, merge, iters, residual)


def test_pcm_offsets_match_oracle():
    import models
    for conn in (1, 2, 3):
        for sl in (False, True):
            for k in (3, 5):
                m = models.PCM((4, 4, 4), 3, 1, 2, 0, 2, 1, k, merge_type="sm", self_loop=sl, connectivity=conn, p_enc_dim=0)
                assert sorted(m.init_graph()) == sorted(tuple(int(v) for v in o) for o in O.pcm_offsets(k, conn, sl))
    with pytest.raises(ValueError, match="f_dim == 1"):      # 'l2' with f_dim = 2: the reference's broadcast is undefined
        m = models.PCM((4, 4, 4), 3, 1, 2, 0, 2, 1, 3, merge_type="l2", p_enc_dim=0).cuda()
        m(torch.zeros(1, 1, 4, 4, 4, device="cuda"), torch.zeros(1, 3, 4, 4, 4, device="cuda"))
    with pytest.raises(NotImplementedError):
        m = models.PCM((4, 4, 4), 3, 1, 2, 0, 2, 1, 3, merge_type="no_such_merge", p_enc_dim=0).cuda()
        m(torch.zeros(1, 1, 4, 4, 4, device="cuda"), torch.zeros(1, 3, 4, 4, 4, device="cuda"))


def _load_att(golden_dir):
    import models
    z = np.load(os.path.join(golden_dir, "dc3dat_slim.npz"))
    tag = "slim_att"
    sd = {k[len(tag + "/sd/"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(tag + "/sd/")}
    m = models.DC3DATGeneric(**SLIM_ATT)
    m.load_state_dict(sd)           # strict: the reference's 108 keys, names and shapes
    return z, tag, sd, m


def test_dc3dat_wiring_matches_reference_golden(golden_dir):
    z, tag, sd, m = _load_att(golden_dir)

    class PassThrough(torch.nn.Module):     # what oracle/make_golden.py:gen_att put in the reference
        def forward(self, cam, feats, args=None):
            self.seen = feats
            return cam
    m.attention_module = PassThrough()
    m = m.cuda().train()
    x = torch.from_numpy(z[tag + "/x"]).cuda()
    d0, d1 = m(x, None)
    feats = m.attention_module.seen
    assert rel(d0, z[tag + "/dense"]) <= TOL and rel(d1, z[tag + "/refined"]) <= TOL and rel(feats, z[tag + "/feats"]) <= TOL
    t = lambda k: torch.from_numpy(z[tag + k]).cuda()
    ((d0 * t("/gout0")).sum() + (d1 * t("/gout1")).sum() + (feats * t("/goutf")).sum()).backward()
    grads = {k[len(tag + "/grad/"):]: z[k] for k in z.files if k.startswith(tag + "/grad/")}
    got = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
    assert set(got) == set(grads)
    gmax = max(float(np.abs(v).max()) for v in grads.values())
    for k in grads:
        if k.startswith("reshape.") and k.endswith(".0.bias"):
            # a conv bias in front of BatchNorm: the exact gradient is 0, both sides hold rounding noise
            assert float(got[k].abs().max()) <= 1e-5 * gmax and float(np.abs(grads[k]).max()) <= 1e-5 * gmax, k
        elif k.startswith(("reshape.", "top_layer", "us_modules.2")):
            assert rel(got[k], grads[k]) <= 2e-3, k      # BatchNorm network in fp32: see test_gpu_parity.check_grads
    for k in z.files:
        if k.startswith(tag + "/sd_after/") and "running" in k:
            name = k[len(tag + "/sd_after/"):]
            assert rel(m.state_dict()[name], z[k]) <= TOL, name


def test_dc3dat_with_attention_matches_oracle(golden_dir):
    z, tag, sd, m = _load_att(golden_dir)
    params, buffers = O.split_state_dict({k: v.double() if v.is_floating_point() else v.clone() for k, v in sd.items()})
    for p in params.values():
        p.requires_grad_(True)
    x = torch.from_numpy(z[tag + "/x"])
    dense_r, refined_r, _ = O.dc3dat_forward(SLIM_ATT, params, buffers, x.double(), training=True, attention=True)
    g1 = torch.from_numpy(z[tag + "/gout1"])
    (refined_r * g1.double()).sum().backward()
    m = m.cuda().train()
    d0, d1 = m(x.cuda(), None)
    assert d0 is not d1
    (d1 * g1.cuda()).sum().backward()
    assert rel(d0, dense_r) <= TOL and rel(d1, refined_r) <= TOL
    for k in ("attention_module.theta.weight", "attention_module.phi.weight", "attention_module.G.weight",
              "attention_module.r.weight", "attention_module.r.bias", "reshape.0.0.weight", "reshape.1.1.bias",
              "top_layer.weight"):
        got = dict(m.named_parameters())[k].grad
        assert rel(got, params[k].grad) <= 5e-3, k


def test_fused_loss_two_outputs_matches_reference_golden(golden_dir):
    from dram_amd.train_step import Batch, DeviceIntRegRefineLoss
    z = np.load(os.path.join(golden_dir, "loss2.npz"))
    t = lambda k: torch.from_numpy(z[k]).cuda()
    batch = Batch(t("images"), t("lobes"), t("lesions"), list(z["ctss"]), {k: 1.0 / 6 for k in range(6)}, band_width=1e-2)
    dense, refined = t("dense").requires_grad_(True), t("refined").requires_grad_(True)
    reg, seg = DeviceIntRegRefineLoss(1e-2, 0.1)(dense, batch, refined=refined)
    assert abs(reg.item() - float(z["reg"])) <= 1e-5 * max(1.0, abs(float(z["reg"])))
    assert abs(seg.item() - float(z["seg"])) <= 1e-5 * max(1.0, abs(float(z["seg"])))
    (2.0 * reg + 1.0 * seg).backward()
    assert rel(dense.grad, z["gdense"]) <= TOL and rel(refined.grad, z["grefined"]) <= TOL


def test_trainer_step_dc3dat(golden_dir):
    """One DataParallelTrainer step drives DC3DATGeneric end to end (two outputs through the fused loss)."""
    from dram_amd.train_step import DataParallelTrainer, synthetic_batch
    z, tag, sd, m = _load_att(golden_dir)
    m = m.cuda().train()
    before = {k: v.detach().clone() for k, v in m.named_parameters()}
    tr = DataParallelTrainer(m, torch.optim.Adam(m.parameters(), lr=1e-3))
    reg, seg = tr.step(synthetic_batch(4, 16, 100, torch.device("cuda")), micro_batch=2)
    assert torch.isfinite(reg) and torch.isfinite(seg)
    moved = [k for k, v in m.named_parameters() if not torch.equal(v, before[k])]
    assert any(k.startswith("attention_module.") for k in moved) and any(k.startswith("ds_modules.0") for k in moved)


def test_dc3dat_runs_on_the_fused_engine_and_equals_the_per_op_path(golden_dir):
    """DC3DATGeneric's U-Net goes through dram_amd/engine.py (one autograd node; the tapped feature maps come out of it
    detached, reference models.py:556,566,578) and gives the step of the per-op path: both outputs, every gradient, the
    BatchNorm buffers (flagged blocks updated twice, with DC3DATGeneric's flag index n_layers + 1 + idx, models.py:573)."""
    from dram_amd import engine
    res = {}
    for fused in (False, True):
        z, tag, sd, m = _load_att(golden_dir)
        with torch.no_grad():       # pre-activations away from the ReLU threshold (see tests/test_gpu_engine.py)
            for mod in m.modules():
                if isinstance(mod, torch.nn.BatchNorm3d):
                    sign = torch.where(torch.arange(mod.bias.numel()) % 3 == 2, -1.0, 1.0)
                    mod.bias.copy_(3.0 * sign * mod.weight.abs().clamp_min(0.1))
        m = m.cuda().train()
        m.fused = fused
        assert engine.supports(m)
        x = torch.from_numpy(z[tag + "/x"]).cuda()
        d0, d1 = m(x, None)
        assert (type(d0.grad_fn).__name__ == "DC3DFusedFnBackward") == fused
        t = lambda k: torch.from_numpy(z[tag + k]).cuda()
        ((d0 * t("/gout0").abs()).sum() + (d1 * t("/gout1").abs()).sum()).backward()
        res[fused] = (d0.detach(), d1.detach(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None},
                      {k: v.clone() for k, v in m.state_dict().items() if "running" in k or "num_batches" in k})
    a, b = res[False], res[True]
    a = (a[0].cpu(), a[1].cpu(), {k: v.cpu() for k, v in a[2].items()}, {k: v.cpu() for k, v in a[3].items()})
    assert rel(b[0], a[0]) <= 2e-5 and rel(b[1], a[1]) <= 2e-5
    assert set(a[2]) == set(b[2])
    gmax = max(float(v.abs().max()) for v in a[2].values())
    for k in a[2]:
        if k.startswith("reshape.") and k.endswith(".0.bias"):
            continue            # a conv bias in front of BatchNorm: exactly zero in theory, rounding noise on both sides
        assert rel(b[2][k], a[2][k]) <= 1e-4 or float((b[2][k].cpu() - a[2][k]).abs().max()) <= 1e-6 * gmax, k
    for k in a[3]:
        assert rel(b[3][k].double(), a[3][k].double()) <= 1e-5, k
