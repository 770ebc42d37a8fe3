"""The fused engine (dram_amd/engine.py: norm statistics in the conv epilogue, normalise + ReLU on load, one autograd
node per network) against the per-op path (one autograd Function per op of the reference, pinned to the reference's
goldens in test_gpu_parity.py) on the same weights and inputs: outputs, every parameter gradient, the input gradient
and the BatchNorm buffers.  The two paths compute the activations with the same fmaf / fmaxf, so the only
differences are the summation trees of the norm statistics (conv-epilogue partials vs row chunks) and of nothing
else: agreement is at fp32 rounding level."""
import numpy as np
import pytest
import torch

from dram_amd.configs import SLIM

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def _run(model, x, gout, fused, need_dx):
    model.fused = fused
    for p in model.parameters():
        p.grad = None
    xg = x.clone().requires_grad_(need_dx)
    d0, d1 = model(xg)
    assert d0 is d1
    (d0 * gout).sum().backward()
    return (d0.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters()},
            xg.grad.clone() if need_dx else None,
            {k: v.clone() for k, v in model.state_dict().items() if "running" in k or "num_batches" in k})


CASES = [
    # norm, N, (D, H, W), need_dx
    ("bn", 2, (16, 16, 16), False),
    ("bn", 3, (21, 18, 20), True),      # odd sizes: floor pooling, centre crops, final resize; W % 4 == 0 at every level? no: 20,10,5
    ("ln", 2, (24, 16, 32), False),
    ("in", 1, (17, 19, 22), True),      # widths 22 / 11 / 5: backward-weights without the lazy path (materialised operands)
    ("bnt", 2, (16, 24, 8), False),     # BatchNorm without running statistics
    ("lnna", 2, (8, 8, 8), True),       # no affine parameters; 1^3 bottleneck
]


@pytest.mark.parametrize("norm,N,shape,need_dx", CASES)
def test_fused_engine_equals_per_op_path(norm, N, shape, need_dx):
    import models
    from dram_amd import engine
    torch.manual_seed(7)
    model = models.DC3D(**SLIM, norm_method=norm)
    model.init(models.HeNorm(mode="fan_in"))
    g = torch.Generator().manual_seed(8)
    # Non-trivial affine parameters, with biases of +-3 sigma: pre-activations then stay away from zero (density
    # 4e-3 there instead of 0.4), so that no ReLU mask bit can flip between the two paths because their statistics
    # differ in the last bit -- at the 2x2x2 bottleneck of these small volumes ONE flip moves a gradient by percents
    # (measured).  Spatially mixed masks are pinned by the reference goldens, which run through the engine too.
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, (torch.nn.BatchNorm3d, torch.nn.GroupNorm)) and m.weight is not None:
                m.weight.copy_(1.0 + 0.3 * torch.randn(m.weight.shape, generator=g))
                sign = torch.where(torch.arange(m.bias.numel()) % 3 == 2, -1.0, 1.0)
                m.bias.copy_(3.0 * sign * m.weight.abs())
    model = model.to(DEV).train()
    assert engine.supports(model)
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    x = torch.rand((N, 1) + shape, generator=g).to(DEV)
    # a POSITIVE upstream gradient: with a random-sign one the parameter gradients of the last layers are sums that
    # cancel to ~1/sqrt(n) of their terms, and a single ReLU-mask flip (an element whose pre-activation is within
    # rounding of zero lands on the other side because the two paths' statistics differ in the last bit) moves them
    # by percents -- measured: ONE flip of 24576 elements in us_modules.2 -> 3.4e-2 on that bias gradient, with
    # the fp64 oracle on the per-op side and torch's own GPU ops on the fused side (scripts/debug_engine4.py)
    gout = ((0.5 + torch.rand((N, 1) + shape, generator=g)) / x.numel()).to(DEV)
    ref = _run(model, x, gout, False, need_dx)
    model.load_state_dict(sd0)
    got = _run(model, x, gout, True, need_dx)
    assert _rel(got[0], ref[0]) <= 2e-5, ("out", _rel(got[0], ref[0]))
    worst = {k: _rel(got[1][k], ref[1][k]) for k in ref[1]}
    tol = 1e-4           # typical agreement 1e-6
    print(f"\n{norm} {N}x{shape}: worst gradient difference {max(worst.values()):.2e}")
    bad = {k: v for k, v in worst.items() if v > tol}
    assert not bad, bad
    if need_dx:
        assert _rel(got[2], ref[2]) <= tol
    for k, v in ref[3].items():
        assert _rel(got[3][k].double(), v.double()) <= 1e-5, k
    # eval mode (running statistics / per-sample statistics), no autograd
    model.eval()
    with torch.no_grad():
        model.fused = False
        e_ref = model(x)[0]
        model.fused = True
        e_got = model(x)[0]
    assert _rel(e_got, e_ref) <= 2e-5


def test_fused_engine_is_what_dc3d_runs_and_frees_its_tape():
    import models
    from dram_amd import engine
    torch.manual_seed(1)
    model = models.DC3D(**SLIM)
    model.init(models.HeNorm(mode="fan_in"))
    model = model.to(DEV).train()
    assert models.DC3D.fused and engine.supports(model)
    x = torch.rand(2, 1, 16, 16, 16, device=DEV)
    out, _ = model(x)
    assert type(out.grad_fn).__name__ == "DC3DFusedFnBackward"
    out.sum().backward()
    assert out.grad_fn.record is None if hasattr(out.grad_fn, "record") else True
    assert all(p.grad is not None for p in model.parameters())
    # recompute mode and unsupported networks fall back to the per-op path
    model.checkpoint_mode = "recompute"
    out2, _ = model(x)
    assert type(out2.grad_fn).__name__ != "DC3DFusedFnBackward"
    model.checkpoint_mode = "stats"
    prelu = models.DC3D(**SLIM, act_method="prelu")      # act_method reaches only the ConvPoolBlock5d blocks
    assert not engine.supports(prelu)


def test_fused_stats_epilogue_extreme_mean():
    """|mean| >> sigma: the conv-epilogue statistics are two-pass per wave and Chan-combined in fp64, so a large
    common offset does not cancel the variance (checked against an fp64 host computation)."""
    from dram_amd import _lib
    from dram_amd import functional as HF
    g = torch.Generator().manual_seed(3)
    N, Ci, Co, D, H, W = 2, 8, 40, 6, 12, 36
    x = torch.randn(N, Ci, D, H, W, generator=g)
    w = torch.randn(Co, Ci, 3, 3, 3, generator=g) * 1e-3
    w[:, :, 1, 1, 1] += 1.0                                  # y ~ sum of the inputs' centre taps ...
    x += 50.0                                                # ... of inputs with mean 50, sigma 1
    xd, wd = x.to(DEV), w.to(DEV)
    wt = HF._pack(wd, 0)
    y = torch.empty(N, Co, D, H, W, device=DEV)
    nparts = _lib.lib.dram_conv3d_k3_stats_parts(Ci, Co, D, H, W)
    parts = torch.full((N * Co * nparts * 3,), float("nan"), device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    _lib.call("dram_conv3d_k3_fwd_fused", xd.data_ptr(), Ci, None, 0, None, 0, None, 0, 0, 0, 0, 0, 0, 0, wt.data_ptr(), None,
              y.data_ptr(), parts.data_ptr(), nparts, N, Co, D, H, W, st)
    mean = torch.empty(Co, device=DEV); rstd = torch.empty(Co, device=DEV); coef = torch.empty(2 * N * Co, device=DEV)
    ws = torch.empty(max(16, _lib.lib.dram_norm_parts_ws_bytes(N, Co, nparts)), dtype=torch.uint8, device=DEV)
    _lib.call("dram_norm_finalize_parts", parts.data_ptr(), nparts, None, None, mean.data_ptr(), rstd.data_ptr(), coef.data_ptr(),
              None, None, 0.0, 1e-5, 0, 1, N, Co, D * H * W, ws.data_ptr(), ws.numel(), st)
    y64 = torch.nn.functional.conv3d(x.double(), w.double(), padding=1)
    m64 = y64.mean(dim=(0, 2, 3, 4))
    v64 = y64.var(dim=(0, 2, 3, 4), unbiased=False)
    assert _rel(y, y64) <= 1e-5
    assert _rel(mean, m64) <= 1e-6
    assert _rel(rstd, 1.0 / torch.sqrt(v64 + 1e-5)) <= 1e-4   # sigma/mean ~ 1e-2: the variance keeps 4+ digits through y's fp32 rounding


def test_fused_engine_sliced_and_fully_lazy(monkeypatch):
    """The memory-saving modes that large batches switch on by themselves -- every activated tensor lazy, the upsampled
    tensors never kept, upsampled-input stages executed in slices of ONE sample (statistics finalised once over all
    slices, backward-weights summed over slices, the gradient of the upsampled tensor never whole) -- forced here on a
    small batch: same step as the per-op path."""
    import models
    from dram_amd import engine
    monkeypatch.setattr(engine, "MEMORY_MODE", "manual")
    monkeypatch.setattr(engine, "MATERIALISE_BELOW", 0.0)
    monkeypatch.setattr(engine, "KEEP_UPSAMPLED_BELOW", 0.0)
    monkeypatch.setattr(engine, "SLICE_UPSAMPLED_ABOVE", 1e-12)
    torch.manual_seed(7)
    model = models.DC3D(**SLIM)
    model.init(models.HeNorm(mode="fan_in"))
    g = torch.Generator().manual_seed(9)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm3d):
                m.weight.copy_(1.0 + 0.3 * torch.randn(m.weight.shape, generator=g))
                m.bias.copy_(3.0 * torch.where(torch.arange(m.bias.numel()) % 3 == 2, -1.0, 1.0) * m.weight.abs())
    model = model.to(DEV).train()
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    N, shape = 3, (16, 24, 16)
    x = torch.rand((N, 1) + shape, generator=g).to(DEV)
    gout = ((0.5 + torch.rand((N, 1) + shape, generator=g)) / x.numel()).to(DEV)
    ref = _run(model, x, gout, False, True)
    model.load_state_dict(sd0)
    got = _run(model, x, gout, True, True)
    assert _rel(got[0], ref[0]) <= 2e-5
    worst = {k: _rel(got[1][k], ref[1][k]) for k in ref[1]}
    assert max(worst.values()) <= 1e-4, {k: v for k, v in worst.items() if v > 1e-4}
    assert _rel(got[2], ref[2]) <= 1e-4
    for k, v in ref[3].items():
        assert _rel(got[3][k].double(), v.double()) <= 1e-5, k


def test_backward_uses_the_plan_of_its_own_forward(monkeypatch):
    """The memory decisions of a forward call (what is written, what is kept, how an upsampled-input stage is sliced) live on
    that call's tape: another forward in between -- here one with every threshold changed and a different batch -- must not
    change how the pending backward slices and what it computes."""
    import models
    from dram_amd import engine
    torch.manual_seed(11)
    model = models.DC3D(**SLIM)
    model.init(models.HeNorm(mode="fan_in"))
    model = model.to(DEV).train()
    g = torch.Generator().manual_seed(12)
    x = torch.rand((3, 1, 16, 16, 16), generator=g).to(DEV)
    gout = ((0.5 + torch.rand((3, 1, 16, 16, 16), generator=g)) / x.numel()).to(DEV)
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    monkeypatch.setattr(engine, "MEMORY_MODE", "manual")

    def step(interleave):
        model.load_state_dict(sd0)
        for p in model.parameters():
            p.grad = None
        monkeypatch.setattr(engine, "MATERIALISE_BELOW", 0.0)
        monkeypatch.setattr(engine, "KEEP_UPSAMPLED_BELOW", 0.0)
        monkeypatch.setattr(engine, "SLICE_UPSAMPLED_ABOVE", 1e-12)        # slices of one sample
        out, _ = model(x)
        assert engine.LAST_PLAN.sliced_stages == 3
        if interleave:
            monkeypatch.setattr(engine, "MATERIALISE_BELOW", 1.0)
            monkeypatch.setattr(engine, "KEEP_UPSAMPLED_BELOW", 1.0)
            monkeypatch.setattr(engine, "SLICE_UPSAMPLED_ABOVE", 1.0)
            with torch.no_grad():
                model.eval()
                model(x[:2])
                model.train()
            assert engine.LAST_PLAN.sliced_stages == 0
        (out * gout).sum().backward()
        return {k: p.grad.clone() for k, p in model.named_parameters()}

    a, b = step(False), step(True)
    for k in a:
        assert torch.equal(a[k], b[k]), k
