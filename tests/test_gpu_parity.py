"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the
golden vectors generated from the reference.  Stated tolerance (BASELINE.json north_star):
1e-4 relative (max-abs error over max |reference|, and relative L2)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import dram_oracle as O
from dram_amd.configs import SLIM, ST_DRAM_REF_MODEL

pytestmark = pytest.mark.gpu
TOL = 1e-4
DEV = "cuda:0"


def rel_err(got, ref):
    got = got.detach().double().cpu() if isinstance(got, torch.Tensor) else torch.as_tensor(np.asarray(got)).double()
    ref = ref.detach().double().cpu() if isinstance(ref, torch.Tensor) else torch.as_tensor(np.asarray(ref)).double()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    if ref.numel() == 0:
        return 0.0, 0.0
    scale = max(ref.abs().max().item(), 1e-30)
    return (got - ref).abs().max().item() / scale, ((got - ref).norm() / max(ref.norm().item(), 1e-30)).item()


def check(got, ref, what, tol=TOL):
    mx, l2 = rel_err(got, ref)
    assert mx <= tol and l2 <= tol, f"{what}: max-rel {mx:.3e}, rel-L2 {l2:.3e} > {tol}"


def g(seed):
    return torch.Generator().manual_seed(seed)


def dev(t):
    return t.to(DEV)


# ------------------------------------------------------------------ conv 3x3x3
CONV_CASES = [
    # N, Cin, Cout, D, H, W, bias
    (2, 3, 4, 6, 10, 12, False),      # W < 12 -> 8-wide boxes... (12 -> 16-wide)
    (1, 5, 7, 7, 9, 11, True),        # odd everything, bias
    (2, 8, 16, 5, 6, 7, False),       # W = 7 -> 8-wide boxes
    (1, 1, 32, 9, 17, 33, False),     # first layer shape class (Cin = 1), 32-wide boxes
    (1, 32, 64, 6, 10, 40, False),    # Cout = 64 (two channel tiles), partial x boxes
    (2, 20, 70, 8, 8, 16, False),     # Cout not a multiple of 64, Cin not a multiple of 4/16
    (1, 64, 64, 4, 36, 36, False),
    (1, 130, 40, 3, 5, 20, True),     # Cin > 128
    (2, 1, 40, 5, 9, 35, False),      # Cin = 1 with two 32-channel output tiles (first-layer wgrad kernel)
    (3, 1, 8, 4, 4, 6, True),         # Cin = 1, tiny
    (1, 40, 72, 3, 5, 32, False),     # W a multiple of the box width: 16-byte staging path of wgrad (32-wide)
    (2, 24, 130, 4, 6, 16, False),    # ... 16-wide boxes, Cout > 128
    (1, 16, 16, 9, 6, 8, True),       # ... 8-wide boxes
    (1, 8, 8, 1, 6, 8, False),        # D = 1: no plane pairs -> the direct kernels
    (2, 12, 9, 2, 1, 4, True),        # a single plane pair, one row
    (2, 16, 40, 5, 20, 20, False),    # 20^2 planes: 10x10-position forward boxes, 4x8 wgrad boxes, odd D
    (1, 32, 64, 4, 10, 10, False),    # 10^2 planes (deepest level of an 80^3 chunk)
    # shapes the Winograd-(z,y) kernel serves (Cout % 64 == 0, W covered by 32-wide boxes):
    (1, 16, 64, 6, 8, 32, True),      # forward only (backward-data has 16 output channels), bias
    (1, 64, 128, 11, 8, 32, False),   # forward and backward-data; odd D (half-empty last plane pair)
    (1, 64, 64, 12, 11, 32, False),   # H = 11: ragged last y pair
    (2, 12, 64, 4, 12, 64, False),    # channel tail inside a chunk (12 = 3 chunks), two boxes along x, two samples
    (1, 64, 64, 3, 9, 70, True),      # W = 70: partial third box (padding 1.37 > 1.2 -> z-only kernel; the boundary of the rule)
    (1, 64, 192, 2, 4, 58, False),    # W = 58 (padding 1.10): partial second box through the (z,y) kernel, three channel tiles
    # shapes the 16-wide boxes (16 x 4 x 4) of the (z,y) kernel serve -- the pyramid of the reference's 80^3 chunks and the 16^3 level:
    (1, 64, 64, 8, 8, 16, False),     # one box per row, two z pairs per box, two boxes along z and y
    (2, 16, 128, 8, 12, 40, True),    # W = 40: ragged third box along x; bias; forward only on (z,y) (backward-data has 16 output channels)
    (1, 64, 64, 20, 20, 20, False),   # 20^3 (level 2 of an 80^3 chunk): ragged x box, five exact boxes along y and z
    (1, 72, 64, 24, 24, 24, False),   # W = 24
    (1, 64, 64, 6, 10, 16, False),    # D = 6, H = 10: half-empty last box along z and along y (padding 1.6, like the z-only kernel's own)
    (1, 64, 64, 7, 12, 16, False),    # D = 7: the last box holds ONE valid plane of its second z pair... (padding 1.14)
    (2, 12, 64, 4, 4, 16, False),     # a single box per sample, channel tail inside a chunk
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv3d_k3_fwd_bwd(case):
    from dram_amd import functional as HF
    N, Ci, Co, D, H, W, use_bias = case
    x = torch.randn(N, Ci, D, H, W, generator=g(1))
    w = torch.randn(Co, Ci, 3, 3, 3, generator=g(2)) / (Ci * 27) ** 0.5
    b = torch.randn(Co, generator=g(3)) if use_bias else None
    gy = torch.randn(N, Co, D, H, W, generator=g(4))
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True) if use_bias else None
    yr = O.conv3d(xr, wr, br, 1)
    yr.backward(gy)
    xg, wg = dev(x).requires_grad_(True), dev(w).requires_grad_(True)
    bg = dev(b).requires_grad_(True) if use_bias else None
    y = HF.conv3d_k3(xg, wg, bg)
    y.backward(dev(gy))
    check(y, yr, f"conv fwd {case}")
    check(xg.grad, xr.grad, f"conv dgrad {case}")
    check(wg.grad, wr.grad, f"conv wgrad {case}")
    if use_bias:
        check(bg.grad, br.grad, f"conv dbias {case}")


@pytest.mark.parametrize("shape", [((2, 6, 4, 6, 8), (2, 5, 4, 6, 8)),       # same size: identity crop
                                   ((2, 6, 6, 10, 12), (2, 4, 7, 11, 13)),   # crop offsets (1,1,1) (ceil)
                                   ((1, 3, 5, 6, 20), (1, 9, 8, 9, 24)),
                                   ((1, 32, 4, 4, 16), (1, 16, 6, 7, 19)),    # vector wgrad path, crop offset (1,2,2), tile-aligned C1
                                   ((2, 16, 3, 4, 32), (2, 5, 3, 4, 32))])    # vector wgrad path for Cout > 64 tiles (16-channel tiles)
def test_conv3d_k3_virtual_concat(shape):
    """conv(crop_concat_5d(up, skip)) without materialising the concatenation."""
    from dram_amd import functional as HF
    s1, s2 = shape
    up = torch.randn(*s1, generator=g(5))
    skip = torch.randn(*s2, generator=g(6))
    Ci, Co = s1[1] + s2[1], (72 if s1[-1] == 32 else 10)   # 72 output channels -> 128x16 tiles
    w = torch.randn(Co, Ci, 3, 3, 3, generator=g(7)) / (Ci * 27) ** 0.5
    gy = torch.randn(s1[0], Co, *s1[2:], generator=g(8))
    ur, sr, wr = up.clone().requires_grad_(True), skip.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = O.conv3d(O.crop_concat_5d(ur, sr), wr, None, 1)
    yr.backward(gy)
    ug, sg, wg = dev(up).requires_grad_(True), dev(skip).requires_grad_(True), dev(w).requires_grad_(True)
    y = HF.conv3d_k3(ug, wg, None, skip=sg)
    y.backward(dev(gy))
    check(y, yr, "cat conv fwd")
    check(ug.grad, ur.grad, "cat conv d(up)")
    check(sg.grad, sr.grad, "cat conv d(skip)")
    check(wg.grad, wr.grad, "cat conv wgrad")
    # and the stand-alone crop_concat_5d kernel
    out = HF.crop_concat(dev(up), dev(skip))
    assert torch.equal(out.cpu(), O.crop_concat_5d(up, skip))


def test_conv3d_k3_wzy_kernel_is_selected():
    """The shapes above that are meant for the Winograd-(z,y) kernel really launch it: the library is asked what it
    chooses (dram_conv3d_k3_fwd_choice, the same fwd_choice the launch goes through; there is no host-side copy of the
    rule), and its launch counters are read around a real call."""
    from dram_amd import functional as HF
    assert HF.conv_fwd_kernel_name((11, 8, 32), 128, 64) == "conv3d_k3_fwd_wzy_kernel"
    assert HF.conv_fwd_kernel_name((11, 8, 32), 64, 128, fused=True) == "conv3d_k3_fwd_wzy_kernel"
    assert HF.conv_fwd_kernel_name((12, 11, 32), 64, 64) == "conv3d_k3_fwd_wzy_kernel"
    assert "wz_kernel" in HF.conv_fwd_kernel_name((5, 7, 32), 128, 64)          # too much padding (1.37)
    assert HF.conv_fwd_kernel_name((2, 4, 56), 192, 64) == "conv3d_k3_fwd_wzy_kernel"
    assert "wz_kernel" in HF.conv_fwd_kernel_name((2, 4, 58), 192, 64)         # rows are fetched as aligned 16-byte pieces: W % 4
    assert "wz_kernel" in HF.conv_fwd_kernel_name((3, 9, 70), 64, 64)          # too much padding
    assert "wz_kernel" in HF.conv_fwd_kernel_name((6, 8, 32), 16, 64)          # 16 output channels
    assert HF.conv_fwd_kernel_name((128, 128, 128), 64, 192, fused=True) == "conv3d_k3_fwd_wzy_kernel"
    # ... and its 16-wide boxes where 32-wide ones pad more (the widths 80 / 40 / 20 / 24 / 16)
    for dhw, co, ci in (((8, 8, 16), 64, 64), ((8, 12, 40), 128, 16), ((20, 20, 20), 64, 64), ((24, 24, 24), 64, 72),
                        ((7, 12, 16), 64, 64), ((4, 4, 16), 64, 12), ((80, 80, 80), 64, 192)):
        assert HF.conv_fwd_kernel_name(dhw, co, ci) == "conv3d_k3_fwd_wzy16_kernel", dhw
    # 16 x 12 x 8 for 16 x 10 x 6: padding 1.6 -- as much as the z-only kernel's own best box (16 x 8 positions) pads: (z,y)
    assert HF.conv_fwd_kernel_name((6, 10, 16), 64, 64) == "conv3d_k3_fwd_wzy16_kernel"
    assert "wz_kernel" in HF.conv_fwd_kernel_name((6, 8, 16), 64, 64)           # 1.33 against an exact z-only box
    before = HF.conv_launch_counts()
    x = dev(torch.randn(1, 64, 8, 8, 16, generator=g(33))).requires_grad_(True)
    w = dev(torch.randn(64, 64, 3, 3, 3, generator=g(34)) * 0.05).requires_grad_(True)
    HF.conv3d_k3(x, w).sum().backward()
    torch.cuda.synchronize()
    delta = [a - b for a, b in zip(HF.conv_launch_counts(), before)]
    assert delta[HF.K3_FWD_WZY] == 2 and delta[HF.K3_WGRAD_WZY] == 1 and sum(delta) == 3, delta
    # what a launch really does: forward (64 -> 128) and backward-data (128 -> 64) on the (z,y) kernel, backward-weights on
    # the transposed Winograd kernel
    before = HF.conv_launch_counts()
    x = dev(torch.randn(1, 64, 4, 8, 32, generator=g(31))).requires_grad_(True)
    w = dev(torch.randn(128, 64, 3, 3, 3, generator=g(32)) * 0.05).requires_grad_(True)
    HF.conv3d_k3(x, w).sum().backward()
    torch.cuda.synchronize()
    after = HF.conv_launch_counts()
    delta = [a - b for a, b in zip(after, before)]
    assert delta[HF.K3_FWD_WZY] == 2, delta
    assert delta[HF.K3_WGRAD_WZY] == 1 and sum(delta) == 3, delta              # (the (z,y) backward-weights kernel, on its own)


def test_conv3d_k3_wzy_needs_aligned_rows():
    """The (z,y) forward kernel fetches input rows as aligned 16-byte pieces (LDS-DMA): a source that starts 4 bytes off a
    16-byte boundary must be served by the z-only kernel -- and give the same values."""
    from dram_amd import functional as HF
    x = torch.randn(1, 64, 4, 8, 32, generator=g(41))
    w = torch.randn(64, 64, 3, 3, 3, generator=g(42)) * 0.05
    ref = O.conv3d(x, w, None, 1)
    store = torch.empty(x.numel() + 4, device=DEV)
    outs = []
    for shift, wzy_launches in ((0, 1), (1, 0)):
        xv = store[shift:shift + x.numel()].view_as(x)
        xv.copy_(x)
        assert xv.data_ptr() % 16 == 4 * shift
        before = HF.conv_launch_counts()
        y = HF.conv3d_k3(xv, dev(w))
        torch.cuda.synchronize()
        after = HF.conv_launch_counts()
        assert after[HF.K3_FWD_WZY] - before[HF.K3_FWD_WZY] == wzy_launches, (shift, before, after)
        check(y, ref, f"conv fwd, source {4 * shift} bytes off alignment")
        outs.append(y)


def test_conv3d_k3_wzy_concat_and_split():
    """Winograd-(z,y) kernel with a virtual concat source (forward) and a destination split over two tensors at a
    32-channel boundary (backward-data of the same conv)."""
    from dram_amd import functional as HF
    up = torch.randn(1, 32, 4, 8, 32, generator=g(15))
    skip = torch.randn(1, 32, 6, 11, 40, generator=g(16))      # (crop window at x offset 4: rows stay 16-byte aligned)
    Ci, Co = 64, 64
    w = torch.randn(Co, Ci, 3, 3, 3, generator=g(17)) / (Ci * 27) ** 0.5
    gy = torch.randn(1, Co, 4, 8, 32, generator=g(18))
    ur, sr, wr = up.clone().requires_grad_(True), skip.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = O.conv3d(O.crop_concat_5d(ur, sr), wr, None, 1)
    yr.backward(gy)
    ug, sg, wg = dev(up).requires_grad_(True), dev(skip).requires_grad_(True), dev(w).requires_grad_(True)
    before = HF.conv_launch_counts()
    y = HF.conv3d_k3(ug, wg, None, skip=sg)
    y.backward(dev(gy))
    torch.cuda.synchronize()
    assert HF.conv_launch_counts()[HF.K3_FWD_WZY] - before[HF.K3_FWD_WZY] == 2      # forward and backward-data
    check(y, yr, "wzy cat conv fwd")
    check(ug.grad, ur.grad, "wzy cat conv d(up)")
    check(sg.grad, sr.grad, "wzy cat conv d(skip)")
    check(wg.grad, wr.grad, "wzy cat conv wgrad")


@pytest.mark.parametrize("case", [
    # N, C1, C2, Cout, D, H, W, lazy, stats
    (1, 12, 0, 64, 11, 8, 32, True, True),       # odd D, channel tail
    (1, 16, 0, 64, 12, 11, 60, True, True),      # ragged y pair and ragged x box
    (2, 10, 0, 128, 4, 12, 64, True, False),
    (1, 8, 8, 64, 8, 8, 32, True, True),         # virtual concat, both sources lazy
    (2, 6, 10, 64, 4, 12, 32, True, True),       # ... with the source boundary inside a 4-channel chunk (raw rows: per-channel source)
    (1, 64, 0, 192, 8, 16, 32, False, True),     # statistics of a plain source, three channel tiles
    (3, 16, 0, 64, 2, 4, 32, True, True),        # one plane pair, one box per sample: items of three samples per block
    # 16-wide boxes
    (2, 12, 0, 64, 8, 8, 16, True, True),        # channel tail; two boxes along z and y
    (1, 16, 0, 128, 20, 20, 20, True, True),     # 20^3: ragged x box; the slot count covers the z-only boxing too (zero-filled tail)
    (1, 8, 8, 64, 8, 12, 40, True, True),        # virtual concat (crop window at x offset 4), W = 40
    (2, 6, 10, 64, 7, 8, 16, True, True),        # odd D (a box with one valid plane in its second z pair), source boundary inside a chunk
    (3, 16, 0, 64, 4, 4, 16, True, False),       # one box per sample
    (1, 64, 0, 64, 24, 24, 24, False, True),     # W = 24, statistics of a plain source
])
def test_conv3d_k3_wzy_fused(case):
    """Fused forward (normalise + ReLU on load, statistics epilogue) of the Winograd-(z,y) kernel through the C ABI:
    against an fp64 reference and against the z-only kernel (DRAM_CONV_NO_WZY is read per call)."""
    from dram_amd import functional as HF
    from dram_amd import _lib
    N, C1, C2, Co, D, H, W, lazy, stats = case
    Ci = C1 + C2
    d = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    p = lambda t: None if t is None else t.data_ptr()
    x = dev(torch.randn(N, C1, D, H, W, generator=g(21)))
    x2 = dev(torch.randn(N, C2, D + 2, H + 3, W + 4, generator=g(22))) if C2 else None
    crop = (1, 2, 4) if C2 else (0, 0, 0)
    w = dev(torch.randn(Co, Ci, 3, 3, 3, generator=g(23)) / (Ci * 27) ** 0.5)
    coef = dev(torch.rand(N * C1 * 2, generator=g(24)) + 0.5) if lazy else None
    coef2 = dev(torch.rand(N * C2 * 2, generator=g(25)) - 0.2) if (lazy and C2) else None
    wt = HF._pack(w, 0)

    def run(wzy):
        if wzy:
            os.environ.pop("DRAM_CONV_NO_WZY", None)
        else:
            os.environ["DRAM_CONV_NO_WZY"] = "1"
        try:
            y = torch.full((N, Co, D, H, W), float("nan"), device=d)
            nparts = _lib.lib.dram_conv3d_k3_stats_parts(Ci, Co, D, H, W) if stats else 0
            parts = torch.full((N * Co * nparts * 3,), float("nan"), device=d) if stats else None
            d2 = (0, 0, 0) if x2 is None else tuple(x2.shape[2:])
            before = HF.conv_launch_counts()
            _lib.call("dram_conv3d_k3_fwd_fused", p(x), C1, p(coef), 1, p(x2), C2, p(coef2), 1, d2[0], d2[1], d2[2], crop[0], crop[1],
                      crop[2], p(wt), None, p(y), p(parts), nparts, N, Co, D, H, W, st)
            torch.cuda.synchronize()
            assert HF.conv_launch_counts()[HF.K3_FWD_WZY] - before[HF.K3_FWD_WZY] == (1 if wzy else 0), case
        finally:
            os.environ.pop("DRAM_CONV_NO_WZY", None)
        return y, parts, nparts

    def act(t, cf):
        if cf is None:
            return t.double()
        c = cf.view(t.shape[0], t.shape[1], 2).double()
        return torch.relu(t.double() * c[:, :, 0, None, None, None] + c[:, :, 1, None, None, None])

    xin = act(x, coef)
    if C2:
        xin = torch.cat([xin, act(x2[:, :, crop[0]:crop[0] + D, crop[1]:crop[1] + H, crop[2]:crop[2] + W], coef2)], 1)
    ref = torch.nn.functional.conv3d(xin.cpu(), w.double().cpu(), None, padding=1)
    ya, pa, na = run(True)
    yb, pb, nb = run(False)
    check(ya, ref, f"wzy fused fwd {case}")
    check(yb, ref, f"wz fused fwd {case}")
    if stats:
        q = pa.view(N * Co, na, 3).double()
        cnt = q[:, :, 2].sum(1)
        mean = (q[:, :, 0] * q[:, :, 2]).sum(1) / cnt
        m2 = (q[:, :, 1] + q[:, :, 2] * (q[:, :, 0] - mean[:, None]) ** 2).sum(1)
        r = ya.double().view(N * Co, -1)
        assert bool((cnt == D * H * W).all()), case
        assert (mean - r.mean(1)).abs().max().item() < 1e-5, case
        assert ((m2 - ((r - r.mean(1, keepdim=True)) ** 2).sum(1)).abs() / m2).max().item() < 1e-4, case


def test_conv3d_k3_fused_forward_source_fallback_keeps_its_partials():
    """A (z,y)-shaped fused forward whose SOURCE the (z,y) kernel refuses -- a cropped skip whose window starts at x % 4 != 0,
    a base pointer 4 bytes off a 16-byte boundary -- runs the z-only kernel on the same 32x4x2 boxes, so the statistics buffer
    sized by dram_conv3d_k3_stats_parts(shape) is exactly what the launch fills (round-3 advisor finding: at H, W = 32, 56 the
    z-only kernel's own choice is 8x16 boxes with another partial count, and the launch raised)."""
    from dram_amd import functional as HF
    from dram_amd import _lib
    N, C1, C2, Co, D, H, W = 2, 8, 8, 64, 4, 32, 56
    Ci = C1 + C2
    st = torch.cuda.current_stream().cuda_stream
    p = lambda t: None if t is None else t.data_ptr()
    w = dev(torch.randn(Co, Ci, 3, 3, 3, generator=g(51)) / (Ci * 27) ** 0.5)
    wt = HF._pack(w, 0)
    coef = dev(torch.rand(N * C1 * 2, generator=g(52)) + 0.5)
    coef2 = dev(torch.rand(N * C2 * 2, generator=g(53)) - 0.2)
    nparts = _lib.lib.dram_conv3d_k3_stats_parts(Ci, Co, D, H, W)
    assert nparts == 2 * 8 * 2 * 4
    x1 = torch.randn(N, C1, D, H, W, generator=g(54))
    for what, x2shape, shift in (("crop window at x offset 3", (N, C2, D + 2, H + 3, W + 6), 0),
                                 ("first source 4 bytes off alignment", (N, C2, D, H, W), 1),
                                 ("aligned sources (control)", (N, C2, D + 2, H + 2, W + 8), 0)):
        x2 = dev(torch.randn(*x2shape, generator=g(55)))
        store = torch.empty(x1.numel() + 4, device=DEV)
        xv = store[shift:shift + x1.numel()].view_as(x1)
        xv.copy_(x1)
        assert xv.data_ptr() % 16 == 4 * shift
        crop = tuple(int(np.ceil((b - a) / 2)) for a, b in zip((D, H, W), x2shape[2:]))
        control = what.endswith("(control)")
        assert HF.conv_fwd_kernel_name((D, H, W), Co, Ci, fused=True, src=(xv, x2, crop[2])) == \
            ("conv3d_k3_fwd_wzy_kernel" if control else "conv3d_k3_fwd_wz_kernel<32, 4, 2, true>"), what
        y = torch.full((N, Co, D, H, W), float("nan"), device=DEV)
        parts = torch.full((N * Co * nparts * 3,), float("nan"), device=DEV)
        before = HF.conv_launch_counts()
        _lib.call("dram_conv3d_k3_fwd_fused", p(xv), C1, p(coef), 1, p(x2), C2, p(coef2), 1, *x2shape[2:], *crop,
                  p(wt), None, p(y), p(parts), nparts, N, Co, D, H, W, st)
        torch.cuda.synchronize()
        delta = [a - b for a, b in zip(HF.conv_launch_counts(), before)]
        assert (delta[HF.K3_FWD_WZY], delta[HF.K3_FWD_WZ]) == ((1, 0) if control else (0, 1)), (what, delta)

        def act(t, cf):
            c = cf.view(t.shape[0], t.shape[1], 2).double()
            return torch.relu(t.double() * c[:, :, 0, None, None, None] + c[:, :, 1, None, None, None])
        win = x2[:, :, crop[0]:crop[0] + D, crop[1]:crop[1] + H, crop[2]:crop[2] + W]
        ref = torch.nn.functional.conv3d(torch.cat([act(xv, coef), act(win, coef2)], 1).cpu(), w.double().cpu(), None, padding=1)
        check(y, ref, f"fused forward, {what}")
        q = parts.view(N * Co, nparts, 3).double()
        cnt = q[:, :, 2].sum(1)
        assert bool((cnt == D * H * W).all()), what
        mean = (q[:, :, 0] * q[:, :, 2]).sum(1) / cnt
        assert (mean - y.double().view(N * Co, -1).mean(1)).abs().max().item() < 1e-5, what


WGRAD_WZY_CASES = [
    # N, C1, C2, Cout, (D, H, W), skip shape beyond (D, H, W), lazy, bytes off alignment of x1 / dy
    (3, 16, 16, 80, (4, 6, 16), (2, 3, 4), True, 0),      # review case: two lazy sources, concat boundary at 16, D = 4, W = 16, N = 3
                                                          # (80 output channels: the z-only kernel of the A/B needs its 128 x 16 tile for that boundary)
    (2, 32, 16, 72, (4, 4, 16), (2, 3, 3), True, 0),      # crop window at x offset 2 (ceil(3/2)), rows of 19 floats: unaligned 16-byte pieces
    (2, 32, 16, 72, (4, 4, 16), (2, 3, 3), False, 0),     # ... plain operands
    (2, 16, 0, 64, (6, 4, 32), (0, 0, 0), True, 4),       # x and dy 4 bytes off a 16-byte boundary, lazy
    (1, 48, 0, 130, (4, 8, 16), (0, 0, 0), False, 4),     # ... plain, three co tiles with a tail
    (2, 16, 16, 80, (4, 6, 40), (2, 3, 4), True, 0),      # W = 40 (the 40^3 level of an 80^3 chunk): ragged third box along x, lazy
    (1, 32, 0, 64, (6, 4, 72), (0, 0, 0), False, 0),      # W = 72: four full boxes + half a box, plain
    (1, 16, 0, 64, (4, 4, 28), (0, 0, 0), True, 0),       # W = 28: the second box holds 12 of its 16 columns
    # exactly ONE lazy source (the first conv of an up-block in the fused engine: upsampled part plain, skip lazy): one launch
    # per source, each with its own split, disjoint columns of the same slabs
    (2, 32, 16, 64, (4, 6, 32), (2, 3, 4), "x2", 0),
    (3, 16, 48, 80, (6, 4, 16), (0, 0, 0), "x2", 0),      # more lazy than plain tiles, two co tiles, same-size skip
    (2, 32, 32, 64, (4, 4, 16), (2, 2, 4), "x1", 0),      # ... the other way round
]


@pytest.mark.parametrize("case", WGRAD_WZY_CASES)
def test_conv3d_k3_wgrad_wzy_fused(case):
    """The Winograd-(z,y) backward-weights kernel ITSELF (launch counter DRAM_K3_WGRAD_WZY, on its own) against an fp64
    reference and against the z-only kernel (DRAM_WGRAD_NO_WZY=1, read per call): two lazy sources with DIFFERENT per-sample
    coefficients, a concat boundary at 16 channels, D = 4 (two boxes per z column: the shortest the raw-plane ring serves),
    W = 16 (one box per row), N = 3; a cropped second source whose window starts at x % 4 != 0 and whose rows are 19 floats (its
    16-byte LDS-DMA pieces are then only dword aligned in global memory -- the kernel drops border pieces by explicit
    out-of-range offsets, not by alignment, unlike the forward kernel's descriptor-range padding: wgrad_wzy.inc); base
    pointers 4 bytes off 16-byte alignment.  This pins that dword-aligned 16-byte LDS-DMA is something the kernel may rely on."""
    from dram_amd import functional as HF
    from dram_amd import _lib
    N, C1, C2, Co, (D, H, W), extra, lazy, off = case
    Ci = C1 + C2
    st = torch.cuda.current_stream().cuda_stream
    p = lambda t: None if t is None else t.data_ptr()

    def place(t, shift_bytes):          # a device copy whose base pointer is `shift_bytes` off a 16-byte boundary
        store = torch.empty(t.numel() + 4, device=DEV)
        v = store[shift_bytes // 4:shift_bytes // 4 + t.numel()].view_as(t)
        v.copy_(t)
        assert v.data_ptr() % 16 == shift_bytes
        return v
    x1 = place(torch.randn(N, C1, D, H, W, generator=g(61)), off)
    x2 = dev(torch.randn(N, C2, D + extra[0], H + extra[1], W + extra[2], generator=g(62))) if C2 else None
    crop = tuple(int(np.ceil(e / 2)) for e in extra) if C2 else (0, 0, 0)
    dy = place(torch.randn(N, Co, D, H, W, generator=g(63)), off)
    coef1 = dev(torch.rand(N * C1 * 2, generator=g(64)) + 0.25) if lazy in (True, "x1") else None   # per (sample, channel): all different
    coef2 = dev(torch.rand(N * C2 * 2, generator=g(65)) - 0.3) if (lazy in (True, "x2") and C2) else None
    mixed = lazy in ("x1", "x2")
    assert "wgrad_wzy" in HF.conv_wgrad_kernel_name(N, (D, H, W), Co, C1, C2, lazy=bool(lazy))

    def run(wzy):
        if wzy:
            os.environ.pop("DRAM_WGRAD_NO_WZY", None)
        else:
            os.environ["DRAM_WGRAD_NO_WZY"] = "1"
        try:
            dw = torch.full((Co, Ci, 3, 3, 3), float("nan"), device=DEV)
            ws = torch.empty(max(16, _lib.lib.dram_conv3d_k3_wgrad_ws_bytes(N, Ci, Co, D, H, W)), dtype=torch.uint8, device=DEV)
            d2 = (0, 0, 0) if x2 is None else tuple(x2.shape[2:])
            before = HF.conv_launch_counts()
            _lib.call("dram_conv3d_k3_wgrad_fused", p(x1), C1, p(coef1), 1, p(x2), C2, p(coef2), 1, *d2, *crop, p(dy), p(dw),
                      p(ws), ws.numel(), N, Co, D, H, W, st)
            torch.cuda.synchronize()
            delta = [a - b for a, b in zip(HF.conv_launch_counts(), before)]
            assert delta[HF.K3_WGRAD_WZY] == ((2 if mixed else 1) if wzy else 0) and sum(delta) == (2 if mixed and wzy else 1), (case, wzy, delta)
        finally:
            os.environ.pop("DRAM_WGRAD_NO_WZY", None)
        return dw

    def act(t, cf):
        if cf is None:
            return t.double()
        c = cf.view(t.shape[0], t.shape[1], 2).double()
        return torch.relu(t.double() * c[:, :, 0, None, None, None] + c[:, :, 1, None, None, None])
    xin = act(x1, coef1)
    if C2:
        xin = torch.cat([xin, act(x2[:, :, crop[0]:crop[0] + D, crop[1]:crop[1] + H, crop[2]:crop[2] + W], coef2)], 1)
    wr = torch.zeros(Co, Ci, 3, 3, 3, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv3d(xin.cpu(), wr, None, padding=1).backward(dy.double().cpu())
    a, b = run(True), run(False)
    check(a, wr.grad, f"wgrad (z,y) {case}", tol=2e-5)
    check(b, wr.grad, f"wgrad z-only {case}", tol=2e-5)


@pytest.mark.parametrize("flags", [("--seed", "11"), ("--seed", "12", "--wzy")])
def test_randomised_conv_sweep(flags):
    """scripts/fuzz_conv.py: 40 random small shapes per run -- fused forward (lazy sources, cropped skip, statistics with the
    total-count check), backward-data with a split destination (nothing written outside the crop window), fused
    backward-weights (no / one / two lazy sources: the per-source launches) -- against an fp64 convolution, whichever kernel the
    library picks; the second run biases the shapes towards the (z,y) kernels (16-wide boxes, ragged boxes).  390 such cases
    were run once by hand (gpurun_out/fuzz*.log); this keeps 80 in the suite."""
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "fuzz_conv.py"), "--cases", "40", *flags], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "cases agree" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


def test_packed_filter_cache_follows_the_weights():
    """Inference calls (functional.cached_packs(): the fused engine's no-gradient branch) reuse a filter's packed form while the
    tensor is unchanged; an in-place update (optimiser step, load_state_dict) or a call outside the context must not see a
    stale copy."""
    from dram_amd import functional as HF
    x = dev(torch.randn(1, 8, 4, 6, 16, generator=g(81)))
    w = dev(torch.randn(16, 8, 3, 3, 3, generator=g(82)) * 0.1)
    with torch.no_grad(), HF.cached_packs():
        y0 = HF.conv3d_k3(x, w)
        packed = HF._pack_cached(w)[0][2]
        y1 = HF.conv3d_k3(x, w)
        assert HF._pack_cached(w)[0][2] is packed and torch.equal(y0, y1)         # second call: no repack
        w.mul_(2.0)                                                               # in-place update: version bump
        y2 = HF.conv3d_k3(x, w)
        assert HF._pack_cached(w)[0][2] is not packed
    check(y2, 2.0 * y0, "conv after an in-place weight update", tol=1e-6)
    wg = w.clone().requires_grad_(True)
    HF.conv3d_k3(x, wg).sum().backward()                                          # a training call caches nothing
    assert HF._pack_cached(wg) is None
    HF.conv3d_k3(x, w.requires_grad_(True))                                       # ... and drops what evaluation left
    assert HF._pack_cached(w) is None
    n = len(HF._PACK_CACHE)
    with torch.no_grad(), HF.cached_packs():
        tmp = dev(torch.randn(16, 8, 3, 3, 3, generator=g(83)))
        HF.conv3d_k3(x, tmp)
        assert len(HF._PACK_CACHE) == n + 1
        del tmp                                                                   # the entry dies with the tensor
    assert len(HF._PACK_CACHE) == n


def test_direct_conv_kernels_still_agree():
    """DRAM_CONV_DIRECT=1 routes every layer to the direct (27-tap) kernels that the Winograd ones replaced by default;
    they stay in the library as the A/B baseline and are kept verified by re-running the conv cases under that switch
    (a child process: the switch is read once per process)."""
    import subprocess
    import sys
    env = dict(os.environ, DRAM_CONV_DIRECT="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-k",
                        "test_conv3d_k3_fwd_bwd or test_conv3d_k3_virtual_concat"], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_conv3d_k3_linearity_and_shift_large():
    """Size-independent properties at a layer-sized shape (config-2 class: 64->64 channels):
    linearity in x and agreement with the oracle on an interior sub-volume."""
    from dram_amd import functional as HF
    N, C, S = 2, 64, 64
    x1 = torch.randn(N, C, S, S, S, generator=g(9), device="cpu")
    x2 = torch.randn(N, C, S, S, S, generator=g(10), device="cpu")
    w = torch.randn(C, C, 3, 3, 3, generator=g(11)) / (C * 27) ** 0.5
    xa, xb, wg = dev(x1), dev(x2), dev(w)
    with torch.no_grad():
        ya, yb = HF.conv3d_k3(xa, wg), HF.conv3d_k3(xb, wg)
        yab = HF.conv3d_k3(2.0 * xa - 0.5 * xb, wg)
    check(yab, 2.0 * ya - 0.5 * yb, "conv linearity", tol=2e-5)
    # oracle on a sub-volume: output [8:24]^3 depends on input [7:25]^3 only
    sub = x1[:, :, 7:25, 7:25, 7:25].contiguous()
    ref = O.conv3d(sub, w, None, 0)    # valid conv of the haloed crop
    check(ya[:, :, 8:24, 8:24, 8:24], ref, "conv fwd interior vs oracle")
    # border voxels see zero padding
    ref0 = O.conv3d(x1[:1, :, :10, :10, :10].contiguous(), w, None, 1)[:, :, :8, :8, :8]
    check(ya[:1, :, :8, :8, :8], ref0, "conv fwd corner vs oracle")


@pytest.mark.parametrize("norm", ["ln", "in", "bn"])
def test_config2_layer_full_size(norm):
    """BASELINE config 2 at its full size: x = [4,64,128^3] ~ U[0,1) (2.1 GB), Conv3d(64,64,3,p=1) + norm + ReLU,
    forward only -- reference layer us_modules.2.conv_blocks.1 (parts.py:140-146).  The conv is checked against
    the oracle on an interior and a corner sub-volume (the full CPU conv takes minutes), the norm + ReLU of the
    whole 2.1 GB conv output against the oracle's torch-CPU norm; tolerance 1e-4 (max-abs and rel-L2)."""
    from dram_amd import functional as HF
    import parts
    N, C, S = 4, 64, 128
    x = torch.rand(N, C, S, S, S, generator=g(2))
    w = torch.randn(C, C, 3, 3, 3, generator=g(3)) * (2.0 / (C * 27)) ** 0.5       # kaiming normal, fan_in
    xg, wg = dev(x), dev(w)
    with torch.no_grad():
        yc = HF.conv3d_k3(xg, wg)
        sub = x[:1, :, 39:73, 59:85, 7:41].contiguous()
        check(yc[:1, :, 40:72, 60:84, 8:40], O.conv3d(sub, w, None, 0), "config2 conv interior")
        ref0 = O.conv3d(x[3:, :, S - 18:, :18, S - 18:].contiguous(), w, None, 1)[:, :, 2:, :16, 2:]
        check(yc[3:, :, S - 16:, :16, S - 16:], ref0, "config2 conv corner")
        m = parts.normal_wrapper(norm, C).to(DEV).train()
        gmm, bta = torch.rand(C, generator=g(4)) + 0.5, torch.randn(C, generator=g(5)) * 0.2
        m.weight.copy_(gmm); m.bias.copy_(bta)
        out = m(yc, relu=True)
        ycc = yc.cpu()
        del yc
        if norm == "bn":
            ref = torch.nn.functional.batch_norm(ycc, None, None, gmm, bta, True, 0.1, O.EPS)
        else:
            ref = torch.nn.functional.group_norm(ycc, 1 if norm == "ln" else C, gmm, bta, O.EPS)
        ref = torch.relu_(ref)
        got = out.cpu()
        err = (got - ref).abs().max().item() / ref.abs().max().item()
        rel_l2 = ((got - ref).double().norm() / ref.double().norm()).item()
        assert err <= 1e-4 and rel_l2 <= 1e-4, (norm, err, rel_l2)


# ------------------------------------------------------------------ norms
NORM_SHAPES = [(2, 6, 6, 10, 12), (3, 4, 7, 9, 11), (2, 8, 24, 24, 24)]   # 693-voxel rows exercise the scalar path


@pytest.mark.parametrize("shape", NORM_SHAPES)
@pytest.mark.parametrize("relu", [False, True])
@pytest.mark.parametrize("affine", [True, False])
def test_batchnorm_train_eval(shape, relu, affine):
    from dram_amd.modules import HipBatchNorm3d
    C = shape[1]
    x = torch.randn(*shape, generator=g(12)) * 1.7 + 0.6
    gy = torch.randn(*shape, generator=g(13))
    ref = torch.nn.BatchNorm3d(C, affine=affine)
    mod = HipBatchNorm3d(C, affine=affine)
    if affine:
        ref.weight.data = torch.rand(C, generator=g(14)) + 0.5
        ref.bias.data = torch.randn(C, generator=g(15)) * 0.3
        mod.load_state_dict(ref.state_dict())
    mod = mod.to(DEV)
    xr = x.clone().requires_grad_(True)
    yr = ref(xr)
    yr = F.relu(yr) if relu else yr
    yr.backward(gy)
    xg = dev(x).requires_grad_(True)
    y = mod(xg, relu=relu)
    y.backward(dev(gy))
    check(y, yr, "bn train fwd")
    check(xg.grad, xr.grad, "bn train dx")
    if affine:
        check(mod.weight.grad, ref.weight.grad, "bn dgamma")
        check(mod.bias.grad, ref.bias.grad, "bn dbeta")
    check(mod.running_mean, ref.running_mean, "bn running_mean")
    check(mod.running_var, ref.running_var, "bn running_var")
    assert int(mod.num_batches_tracked) == 1
    # eval mode with the updated buffers, including backward through running statistics
    ref.eval(), mod.eval()
    xr2, xg2 = x.clone().requires_grad_(True), dev(x).requires_grad_(True)
    yr2 = ref(xr2)
    yr2 = F.relu(yr2) if relu else yr2
    yr2.backward(gy)
    y2 = mod(xg2, relu=relu)
    y2.backward(dev(gy))
    check(y2, yr2, "bn eval fwd")
    check(xg2.grad, xr2.grad, "bn eval dx")


@pytest.mark.parametrize("shape", NORM_SHAPES)
@pytest.mark.parametrize("groups", ["one", "all", "two"])
@pytest.mark.parametrize("relu", [False, True])
def test_groupnorm(shape, groups, relu):
    from dram_amd.modules import HipGroupNorm
    C = shape[1]
    G = {"one": 1, "all": C, "two": 2}[groups]
    x = torch.randn(*shape, generator=g(16)) * 2.0 - 0.4
    gy = torch.randn(*shape, generator=g(17))
    ref = torch.nn.GroupNorm(G, C)
    ref.weight.data = torch.rand(C, generator=g(18)) + 0.5
    ref.bias.data = torch.randn(C, generator=g(19)) * 0.3
    mod = HipGroupNorm(G, C)
    mod.load_state_dict(ref.state_dict())
    mod = mod.to(DEV)
    xr = x.clone().requires_grad_(True)
    yr = ref(xr)
    yr = F.relu(yr) if relu else yr
    yr.backward(gy)
    xg = dev(x).requires_grad_(True)
    y = mod(xg, relu=relu)
    y.backward(dev(gy))
    check(y, yr, "gn fwd")
    check(xg.grad, xr.grad, "gn dx")
    check(mod.weight.grad, ref.weight.grad, "gn dgamma")
    check(mod.bias.grad, ref.bias.grad, "gn dbeta")


def test_batchnorm_statistics_with_large_mean():
    """|mean| >> std: a naive E[x^2]-E[x]^2 in fp32 fails this; the Chan combination must not."""
    from dram_amd.modules import HipBatchNorm3d
    x = torch.randn(4, 3, 40, 40, 40, generator=g(20)) * 0.05 + torch.tensor([100.0, -50.0, 10.0]).view(1, 3, 1, 1, 1)
    mod = HipBatchNorm3d(3).to(DEV)
    y = mod(dev(x))
    xd = x.double()
    mean = xd.mean(dim=(0, 2, 3, 4), keepdim=True)
    var = xd.var(dim=(0, 2, 3, 4), unbiased=False, keepdim=True)
    ref = (xd - mean) / torch.sqrt(var + 1e-5)
    check(y, ref.float(), "bn large-mean", tol=2e-3)   # the input itself only carries ~1e-7*100/0.05 = 2e-4 relative
    check(mod.running_var, (0.9 + 0.1 * xd.var(dim=(0, 2, 3, 4), unbiased=True)).float(), "bn large-mean running_var", tol=1e-3)


# ------------------------------------------------------------------ pool / resize / head
@pytest.mark.parametrize("shape", [(2, 3, 6, 10, 12), (1, 4, 7, 9, 11), (2, 2, 16, 16, 32), (1, 2, 7, 9, 8), (1, 1, 2, 3, 4)])
def test_maxpool_with_ties(shape):
    from dram_amd import functional as HF
    x = torch.relu(torch.randn(*shape, generator=g(21)))     # ~50 % zeros: ties in most windows
    x[0, 0, :2, :2, :2] = 3.0                                 # an all-equal window
    xr = x.clone().requires_grad_(True)
    yr = O.max_pool3d_2(xr)
    gy = torch.randn(yr.shape, generator=g(22))
    yr.backward(gy)
    xg = dev(x).requires_grad_(True)
    y = HF.max_pool3d_2(xg)
    y.backward(dev(gy))
    assert torch.equal(y.cpu(), yr.detach())
    assert torch.equal(xg.grad.cpu(), xr.grad)                # same tie-breaking as ATen: bit-exact routing


@pytest.mark.parametrize("case", [((2, 3, 3, 5, 6), None, 2), ((1, 2, 4, 4, 4), None, (2, 2, 2)),
                                  ((1, 2, 5, 6, 7), (9, 13, 8), None), ((2, 1, 16, 16, 16), (21, 18, 20), None),
                                  ((1, 2, 1, 4, 5), (3, 8, 10), None), ((1, 1, 6, 6, 6), (6, 6, 6), None),
                                  ((1, 2, 8, 7, 9), (4, 5, 3), None), ((1, 2, 3, 4, 2), (14, 9, 11), None),
                                  ((2, 9, 4, 4, 4), None, 2), ((1, 5, 9, 10, 40), None, 2), ((2, 3, 16, 16, 16), None, 2),
                                  ((1, 2, 6, 7, 33), (12, 15, 60), None),
                                  # the LDS-tiled forward kernel: ragged tiles in z and y, plane tail, widest rows, 3x along x
                                  ((1, 3, 8, 12, 64), None, 2), ((1, 1, 3, 4, 128), None, 2), ((1, 6, 5, 11, 8), (10, 23, 24), None)])
def test_trilinear_align_corners(case):
    from dram_amd import functional as HF
    shape, size, sf = case
    x = torch.randn(*shape, generator=g(23))
    xr = x.clone().requires_grad_(True)
    yr = O.upsample_trilinear_ac(xr, scale_factor=sf, size=size)
    gy = torch.randn(yr.shape, generator=g(24))
    yr.backward(gy)
    xg = dev(x).requires_grad_(True)
    y = HF.upsample_trilinear_ac(xg, size=size, scale_factor=sf)
    y.backward(dev(gy))
    check(y, yr, f"trilinear fwd {case}", tol=1e-5)
    check(xg.grad, xr.grad, f"trilinear bwd {case}", tol=1e-5)


@pytest.mark.parametrize("case", [(2, 64, 1, (6, 10, 12)), (1, 8, 3, (7, 9, 11)), (2, 5, 11, (4, 4, 8)),
                                  # several output channels on 16-byte rows: the multi-output backward-weights kernel (the 1x1x1
                                  # reshape convs of DC3DATGeneric, models.py:488-494): one partial block / three with a ragged
                                  # last one / one input channel / two tiles of input and of output channels with tails
                                  (2, 64, 8, (8, 16, 16)), (1, 20, 8, (12, 40, 40)), (3, 1, 8, (8, 8, 8)), (2, 13, 9, (4, 40, 60))])
def test_conv1x1_head(case):
    from dram_amd import functional as HF
    N, Ci, Co, sp = case
    x = torch.randn(N, Ci, *sp, generator=g(25))
    w = torch.randn(Co, Ci, 1, 1, 1, generator=g(26))
    b = torch.randn(Co, generator=g(27))
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = O.conv3d(xr, wr, br, 0)
    gy = torch.randn(yr.shape, generator=g(28))
    yr.backward(gy)
    xg, wg, bg = dev(x).requires_grad_(True), dev(w).requires_grad_(True), dev(b).requires_grad_(True)
    y = HF.conv3d_k1(xg, wg, bg)
    y.backward(dev(gy))
    check(y, yr, "conv1x1 fwd")
    check(xg.grad, xr.grad, "conv1x1 dx")
    check(wg.grad, wr.grad, "conv1x1 dw")
    check(bg.grad, br.grad, "conv1x1 db")


def test_conv1x1_bwd_lazy_several_outputs():
    """dram_conv3d_k1_bwd_lazy with more than one output channel (the multi-output backward-weights kernel with the input
    normalised + rectified on load): equals the plain entry on the materialised input."""
    from dram_amd import _lib
    N, Ci, Co, S = 2, 12, 8, 24 * 40 * 12
    st = torch.cuda.current_stream().cuda_stream
    x = dev(torch.randn(N, Ci, S, generator=g(71)))
    dy = dev(torch.randn(N, Co, S, generator=g(72)))
    w = dev(torch.randn(Co, Ci, generator=g(73)))
    coef = dev(torch.rand(N * Ci * 2, generator=g(74)) - 0.3)
    c = coef.view(N, Ci, 2)
    xa = torch.relu(x * c[:, :, 0:1] + c[:, :, 1:2]).contiguous()
    ws = torch.empty(max(16, _lib.lib.dram_conv3d_k1_bwd_ws_bytes(N, Ci, Co, S)), dtype=torch.uint8, device=DEV)
    out = {}
    for name, args in (("lazy", ("dram_conv3d_k1_bwd_lazy", dy.data_ptr(), x.data_ptr(), coef.data_ptr(), 1)),
                       ("plain", ("dram_conv3d_k1_bwd", dy.data_ptr(), xa.data_ptr()))):
        dx, dw, db = torch.empty_like(x), torch.empty_like(w), torch.empty(Co, device=DEV)
        _lib.call(*args, w.data_ptr(), dx.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), ws.numel(), N, Ci, Co, S, st)
        out[name] = (dx, dw, db)
    ref_dw = torch.einsum("nos,ncs->oc", dy.double(), xa.double())
    check(out["lazy"][1], ref_dw, "k1 lazy dw vs fp64")
    check(out["plain"][1], ref_dw, "k1 plain dw vs fp64")
    check(out["lazy"][2], dy.double().sum((0, 2)), "k1 lazy dbias")
    assert torch.equal(out["lazy"][0], out["plain"][0])                       # dx does not look at x


def test_masked_mean_pooling():
    import models
    x = torch.randn(3, 4, 6, 7, 9, generator=g(29))
    m = (torch.rand(3, 1, 6, 7, 9, generator=g(30)) > 0.4).float()
    xr = x.clone().requires_grad_(True)
    yr = O.pooling_dense_features(xr, m)
    gy = torch.randn(yr.shape, generator=g(31))
    yr.backward(gy)
    xg = dev(x).requires_grad_(True)
    y = models.pooling_dense_features(xg, dev(m))
    y.backward(dev(gy))
    check(y, yr, "masked mean fwd")
    check(xg.grad, xr.grad, "masked mean bwd")
    check(models.pooling_dense_features(dev(x), dev(m), "global_avg"), O.pooling_dense_features(x, m, "global_avg"), "global avg")


# ------------------------------------------------------------------ blocks vs the reference's own outputs
def _sub(z, prefix):
    return {k[len(prefix):]: z[k] for k in z.files if k.startswith(prefix)}


def _load_block(block, z, tag):
    sd = {k: torch.from_numpy(v) for k, v in _sub(z, tag + "sd/").items()}
    block.load_state_dict(sd)
    return block.to(DEV)


@pytest.mark.parametrize("norm", ["bn", "ln", "in", "bnt", "bntna", "lnna", "None"])
@pytest.mark.parametrize("shape", ["even", "odd"])
def test_convpool_block_golden(golden_dir, norm, shape):
    import parts
    z = np.load(os.path.join(golden_dir, "blocks.npz"))
    tag = f"convpool/{norm}/{shape}/"
    nm = None if norm == "None" else norm
    blk = parts.ConvPoolBlock5d([3, 4], [4, 6], 0, (3, 3), nm is None, (1, 1), 2, 2, 0, dropout=0.0,
                                norm_method=nm, act_method="relu")
    blk = _load_block(blk, z, tag)
    x = dev(torch.from_numpy(z[tag + "in0"])).requires_grad_(True)
    y, pooled = blk(x)
    check(y, z[tag + "out/y"], tag + "y")
    check(pooled, z[tag + "out/pooled"], tag + "pooled")
    ((y * dev(torch.from_numpy(z[tag + "gout/y"]))).sum() + (pooled * dev(torch.from_numpy(z[tag + "gout/pooled"]))).sum()).backward()
    check(x.grad, z[tag + "gin0"], tag + "gin")
    for k, p in blk.named_parameters():
        check(p.grad, z[tag + "gparam/" + k], tag + "gparam/" + k)
    for k, v in _sub(z, tag + "sd_after/").items():
        check(blk.state_dict()[k].float(), v.astype(np.float32), tag + "sd_after/" + k)
    blk.eval()
    with torch.no_grad():
        ye, pe = blk(x.detach())
    check(ye, z[tag + "eval/y"], tag + "eval y")
    check(pe, z[tag + "eval/pooled"], tag + "eval pooled")


@pytest.mark.parametrize("kind", ["bn", "ln", "lite"])
def test_conv_block_golden(golden_dir, kind):
    import parts
    z = np.load(os.path.join(golden_dir, "blocks.npz"))
    tag = f"conv/{kind}/"
    if kind == "lite":
        blk = parts.ConvBlock5d([5, 8], [8, 7], 0, 3, True, 1, 0.0, lite=True)
    else:
        blk = parts.ConvBlock5d([5, 8], [8, 7], 0, 3, False, 1, 0.0, norm_method=kind)
    blk = _load_block(blk, z, tag)
    x = dev(torch.from_numpy(z[tag + "in0"])).requires_grad_(True)
    y = blk(x)
    check(y, z[tag + "out/y"], tag + "y")
    (y * dev(torch.from_numpy(z[tag + "gout/y"]))).sum().backward()
    check(x.grad, z[tag + "gin0"], tag + "gin")
    for k, p in blk.named_parameters():
        check(p.grad, z[tag + "gparam/" + k], tag + "gparam/" + k)


@pytest.mark.parametrize("norm", ["bn", "in"])
def test_upconv_block_golden(golden_dir, norm):
    import parts
    z = np.load(os.path.join(golden_dir, "blocks.npz"))
    tag = f"upconv/{norm}/"
    blk = parts.UpsampleConvBlock5d([10, 5], [5, 4], 0, (2, 2, 2), (3, 3), False, (1, 1), dropout=0.0, norm_method=norm)
    blk = _load_block(blk, z, tag)
    lo = dev(torch.from_numpy(z[tag + "in0"])).requires_grad_(True)
    cat = dev(torch.from_numpy(z[tag + "in1"])).requires_grad_(True)
    y = blk(lo, cat)
    check(y, z[tag + "out/y"], tag + "y")
    (y * dev(torch.from_numpy(z[tag + "gout/y"]))).sum().backward()
    check(lo.grad, z[tag + "gin0"], tag + "gin0")
    check(cat.grad, z[tag + "gin1"], tag + "gin1")
    for k, p in blk.named_parameters():
        check(p.grad, z[tag + "gparam/" + k], tag + "gparam/" + k)


# ------------------------------------------------------------------ whole model vs the reference's own outputs
def _fp64_oracle_grads(cfg, sd, x, gout, norm):
    """Parameter gradients of the oracle evaluated in float64 (the 'true' values).

    Gradients of the early layers of a BatchNorm U-Net are ill-conditioned in fp32: the
    reference's own fp32 CPU gradients differ from the fp64 ones by ~3e-3 (measured for the slim
    fixture), far above the 1e-4 that holds for activations.  So deep gradients are judged
    against fp64 with the reference's own fp32 error as the yardstick (check_grad)."""
    params, buffers = O.split_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in sd.items()})
    params = {k: v.double().requires_grad_(True) for k, v in params.items()}
    buffers = {k: (v.double() if v.is_floating_point() else v) for k, v in buffers.items()}
    out = O.dc3d_forward(cfg, params, buffers, torch.as_tensor(np.asarray(x)).double(), training=True, norm_method=norm)
    (out * torch.as_tensor(np.asarray(gout)).double()).sum().backward()
    return {k: p.grad for k, p in params.items()}


def check_grads(hip_grads, ref32_grads, true64, what):
    """Noisy (mask-flipping) BatchNorm fixtures: every HIP gradient's error w.r.t. fp64 must stay within 3x the fp32
    noise level of this network, measured as the reference's own worst fp32-vs-fp64 error over the stored gradients
    (floor 5e-4).  The noise comes from single ReLU-mask flips of elements whose pre-activation is within rounding of
    zero, which hit one implementation at one layer and the other at another -- so no per-tensor yardstick taken from
    the reference's run can bound the HIP run (tried in round 2, also in the layer-aware form "reference noise of this
    layer and the layers after it": the HIP path flips an element of us_modules.2.conv_blocks.0 that the reference
    does not, 1.5e-2 on that tensor against a reference noise of 3e-6 there).  The tight, per-tensor check is
    test_dc3d_clean_golden below: a fixture without such elements, where every tensor is held to the reference's own
    fp32 values.  Returns the noise level."""
    noise = max(max(rel_err(ref32_grads[k], true64[k])) for k in ref32_grads)
    bound = max(3.0 * noise, 5e-4)
    worst = {k: max(rel_err(hip_grads[k], true64[k])) for k in ref32_grads}
    bad = {k: v for k, v in worst.items() if v > bound}
    assert not bad, f"{what}: HIP-vs-fp64 gradient errors above {bound:.2e} (reference fp32 noise {noise:.2e}): {bad}"
    return noise


def test_dc3d_clean_golden(golden_dir):
    """Model-level BatchNorm gradient parity without the mask-flip noise: tests/golden/dc3d_clean.npz is a slim
    BatchNorm DC3D whose BatchNorm biases were chosen (oracle/make_golden.py:gen_clean) so that no pre-activation lies
    within 1e-4 sigma of zero -- the reference's own fp32 gradients are then 3e-6 from fp64 instead of 2e-2 -- and
    every HIP parameter gradient must equal the REFERENCE's own fp32 value to 1e-4 (max-abs / max|ref| and rel-L2)."""
    import models
    z = np.load(os.path.join(golden_dir, "dc3d_clean.npz"))
    tag = "slim_bn_clean"
    model = models.DC3D(**SLIM)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in _sub(z, tag + "/sd/").items()})
    model = model.to(DEV).train()
    d0, _ = model(dev(torch.from_numpy(z[tag + "/x"])), None)
    check(d0, z[tag + "/train_out"], tag + " train")
    (d0 * dev(torch.from_numpy(z[tag + "/gout"]))).sum().backward()
    grads = _sub(z, tag + "/grad/")
    params = dict(model.named_parameters())
    assert set(grads) == set(params)
    worst = 0.0
    for k, gref in grads.items():
        check(params[k].grad, gref, f"{tag} grad {k}", tol=1e-4)
        worst = max(worst, max(rel_err(params[k].grad, gref)))
    print(f"\nclean BatchNorm fixture: worst HIP-vs-reference gradient error over {len(grads)} tensors: {worst:.2e}")


@pytest.mark.parametrize("tag,norm", [("slim_bn", "bn"), ("slim_ln", "ln"), ("slim_in_odd", "in")])
def test_dc3d_slim_golden(golden_dir, tag, norm):
    import models
    z = np.load(os.path.join(golden_dir, "dc3d_slim.npz"))
    model = models.DC3D(**SLIM, norm_method=norm)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in _sub(z, tag + "/sd/").items()})
    model = model.to(DEV)
    x = dev(torch.from_numpy(z[tag + "/x"]))
    model.eval()
    with torch.no_grad():
        ev = model(x)[0]
    check(ev, z[tag + "/eval_out"], tag + " eval")
    model.train()
    d0, d1 = model(x, None)
    assert d0 is d1
    check(d0, z[tag + "/train_out"], tag + " train")
    (d0 * dev(torch.from_numpy(z[tag + "/gout"]))).sum().backward()
    grads = dict(model.named_parameters())
    g64 = _fp64_oracle_grads(SLIM, _sub(z, tag + "/sd/"), z[tag + "/x"], z[tag + "/gout"], norm)
    noise = check_grads({k: p.grad for k, p in grads.items()}, _sub(z, tag + "/grad/"), g64, tag)
    if norm != "bn":    # without BatchNorm's batch coupling the gradients are well conditioned
        assert noise < 5e-4
        for k, gref in _sub(z, tag + "/grad/").items():
            check(grads[k].grad, gref, f"{tag} grad {k}", tol=5e-4)   # top_layer.bias: a cancelling sum, ~2e-4
    for k, v in _sub(z, tag + "/sd_after/").items():
        check(model.state_dict()[k].double(), v.astype(np.float64), f"{tag} buffer {k}")


def test_dc3d_full_golden(golden_dir):
    """st_dram_ref.MODEL (16.3 M parameters): weights re-created from seed 0 (identical to the
    reference's, checked on CPU in test_host_cpu.py), output and gradients vs the reference run."""
    import models
    z = np.load(os.path.join(golden_dir, "dc3d_full.npz"))
    torch.manual_seed(0)
    model = models.DC3D(**ST_DRAM_REF_MODEL)
    model.init(models.HeNorm(mode="fan_in"))
    sd0 = {k: v.clone().numpy() for k, v in model.state_dict().items()}
    model = model.to(DEV)
    x = dev(torch.from_numpy(z["full_bn/x"]))
    model.eval()
    with torch.no_grad():
        ev = model(x)[0]
    check(ev, z["full_bn/eval_out"], "full eval")
    model.train()
    d0, _ = model(x)
    check(d0, z["full_bn/train_out"], "full train")
    (d0 * dev(torch.from_numpy(z["full_bn/gout"]))).sum().backward()
    grads = dict(model.named_parameters())
    g64 = _fp64_oracle_grads(ST_DRAM_REF_MODEL, sd0, z["full_bn/x"], z["full_bn/gout"], "bn")
    check_grads({k: p.grad for k, p in grads.items()}, _sub(z, "full_bn/grad/"), g64, "full_bn")
    ref_norms = _sub(z, "full_bn/gradnorm/")
    nerr_ref = max(abs(float(v) - g64[k].norm().item()) / g64[k].norm().item() for k, v in ref_norms.items())
    for k in ref_norms:
        n64 = g64[k].norm().item()
        got = grads[k].grad.double().norm().item()
        assert abs(got - n64) / n64 <= max(3.0 * nerr_ref, 5e-4), (k, got, n64, nerr_ref)
    for k, v in _sub(z, "full_bn/sd_after/").items():
        check(model.state_dict()[k].double(), v.astype(np.float64), f"full buffer {k}")
    assert int(model.state_dict()["ds_modules.1.conv_blocks.0.1.num_batches_tracked"]) == 2


def test_checkpoint_modes_agree(golden_dir):
    """DC3D.checkpoint_mode 'stats' (no recomputation, BatchNorm buffers updated twice) and 'recompute'
    (torch.utils.checkpoint like the reference) give the same step: bit for bit on the per-op path (same kernels, same
    inputs), and to rounding when 'stats' runs through the fused engine (its norm statistics are summed in a
    different order)."""
    import models
    z = np.load(os.path.join(golden_dir, "dc3d_slim.npz"))
    res = {}
    for mode, fused in (("stats", False), ("recompute", False), ("stats", True)):
        model = models.DC3D(**SLIM)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in _sub(z, "slim_bn/sd/").items()})
        model.checkpoint_mode = mode
        model.fused = fused
        model = model.to(DEV).train()
        d0, _ = model(dev(torch.from_numpy(z["slim_bn/x"])))
        (d0 * dev(torch.from_numpy(z["slim_bn/gout"]))).sum().backward()
        res[(mode, fused)] = (d0.detach().cpu(), {k: p.grad.cpu() for k, p in model.named_parameters()},
                              {k: v.cpu().double() for k, v in model.state_dict().items() if "running" in k or "num_batches" in k})
    a, b, c = res[("stats", False)], res[("recompute", False)], res[("stats", True)]
    assert torch.equal(a[0], b[0])
    for k in a[1]:
        assert torch.equal(a[1][k], b[1][k]), k     # same kernels, same inputs: bit-exact
    for k in a[2]:
        check(a[2][k], b[2][k], f"buffer {k}", tol=1e-6)
    check(c[0], b[0], "fused engine output", tol=1e-5)
    for k in c[2]:
        check(c[2][k], b[2][k], f"fused engine buffer {k}", tol=1e-5)
