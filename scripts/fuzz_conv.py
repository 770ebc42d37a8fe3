"""Randomised sweep of the 3x3x3 conv entry points against an fp64 reference on small random shapes: fused forward (lazy
sources, virtual concat with a cropped skip, statistics), backward-data with a split destination, fused backward-weights
(no / one / two lazy sources) -- whichever kernel the library picks for each shape (the launch counters say which).
    python scripts/fuzz_conv.py [--cases 60] [--seed 1]
Exit status 1 on the first mismatch (prints the case)."""
import argparse, os, sys
import numpy as np
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bodyct-dram_amd")]
from dram_amd import functional as HF
from dram_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=60)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--wzy", action="store_true", help="bias the shapes towards the Winograd-(z,y) kernels (Cout % 64 == 0, W % 4 == 0, even H / D)")
args = ap.parse_args()
rng = np.random.default_rng(args.seed)
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
p = lambda t: None if t is None else t.data_ptr()
names = ["fwd_direct", "fwd_wz", "fwd_wzy", "wg_direct", "wg_vec", "wg_wz", "wg_wz_lazy", "wg_c1", "fwd_c1", "wg_wzy"]
seen = np.zeros(len(names), dtype=np.int64)


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def act(t, cf):
    if cf is None:
        return t.double()
    c = cf.view(t.shape[0], t.shape[1], 2).double()
    return torch.relu(t.double() * c[:, :, 0, None, None, None] + c[:, :, 1, None, None, None])


for case in range(args.cases):
    N = int(rng.integers(1, 4))
    W = int(rng.choice([4, 8, 12, 16, 20, 24, 28, 32, 36, 40, 48, 7, 10, 33]))
    H = int(rng.integers(1, 13)) if rng.random() < 0.3 else int(rng.choice([2, 4, 6, 8, 10, 12, 16]))
    D = int(rng.integers(1, 10)) if rng.random() < 0.3 else int(rng.choice([2, 4, 6, 8, 12]))
    Co = int(rng.choice([8, 16, 32, 40, 64, 64, 64, 72, 128, 192]))
    C1 = int(rng.choice([1, 4, 8, 12, 16, 16, 32, 32, 48, 64]))
    C2 = int(rng.choice([0, 0, 8, 16, 16, 32]))
    if C1 == 1:
        C2 = 0
    if args.wzy:
        W = int(rng.choice([16, 20, 24, 28, 32, 40, 48, 56, 64, 72]))
        H = int(rng.choice([2, 4, 6, 8, 10, 12, 5, 7]))
        D = int(rng.choice([2, 4, 6, 8, 10, 3, 7]))
        Co = int(rng.choice([64, 64, 128, 192]))
        C1 = int(rng.choice([8, 12, 16, 32, 48, 64, 96]))
        C2 = int(rng.choice([0, 0, 16, 32, 64]))
    Ci = C1 + C2
    ext = tuple(int(v) for v in rng.choice([0, 0, 1, 2, 3, 4, 8], size=3)) if C2 else (0, 0, 0)
    if args.wzy and C2 and rng.random() < 0.6:
        ext = (ext[0], ext[1], int(rng.choice([0, 8, 16])))          # 16-byte multiples along x: the (z,y) forward kernel takes the skip
    lazy1, lazy2 = bool(rng.random() < 0.5) and C1 > 1, bool(rng.random() < 0.5) and C2 > 0
    stats = bool(rng.random() < 0.6)
    desc = dict(N=N, C1=C1, C2=C2, Co=Co, D=D, H=H, W=W, ext=ext, lazy1=lazy1, lazy2=lazy2, stats=stats)
    g = torch.Generator().manual_seed(1000 + case)
    x1 = torch.randn(N, C1, D, H, W, generator=g).to(dev)
    x2 = torch.randn(N, C2, D + ext[0], H + ext[1], W + ext[2], generator=g).to(dev) if C2 else None
    crop = tuple(int(np.ceil(e / 2)) for e in ext)
    d2 = tuple(x2.shape[2:]) if C2 else (0, 0, 0)
    w = (torch.randn(Co, Ci, 3, 3, 3, generator=g) / (Ci * 27) ** 0.5).to(dev)
    dy = torch.randn(N, Co, D, H, W, generator=g).to(dev)
    cf1 = (torch.rand(N * C1 * 2, generator=g) + 0.3).to(dev) if lazy1 else None
    cf2 = (torch.rand(N * C2 * 2, generator=g) - 0.2).to(dev) if lazy2 else None
    xin = act(x1, cf1)
    if C2:
        xin = torch.cat([xin, act(x2[:, :, crop[0]:crop[0] + D, crop[1]:crop[1] + H, crop[2]:crop[2] + W], cf2)], 1)
    xr = xin.cpu().requires_grad_(True)
    wr = w.double().cpu().requires_grad_(True)
    yr = torch.nn.functional.conv3d(xr, wr, None, padding=1)
    yr.backward(dy.double().cpu())
    before = HF.conv_launch_counts()
    try:
        # fused forward
        wt = HF._pack(w, 0)
        y = torch.full((N, Co, D, H, W), float("nan"), device=dev)
        nparts = _lib.lib.dram_conv3d_k3_stats_parts(Ci, Co, D, H, W) if stats else 0
        parts = torch.full((N * Co * max(nparts, 1) * 3,), float("nan"), device=dev) if stats else None
        _lib.call("dram_conv3d_k3_fwd_fused", p(x1), C1, p(cf1), 1, p(x2), C2, p(cf2), 1, *d2, *crop, p(wt), None, p(y), p(parts), nparts,
                  N, Co, D, H, W, st)
        e = rel(y, yr)
        assert e <= 1e-4, ("forward", e)
        if stats:
            q = parts.view(N * Co, nparts, 3).double()
            cnt = q[:, :, 2].sum(1)
            assert bool((cnt == D * H * W).all()), ("statistics count", cnt.min().item(), cnt.max().item(), D * H * W)
            mean = (q[:, :, 0] * q[:, :, 2]).sum(1) / cnt
            assert float((mean - y.double().view(N * Co, -1).mean(1)).abs().max()) < 1e-4, "statistics mean"
        # backward-data (split destination when the source was a concat)
        wtb = HF._pack(w, 1)
        dx1 = torch.full((N, C1, D, H, W), float("nan"), device=dev)
        dx2 = torch.zeros_like(x2) if C2 else None
        _lib.call("dram_conv3d_k3_fwd_ex", p(dy), Co, None, 0, 0, 0, 0, 0, 0, 0, p(wtb), None, p(dx1), C1, p(dx2), C2, *d2, *crop,
                  N, D, H, W, st)
        gx = xr.grad
        if not (lazy1 or lazy2):        # d(activated input): comparable as it is only for plain sources
            assert rel(dx1, gx[:, :C1]) <= 1e-4, "backward-data (first source)"
            if C2:
                win = dx2[:, :, crop[0]:crop[0] + D, crop[1]:crop[1] + H, crop[2]:crop[2] + W]
                assert rel(win, gx[:, C1:]) <= 1e-4, "backward-data (second source)"
                rest = dx2.clone()
                rest[:, :, crop[0]:crop[0] + D, crop[1]:crop[1] + H, crop[2]:crop[2] + W] = 0
                assert float(rest.abs().max()) == 0.0, "backward-data wrote outside the crop window"
        # fused backward-weights
        lazy_ok = bool(_lib.lib.dram_conv3d_k3_wgrad_lazy_ok(N, C1, C2, Co, D, H, W))
        a1, a2, c1_, c2_ = x1, x2, cf1, cf2
        if (lazy1 or lazy2) and not lazy_ok:      # the engine materialises then
            a1 = act(x1, cf1).float(); c1_ = None
            if C2:
                a2 = act(x2, cf2).float(); c2_ = None
        dw = torch.full_like(w, float("nan"))
        ws = torch.empty(max(16, _lib.lib.dram_conv3d_k3_wgrad_ws_bytes(N, Ci, Co, D, H, W)), dtype=torch.uint8, device=dev)
        _lib.call("dram_conv3d_k3_wgrad_fused", p(a1), C1, p(c1_), 1, p(a2), C2, p(c2_), 1, *d2, *crop, p(dy), p(dw), p(ws), ws.numel(),
                  N, Co, D, H, W, st)
        e = rel(dw, wr.grad)
        assert e <= 1e-4, ("backward-weights", e)
        torch.cuda.synchronize()
    except Exception as ex:
        print("MISMATCH / ERROR in case", case, desc, "->", repr(ex), flush=True)
        sys.exit(1)
    delta = np.array(HF.conv_launch_counts()) - np.array(before)
    seen += delta
    print(case, desc, {n: int(v) for n, v in zip(names, delta) if v}, flush=True)
print("all", args.cases, "cases agree; launches by kernel family:", {n: int(v) for n, v in zip(names, seen)})
