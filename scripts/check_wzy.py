"""Winograd-(z,y) forward kernel against the z-only kernel (same process, DRAM_CONV_NO_WZY toggled per call) and an
fp64 reference on small shapes; then timing of both on the layer shapes of DC3D(st_dram_ref)."""
import argparse, os, sys
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bodyct-dram_amd")]
from dram_amd import functional as HF
from dram_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--no-check", action="store_true")
ap.add_argument("--relu2", type=int, default=0)
ap.add_argument("--shapes", default="4,32,64,128;4,64,64,128;4,192,64,128;8,64,128,64;8,384,128,64;8,128,128,64;16,128,256,32;16,768,256,32;16,256,256,32")
args = ap.parse_args()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
p = lambda t: None if t is None else t.data_ptr()


def run(x, wt, Co, coef=None, stats=False, x2=None, crop=(0, 0, 0), coef2=None, bias=None, wzy=True):
    if wzy:
        os.environ.pop("DRAM_CONV_NO_WZY", None)
    else:
        os.environ["DRAM_CONV_NO_WZY"] = "1"
    N, C1, D, H, W = x.shape
    C2 = 0 if x2 is None else x2.shape[1]
    y = torch.full((N, Co, D, H, W), float("nan"), device=dev)
    parts = None
    nparts = 0
    if stats:
        nparts = _lib.lib.dram_conv3d_k3_stats_parts(C1 + C2, Co, D, H, W)
        parts = torch.full((N * Co * nparts * 3,), float("nan"), device=dev)
    d2 = (0, 0, 0) if x2 is None else tuple(x2.shape[2:])
    _lib.call("dram_conv3d_k3_fwd_fused", p(x), C1, p(coef), 1, p(x2), C2, p(coef2), 1, d2[0], d2[1], d2[2], crop[0], crop[1], crop[2],
              p(wt), p(bias), p(y), p(parts), nparts, N, Co, D, H, W, st)
    torch.cuda.synchronize()
    os.environ.pop("DRAM_CONV_NO_WZY", None)
    return y, parts, nparts


def moments(parts, nparts, rows):
    q = parts.view(rows, nparts, 3).double()
    cnt = q[:, :, 2].sum(1)
    mean = (q[:, :, 0] * q[:, :, 2]).sum(1) / cnt
    m2 = (q[:, :, 1] + q[:, :, 2] * (q[:, :, 0] - mean[:, None]) ** 2).sum(1)
    return cnt, mean, m2


def check():
    torch.manual_seed(0)
    worst = 0.0
    cases = [  # N, C1, C2, Co, D, H, W, lazy, stats, bias
        (2, 16, 0, 64, 6, 8, 32, False, False, False),
        (1, 12, 0, 64, 11, 8, 32, True, True, False),      # odd D, channel tail (12 = 3 chunks)
        (1, 16, 0, 64, 12, 11, 58, True, True, False),     # ragged y pair, ragged x box
        (2, 10, 0, 128, 4, 12, 64, True, False, False),    # Cin tail inside a chunk
        (1, 8, 8, 64, 8, 8, 32, True, True, False),        # virtual concat, both lazy
        (1, 24, 0, 64, 2, 4, 58, False, False, True),      # W not a multiple of 32 (padding 1.10), bias
        (1, 64, 0, 192, 8, 16, 32, False, True, False),
    ]
    for (N, C1, C2, Co, D, H, W, lazy, stats, bias) in cases:
        Ci = C1 + C2
        x = torch.randn(N, C1, D, H, W, device=dev)
        x2 = torch.randn(N, C2, D + 2, H + 3, W + 1, device=dev) if C2 else None
        crop = (1, 2, 1) if C2 else (0, 0, 0)
        w = torch.randn(Co, Ci, 3, 3, 3, device=dev) / (Ci * 27) ** 0.5
        b = torch.randn(Co, device=dev) if bias else None
        coef = (torch.rand(N * C1 * 2, device=dev) + 0.5) if lazy else None
        coef2 = (torch.rand(N * C2 * 2, device=dev) - 0.2) if (lazy and C2) else None
        wt = HF._pack(w, 0)
        ya, pa, na = run(x, wt, Co, coef, stats, x2, crop, coef2, b, wzy=True)
        yb, pb, nb = run(x, wt, Co, coef, stats, x2, crop, coef2, b, wzy=False)
        # fp64 reference
        def act(t, cf):
            if cf is None:
                return t.double()
            c = cf.view(t.shape[0], t.shape[1], 2).double()
            return torch.relu(t.double() * c[:, :, 0, None, None, None] + c[:, :, 1, None, None, None])
        xin = act(x, coef)
        if C2:
            x2c = x2[:, :, crop[0]:crop[0] + D, crop[1]:crop[1] + H, crop[2]:crop[2] + W]
            xin = torch.cat([xin, act(x2c, coef2)], 1)
        ref = torch.nn.functional.conv3d(xin.cpu(), w.double().cpu(), None if b is None else b.double().cpu(), padding=1).to(dev)
        sc = ref.abs().max().item()
        ea = (ya.double() - ref).abs().max().item() / sc
        eb = (yb.double() - ref).abs().max().item() / sc
        msg = f"[{N},{C1}+{C2}->{Co},{D}x{H}x{W}] lazy={lazy} stats={stats} bias={bias}: wzy {ea:.2e}  wz {eb:.2e}"
        if stats:
            ca, ma, qa = moments(pa, na, N * Co)
            r = ya.double().view(N * Co, -1)
            em = (ma - r.mean(1)).abs().max().item()
            eq = ((qa - ((r - r.mean(1, keepdim=True)) ** 2).sum(1)).abs() / qa).max().item()
            okc = bool((ca == D * H * W).all())
            msg += f" | stats: count ok {okc}, mean err {em:.2e}, M2 rel err {eq:.2e} (parts {na})"
            assert okc and em < 1e-5 and eq < 1e-4, msg
        print(msg, flush=True)
        assert ea < 2e-5 and not torch.isnan(ya).any(), msg
        worst = max(worst, ea)
    print("check OK, worst", worst)


def timeit(fn, iters):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


if not args.no_check:
    check()
for spec in args.shapes.split(";"):
    N, Ci, Co, S = (int(v) for v in spec.split(","))
    x = torch.rand(N, Ci, S, S, S, device=dev) - 0.5
    w = torch.randn(Co, Ci, 3, 3, 3, device=dev) / (Ci * 27) ** 0.5
    coef = torch.rand(N * Ci * 2, device=dev) + 0.5
    wt = HF._pack(w, 0)
    y = torch.empty(N, Co, S, S, S, device=dev)
    nparts = _lib.lib.dram_conv3d_k3_stats_parts(Ci, Co, S, S, S)
    parts = torch.empty(N * Co * nparts * 3, device=dev)
    flops = 54.0 * Ci * Co * N * S ** 3

    def fwd(cf, stt):
        return lambda: _lib.call("dram_conv3d_k3_fwd_fused", p(x), Ci, p(cf), 1, None, 0, None, args.relu2, 0, 0, 0, 0, 0, 0, p(wt), None,
                                 p(y), p(stt), nparts if stt is not None else 0, N, Co, S, S, S, st)
    out = []
    for mode in ("wzy", "wz"):
        if mode == "wz":
            os.environ["DRAM_CONV_NO_WZY"] = "1"
        else:
            os.environ.pop("DRAM_CONV_NO_WZY", None)
        tp = timeit(fwd(None, None), args.iters)
        tb = timeit(fwd(coef, parts), args.iters)
        tl = timeit(fwd(coef, None), args.iters)
        ts = timeit(fwd(None, parts), args.iters)
        out.append(f"{mode}: plain {tp:7.3f} ms ({flops / tp / 1e9:6.1f} TF/s direct-equiv) fused {tb:7.3f} ms ({flops / tb / 1e9:6.1f}) lazy-only {tl:7.3f} stats-only {ts:7.3f}")
    os.environ.pop("DRAM_CONV_NO_WZY", None)
    print(f"[{N},{Ci}->{Co},{S}^3] " + " | ".join(out), flush=True)
