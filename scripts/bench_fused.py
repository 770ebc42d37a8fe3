"""Micro-benchmark of the fused conv variants against the plain kernels (HIP events):
forward plain / +statistics epilogue / +lazy operand / both; backward-weights plain / lazy operand."""
import argparse, os, sys
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bodyct-dram_amd")]
from dram_amd import functional as HF
from dram_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", default="4,64,64,128;4,32,64,128;4,192,64,128;8,384,128,64;16,256,256,32")
ap.add_argument("--iters", type=int, default=5)
args = ap.parse_args()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
p = lambda t: None if t is None else t.data_ptr()


def timeit(fn, iters):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def cat_case(spec):
    """'N,C1+C2,Co,S': the first conv of an UpsampleConvBlock5d -- x1 (C1 channels) plain, x2 (C2 channels) lazy, virtual concat --
    forward with statistics and backward-weights, as the fused engine launches them."""
    N, cc, Co, S = spec.split(",")
    N, Co, S = int(N), int(Co), int(S)
    C1, C2 = (int(v) for v in cc.split("+"))
    Ci = C1 + C2
    x1 = torch.rand(N, C1, S, S, S, device=dev) - 0.5
    x2 = torch.rand(N, C2, S, S, S, device=dev) - 0.5
    dy = torch.rand(N, Co, S, S, S, device=dev) - 0.5
    w = torch.randn(Co, Ci, 3, 3, 3, device=dev) / (Ci * 27) ** 0.5
    coef2 = torch.rand(N * C2 * 2, device=dev) + 0.5
    wt = HF._pack(w, 0)
    y = torch.empty(N, Co, S, S, S, device=dev)
    dw = torch.empty_like(w)
    nb = _lib.lib.dram_conv3d_k3_wgrad_ws_bytes(N, Ci, Co, S, S, S)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    nparts = _lib.lib.dram_conv3d_k3_stats_parts(Ci, Co, S, S, S)
    parts = torch.empty(N * Co * nparts * 3, device=dev)
    flops = 54.0 * Ci * Co * N * S ** 3
    f = lambda c2: (lambda: _lib.call("dram_conv3d_k3_fwd_fused", p(x1), C1, None, 0, p(x2), C2, p(c2), 1, S, S, S, 0, 0, 0, p(wt), None,
                                      p(y), p(parts), nparts, N, Co, S, S, S, st))
    g = lambda c2: (lambda: _lib.call("dram_conv3d_k3_wgrad_fused", p(x1), C1, None, 0, p(x2), C2, p(c2), 1, S, S, S, 0, 0, 0, p(dy), p(dw),
                                      p(ws), nb, N, Co, S, S, S, st))
    t = {k: timeit(fn, args.iters) for k, fn in [("f", f(None)), ("fl", f(coef2)), ("g", g(None)), ("gl", g(coef2))]}
    tf = lambda ms: flops / ms / 1e9
    print(f"[{N},{C1}+{C2}->{Co},{S}^3] fwd + stats, both plain {t['f']:7.3f} ms {tf(t['f']):6.1f} | x2 lazy {t['fl']:7.3f} ({100 * (t['fl'] / t['f'] - 1):+.1f}%)"
          f" || wgrad both plain {t['g']:7.3f} ms {tf(t['g']):6.1f} | x2 lazy {t['gl']:7.3f} ({100 * (t['gl'] / t['g'] - 1):+.1f}%)", flush=True)


for spec in args.shapes.split(";"):
    if "+" in spec:
        cat_case(spec)
        continue
    N, Ci, Co, S = (int(v) for v in spec.split(","))
    x = torch.rand(N, Ci, S, S, S, device=dev) - 0.5
    dy = torch.rand(N, Co, S, S, S, device=dev) - 0.5
    w = torch.randn(Co, Ci, 3, 3, 3, device=dev) / (Ci * 27) ** 0.5
    coef = torch.rand(N * Ci * 2, device=dev) + 0.5
    wt = HF._pack(w, 0)
    y = torch.empty(N, Co, S, S, S, device=dev)
    dw = torch.empty_like(w)
    nb = _lib.lib.dram_conv3d_k3_wgrad_ws_bytes(N, Ci, Co, S, S, S)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    nparts = _lib.lib.dram_conv3d_k3_stats_parts(Ci, Co, S, S, S)
    parts = torch.empty(N * Co * nparts * 3, device=dev)
    flops = 54.0 * Ci * Co * N * S ** 3

    def fwd(cf, stt):
        return lambda: _lib.call("dram_conv3d_k3_fwd_fused", p(x), Ci, p(cf), 1, None, 0, None, 0, 0, 0, 0, 0, 0, 0, p(wt), None,
                                 p(y), p(stt), nparts if stt is not None else 0, N, Co, S, S, S, st)

    def wg(cf):
        return lambda: _lib.call("dram_conv3d_k3_wgrad_fused", p(x), Ci, p(cf), 1, None, 0, None, 0, 0, 0, 0, 0, 0, 0, p(dy), p(dw),
                                 p(ws), nb, N, Co, S, S, S, st)
    t = {k: timeit(f, args.iters) for k, f in [("plain", fwd(None, None)), ("stats", fwd(None, parts)), ("lazy", fwd(coef, None)),
                                                 ("both", fwd(coef, parts)), ("wg", wg(None)), ("wg_lazy", wg(coef))]}
    tf = lambda ms: flops / ms / 1e9
    print(f"[{N},{Ci}->{Co},{S}^3] fwd plain {t['plain']:7.3f} ms {tf(t['plain']):6.1f} | +stats {t['stats']:7.3f} ({100 * (t['stats'] / t['plain'] - 1):+.1f}%)"
          f" | +lazy {t['lazy']:7.3f} ({100 * (t['lazy'] / t['plain'] - 1):+.1f}%) | both {t['both']:7.3f} ({100 * (t['both'] / t['plain'] - 1):+.1f}%)"
          f" || wgrad plain {t['wg']:7.3f} ms {tf(t['wg']):6.1f} | lazy {t['wg_lazy']:7.3f} ({100 * (t['wg_lazy'] / t['wg'] - 1):+.1f}%)", flush=True)
