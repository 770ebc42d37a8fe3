"""Layer-level micro-benchmark of the 3x3x3 conv kernels (config 2 of BASELINE.json:
[4,64,128^3] 64->64 by default).  Prints ms and TFLOP/s per kernel, HIP-event timed."""
import argparse, os, sys
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bodyct-dram_amd")]
from dram_amd import functional as HF
from dram_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", default="4,64,64,128;4,192,64,128;8,384,128,64;16,768,256,32;16,256,512,16")
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--only", default="both", choices=["both", "fwd", "wgrad"])
args = ap.parse_args()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream


def timeit(fn, iters):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for spec in args.shapes.split(";"):
    N, Ci, Co, S = (int(v) for v in spec.split(","))
    x = torch.rand(N, Ci, S, S, S, device=dev) - 0.5
    dy = torch.rand(N, Co, S, S, S, device=dev) - 0.5
    w = torch.randn(Co, Ci, 3, 3, 3, device=dev) / (Ci * 27) ** 0.5
    wt = HF._pack(w, 0)
    y = torch.empty(N, Co, S, S, S, device=dev)
    dw = torch.empty_like(w)
    nb = _lib.lib.dram_conv3d_k3_wgrad_ws_bytes(N, Ci, Co, S, S, S)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    flops = 54.0 * Ci * Co * N * S ** 3
    p = lambda t: t.data_ptr()
    t_f = t_w = float("nan")
    if args.only in ("both", "fwd"):
        t_f = timeit(lambda: _lib.call("dram_conv3d_k3_fwd", p(x), p(wt), None, p(y), N, Ci, Co, S, S, S, st), args.iters)
    if args.only in ("both", "wgrad"):
        t_w = timeit(lambda: _lib.call("dram_conv3d_k3_wgrad", p(x), p(dy), p(dw), p(ws), nb, N, Ci, Co, S, S, S, st), args.iters)
    print(f"[{N},{Ci}->{Co},{S}^3] fwd {t_f:8.3f} ms {flops / t_f / 1e9:7.1f} TF/s | wgrad {t_w:8.3f} ms {flops / t_w / 1e9:7.1f} TF/s", flush=True)
