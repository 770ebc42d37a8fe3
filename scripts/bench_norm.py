"""Micro-benchmark of the norm kernels of the fused engine's backward (dram_norm_bwd: reduce + finalise + apply, in place) and of
dram_row_affine_act at a full-resolution stage of the benchmark: [N, 64, 128^3]."""
import os, sys
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bodyct-dram_amd")]
from dram_amd import _lib
N, C, S = int(os.environ.get("N", 16)), 64, 128 ** 3
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
g = torch.rand(N, C, S, device=dev); y = torch.rand(N, C, S, device=dev)
gamma = torch.rand(C, device=dev) + 0.5
mean = torch.rand(C, device=dev); rstd = torch.rand(C, device=dev) + 0.5
coef = torch.rand(2 * N * C, device=dev)
dgamma = torch.empty(C, device=dev); dbeta = torch.empty(C, device=dev)
ws = torch.empty(_lib.lib.dram_norm_ws_bytes(N, C, S), dtype=torch.uint8, device=dev)
out = torch.empty_like(g)
p = lambda t: t.data_ptr()
def bwd():
    _lib.call("dram_norm_bwd", p(g), p(y), p(gamma), p(mean), p(rstd), p(coef), p(g), p(dgamma), p(dbeta), 0, 1, 1, 1, N, C, S, p(ws), ws.numel(), st)
def act():
    _lib.call("dram_row_affine_act", p(y), p(coef), p(out), 1, N * C, S, st)
def t(fn, it=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
nb = g.numel() * 4
tb, ta = t(bwd), t(act)
print(f"norm_bwd [{N},{C},128^3]: {tb:.3f} ms = {5 * nb / tb / 1e9:.2f} TB/s (4R+1W)   row_affine_act: {ta:.3f} ms = {2 * nb / ta / 1e9:.2f} TB/s (1R+1W)")
