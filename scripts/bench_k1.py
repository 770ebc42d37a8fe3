"""Micro-benchmark of the 1x1x1 head's backward (dx, dw, dbias) on the benchmark's shape, plain and lazy."""
import os, sys, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bodyct-dram_amd")]
from dram_amd import _lib
st = torch.cuda.current_stream().cuda_stream
def t(fn, it=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it
for (N, Ci, Co, S) in [(16, 64, 1, 128 ** 3), (64, 64, 1, 64 ** 3)]:
    x = torch.rand(N, Ci, S, device="cuda"); dy = torch.rand(N, Co, S, device="cuda"); w = torch.rand(Co, Ci, device="cuda")
    coef = torch.rand(N * Ci * 2, device="cuda")
    dw = torch.empty_like(w); db = torch.empty(Co, device="cuda")
    nb = _lib.lib.dram_conv3d_k1_bwd_ws_bytes(N, Ci, Co, S)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    tw = t(lambda: _lib.call("dram_conv3d_k1_bwd", dy.data_ptr(), x.data_ptr(), w.data_ptr(), None, dw.data_ptr(), db.data_ptr(), ws.data_ptr(), nb, N, Ci, Co, S, st))
    tl = t(lambda: _lib.call("dram_conv3d_k1_bwd_lazy", dy.data_ptr(), x.data_ptr(), coef.data_ptr(), 1, w.data_ptr(), None, dw.data_ptr(), db.data_ptr(), ws.data_ptr(), nb, N, Ci, Co, S, st))
    gb = (x.numel() + dy.numel()) * 4 / 1e9
    print(f"[{N},{Ci}->{Co},S={S}] wgrad {tw:.2f} ms ({gb / tw * 1e3:.0f} GB/s), lazy {tl:.2f} ms ({gb / tl * 1e3:.0f} GB/s)", flush=True)
