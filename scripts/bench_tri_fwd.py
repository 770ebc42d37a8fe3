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
for (N, C, S) in [(16, 128, 64), (16, 256, 32), (16, 512, 16)]:
    x = torch.rand(N, C, S, S, S, device="cuda"); y = torch.empty(N, C, 2 * S, 2 * S, 2 * S, device="cuda")
    coef = torch.rand(N * C * 2, device="cuda")
    tf = t(lambda: _lib.call("dram_upsample_trilinear_ac_fwd", x.data_ptr(), y.data_ptr(), N, C, S, S, S, 2 * S, 2 * S, 2 * S, st))
    tl = t(lambda: _lib.call("dram_upsample_trilinear_ac_fwd_lazy", x.data_ptr(), coef.data_ptr(), 1, y.data_ptr(), N, C, S, S, S, 2 * S, 2 * S, 2 * S, st))
    gb = (y.numel() + x.numel()) * 4 / 1e9
    print(f"[{N},{C},{S}^3 -> {2*S}^3] fwd {tf:.2f} ms ({gb / tf * 1e3:.0f} GB/s read+write), lazy {tl:.2f} ms")
