import os, sys, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bodyct-dram_amd")]
from dram_amd import _lib
st = torch.cuda.current_stream().cuda_stream
for (N, C, S) in [(16, 128, 64), (16, 256, 32), (16, 512, 16)]:
    dy = torch.rand(N, C, 2 * S, 2 * S, 2 * S, device="cuda"); dx = torch.empty(N, C, S, S, S, device="cuda")
    x = torch.rand(N, C, S, S, S, device="cuda"); y = torch.empty_like(dy)
    def t(fn, it=5):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it): fn()
        e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it
    tb = t(lambda: _lib.call("dram_upsample_trilinear_ac_bwd", dy.data_ptr(), dx.data_ptr(), N, C, S, S, S, 2 * S, 2 * S, 2 * S, st))
    for mb in (32, 128, 512, 2048, 1 << 20):
        full = _lib.lib.dram_upsample_trilinear_ac_bwd_ws_bytes(N, C, S, S, S, 2 * S, 2 * S, 2 * S)
        ws = torch.empty(min(full, mb << 20), dtype=torch.uint8, device="cuda")
        dx2 = torch.empty_like(dx)
        t2 = t(lambda: _lib.call("dram_upsample_trilinear_ac_bwd_ws", dy.data_ptr(), dx2.data_ptr(), ws.data_ptr(), ws.numel(), N, C, S, S, S, 2 * S, 2 * S, 2 * S, st))
        err = ((dx2 - dx).abs().max() / dx.abs().max()).item()
        print(f"    two-stage, workspace {ws.numel() >> 20} MB: {t2:.2f} ms, rel diff vs single-stage {err:.1e}")
        del ws
    tf = t(lambda: _lib.call("dram_upsample_trilinear_ac_fwd", x.data_ptr(), y.data_ptr(), N, C, S, S, S, 2 * S, 2 * S, 2 * S, st))
    gb = dy.numel() * 4 / 1e9
    print(f"[{N},{C},{S}^3 -> {2*S}^3] bwd {tb:.2f} ms ({gb / tb * 1e3:.0f} GB/s of dy), fwd {tf:.2f} ms ({gb / tf * 1e3:.0f} GB/s of y)")
