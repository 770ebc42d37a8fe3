import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bodyct-dram_amd")]
import torch
from dram_amd import _lib, functional as HF
call = _lib.call
dev = "cuda"
g = torch.Generator().manual_seed(3)
def run(kind, G, N, Ci, Co, D, H, W):
    x = torch.randn(N, Ci, D, H, W, generator=g).to(dev)
    w = (torch.randn(Co, Ci, 3, 3, 3, generator=g) / (27 * Ci) ** 0.5).to(dev)
    gamma = (1 + 0.3 * torch.randn(Co, generator=g)).to(dev); beta = (0.2 * torch.randn(Co, generator=g)).to(dev)
    wt = HF._pack(w, 0)
    st = torch.cuda.current_stream().cuda_stream
    S = D * H * W
    y = torch.empty(N, Co, D, H, W, device=dev)
    nparts = _lib.lib.dram_conv3d_k3_stats_parts(Ci, Co, D, H, W)
    parts = torch.full((N * Co * nparts * 3,), float("nan"), device=dev)
    call("dram_conv3d_k3_fwd_fused", x.data_ptr(), Ci, None, 0, None, 0, None, 0, 0, 0, 0, 0, 0, 0, wt.data_ptr(), None,
         y.data_ptr(), parts.data_ptr(), nparts, N, Co, D, H, W, st)
    nstat = Co if kind == 0 else N * G
    m1 = torch.empty(nstat, device=dev); r1 = torch.empty(nstat, device=dev); c1 = torch.empty(2 * N * Co, device=dev)
    ws = torch.empty(max(16, _lib.lib.dram_norm_parts_ws_bytes(N, Co, nparts)), dtype=torch.uint8, device=dev)
    call("dram_norm_finalize_parts", parts.data_ptr(), nparts, gamma.data_ptr(), beta.data_ptr(), m1.data_ptr(), r1.data_ptr(), c1.data_ptr(),
         None, None, 0.0, 1e-5, kind, G, N, Co, S, ws.data_ptr(), ws.numel(), st)
    m2 = torch.empty(nstat, device=dev); r2 = torch.empty(nstat, device=dev); c2 = torch.empty(2 * N * Co, device=dev)
    yo = torch.empty_like(y)
    ws2 = torch.empty(max(16, _lib.lib.dram_norm_ws_bytes(N, Co, S)), dtype=torch.uint8, device=dev)
    call("dram_norm_fwd_train", y.data_ptr(), gamma.data_ptr(), beta.data_ptr(), yo.data_ptr(), m2.data_ptr(), r2.data_ptr(), c2.data_ptr(),
         None, None, 0.0, 1e-5, kind, G, 1, N, Co, S, ws2.data_ptr(), ws2.numel(), st)
    rel = lambda a, b: ((a - b).abs().max() / b.abs().max()).item()
    print(f"kind {kind} G {G} N {N} C {Ci}->{Co} {D}x{H}x{W}: nparts {nparts} mean {rel(m1, m2):.2e} rstd {rel(r1, r2):.2e} coef {rel(c1, c2):.2e}")
    if rel(m1, m2) > 1e-4 or rel(r1,r2) > 1e-4:
        print("   mean", m1.tolist()[:4], m2.tolist()[:4]); print("   rstd", r1.tolist()[:4], r2.tolist()[:4])
run(1, 1, 2, 8, 8, 24, 16, 32)
run(1, 1, 2, 8, 8, 8, 8, 8)
run(1, 8, 2, 8, 8, 24, 16, 32)
run(0, 1, 2, 8, 8, 24, 16, 32)
run(1, 1, 3, 16, 40, 12, 8, 16)
run(1, 2, 3, 16, 40, 12, 8, 16)
