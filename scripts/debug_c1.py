import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bodyct-dram_amd")]
import torch
import torch.nn.functional as F
from dram_amd import functional as HF
from dram_amd import _lib
dev = "cuda:0"
st = torch.cuda.current_stream().cuda_stream
p = lambda t: None if t is None else t.data_ptr()
for N, Co, D, H, W, bias in ((1, 32, 5, 6, 128, False), (2, 40, 4, 9, 100, True), (1, 32, 4, 8, 32, False), (2, 32, 5, 9, 33, False), (1, 40, 8, 16, 64, True), (3, 8, 6, 7, 10, False), (1, 32, 16, 64, 128, False), (1, 32, 128, 128, 128, False), (2, 32, 64, 128, 128, False)):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, 1, D, H, W, generator=g)
    w = torch.randn(Co, 1, 3, 3, 3, generator=g)
    b = torch.randn(Co, generator=g) if bias else None
    ref = F.conv3d(x.double(), w.double(), None if b is None else b.double(), padding=1)
    xd, wd = x.to(dev), w.to(dev)
    wt = HF._pack(wd, 0)
    guard = 1024
    ybuf = torch.full((N * Co * D * H * W + 2 * guard,), 7.0, device=dev)
    y = ybuf[guard:guard + N * Co * D * H * W].view(N, Co, D, H, W)
    nparts = _lib.lib.dram_conv3d_k3_stats_parts(1, Co, D, H, W)
    parts = torch.zeros(N * Co * nparts * 3, device=dev) if not bias else None
    _lib.call("dram_conv3d_k3_fwd_fused", p(xd), 1, None, 0, None, 0, None, 0, 0, 0, 0, 0, 0, 0, p(wt), p(b.to(dev)) if bias else None,
              p(y), p(parts), nparts if parts is not None else 0, N, Co, D, H, W, st)
    torch.cuda.synchronize()
    err = (y.cpu().double() - ref).abs()
    print(HF.conv_fwd_kernel_name((D, H, W), Co, 1, fused=True), (N, Co, D, H, W), "max err", err.max().item(), "guards intact",
          bool((ybuf[:guard] == 7).all() and (ybuf[-guard:] == 7).all()))
    if err.max() > 1e-4:
        bad = (err > 1e-4).nonzero()
        print("  first bad", bad[:5].tolist(), "n bad", len(bad), "of", err.numel())
        print("  per-channel bad counts", (err > 1e-4).sum(dim=(0, 2, 3, 4)).tolist())
    if parts is not None:
        pr = parts.view(N, Co, nparts, 3).cpu().double()
        cnt = pr[..., 2].sum(-1)
        mean = (pr[..., 0] * pr[..., 2]).sum(-1) / cnt
        m2 = (pr[..., 1] + pr[..., 2] * (pr[..., 0] - mean[..., None]) ** 2).sum(-1)
        print("  counts ok", bool((cnt == D * H * W).all()), "mean err", (mean - ref.mean(dim=(2, 3, 4))).abs().max().item(),
              "var rel err", ((m2 / cnt - ref.var(dim=(2, 3, 4), unbiased=False)).abs() / ref.var(dim=(2, 3, 4), unbiased=False)).max().item())
        y1, p1 = y.clone(), parts.clone()
        parts.zero_()
        _lib.call("dram_conv3d_k3_fwd_fused", p(xd), 1, None, 0, None, 0, None, 0, 0, 0, 0, 0, 0, 0, p(wt), None,
                  p(y), p(parts), nparts, N, Co, D, H, W, st)
        torch.cuda.synchronize()
        print("  rerun bitwise equal: y", bool((y1 == y).all()), "parts", bool((p1 == parts).all()))
