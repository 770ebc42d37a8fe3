"""Rounding error of the 3x3x3 conv kernels against an fp64 CPU convolution: the default kernels (Winograd F(2x2,3x3)
over (z,y) for forward / backward-data where it applies -- the shapes with 32-wide rows below -- and F(2,3) along z
otherwise and for backward-weights), the z-only kernels everywhere (DRAM_CONV_NO_WZY=1), the direct kernels
(DRAM_CONV_DIRECT=1, read once per process -> child processes), and torch's own fp32 CPU convolution on the same
data as the yardstick.  Errors are max |got - ref64| / max |ref64| and relative L2.
    python scripts/conv_accuracy.py"""
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def child():
    import torch
    import torch.nn.functional as F
    sys.path[:0] = [ROOT, os.path.join(ROOT, "bodyct-dram_amd")]
    from dram_amd import functional as HF
    g = torch.Generator().manual_seed(5)
    rows = []
    for (N, Ci, Co, S) in [(2, 64, 64, 24), (1, 192, 64, 20), (1, 256, 256, 12), (1, 64, 64, 32), (1, 192, 64, 32), (1, 256, 256, 32)]:
        x = torch.rand(N, Ci, S, S, S, generator=g)                      # post-ReLU-like, non-negative inputs
        w = torch.randn(Co, Ci, 3, 3, 3, generator=g) * (2.0 / (Ci * 27)) ** 0.5
        gy = torch.randn(N, Co, S, S, S, generator=g)
        x64, w64 = x.double().requires_grad_(True), w.double().requires_grad_(True)
        y64 = F.conv3d(x64, w64, padding=1)
        y64.backward(gy.double())
        x32, w32 = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        y32 = F.conv3d(x32, w32, padding=1)
        y32.backward(gy)
        xg, wg = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True)
        yg = HF.conv3d_k3(xg, wg)
        yg.backward(gy.cuda())

        def err(a, ref):
            a, ref = a.detach().cpu().double(), ref.detach().double()
            return ((a - ref).abs().max() / ref.abs().max()).item(), ((a - ref).norm() / ref.norm()).item()
        for name, hip, cpu32, ref in (("fwd", yg, y32, y64), ("dgrad", xg.grad, x32.grad, x64.grad), ("wgrad", wg.grad, w32.grad, w64.grad)):
            (m, l), (m32, l32) = err(hip, ref), err(cpu32, ref)
            kern = HF.conv_fwd_kernel_name((S, S, S), Co, Ci).split("_kernel")[0][len("conv3d_k3_"):] if name != "wgrad" else "wgrad"
            rows.append(f"[{N},{Ci}->{Co},{S}^3] {name:5s} ({kern:7s})  HIP max {m:.2e} L2 {l:.2e}   torch-CPU-fp32 max {m32:.2e} L2 {l32:.2e}")
    print("\n".join(rows))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        for label, env in (("default kernels: Winograd (z,y) for fwd / dgrad where it applies, z-only otherwise", {}),
                           ("Winograd F(2,3)-z kernels everywhere (DRAM_CONV_NO_WZY=1)", {"DRAM_CONV_NO_WZY": "1"}),
                           ("direct kernels (DRAM_CONV_DIRECT=1)", {"DRAM_CONV_DIRECT": "1"})):
            print("==", label, flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, **env), check=True)
