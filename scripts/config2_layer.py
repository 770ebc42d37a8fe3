"""BASELINE config 2: x = [4,64,128^3], Conv3d(64,64,3,p=1) + {GroupNorm(1,64), GroupNorm(64,64), BatchNorm3d(64)} +
ReLU, forward only (reference layer us_modules.2.conv_blocks.1).  Prints the layer time on the GPU (HIP events) and
of the oracle's torch-CPU layer on a [1,64,64^3] sample (host threads as in bench.py)."""
import os
import sys
import time

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bodyct-dram_amd")]
import parts  # noqa: E402
from dram_amd import functional as HF  # noqa: E402

N, C, S = 4, 64, 128
x = torch.rand(N, C, S, S, S, device="cuda")
w = torch.randn(C, C, 3, 3, 3, device="cuda") * (2.0 / (C * 27)) ** 0.5
vox = N * S ** 3


def timeit(fn, it=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


with torch.no_grad():
    tc = timeit(lambda: HF.conv3d_k3(x, w))
    print(f"conv 64->64 on [4,64,128^3]: {tc:.2f} ms = {54.0 * C * C * vox / tc / 1e9:.0f} TFLOP/s (direct-equivalent), "
          f"{2 * 4 * C * vox / tc / 1e6:.0f} GB/s of algorithmic traffic")
    for norm in ("ln", "in", "bn"):
        m = parts.normal_wrapper(norm, C).cuda().train()
        y = HF.conv3d_k3(x, w)
        tn = timeit(lambda: m(y, relu=True))
        tl = timeit(lambda: m(HF.conv3d_k3(x, w), relu=True))
        print(f"  + {norm:3s} + ReLU: norm pass {tn:.2f} ms ({3 * 4 * C * vox / tn / 1e6:.0f} GB/s for 2R+1W), whole layer {tl:.2f} ms "
              f"= {vox / tl / 1e3:.1f} M voxels/s")
threads = min(os.cpu_count() or 1, int(os.environ.get("DRAM_CPU_THREADS", "16")))
torch.set_num_threads(threads)
xc = torch.rand(1, C, 64, 64, 64)
wc = w.cpu()
gn = torch.nn.GroupNorm(1, C)
with torch.no_grad():
    f = lambda: torch.relu_(gn(torch.nn.functional.conv3d(xc, wc, padding=1)))
    f()
    t0 = time.perf_counter()
    for _ in range(3):
        f()
    tcpu = (time.perf_counter() - t0) / 3
print(f"CPU (torch, {threads} threads) conv + GroupNorm(1,64) + ReLU on [1,64,64^3]: {tcpu * 1e3:.0f} ms = {64 ** 3 / tcpu / 1e6:.2f} M voxels/s")
