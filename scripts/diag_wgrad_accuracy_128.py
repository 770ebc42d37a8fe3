"""Who is right at 128^3?  Backward-weights of the full-resolution layers of DC3D(st_dram_ref): the HIP kernel, torch's fp32
CPU convolution (the oracle's arithmetic) and EXACT fp64 values of sampled entries dW[co, ci, tap] = sum_v dy[co, v] x[ci, v + tap],
all on the same operands (taken from the per-op device run of the model with hooks).

    python scripts/diag_wgrad_accuracy_128.py [ln|bn]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "bodyct-dram_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch
import torch.nn.functional as F


def main():
    norm = sys.argv[1] if len(sys.argv) > 1 else "ln"
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_bench_shapes import _chunk, _model
    torch.set_num_threads(16)
    model = _model(norm).to("cuda:0").train()
    model.fused = False
    layers = {"ds0.1": model.ds_modules[0].conv_blocks[1][0], "us2.1": model.us_modules[2].conv_blocks[1][0],
              "us1.1": model.us_modules[1].conv_blocks[1][0]}
    cap = {}
    for name, conv in layers.items():
        conv.register_forward_hook(lambda m, i, o, name=name: cap.__setitem__(name + "/x", i[0].detach().cpu()))
        conv.register_full_backward_hook(lambda m, gi, go, name=name: cap.__setitem__(name + "/dy", go[0].detach().cpu()))
    x, gout = _chunk(1, 128, 21)
    out, _ = model(x.cuda())
    (out * gout.cuda()).sum().backward()
    torch.cuda.synchronize()
    rng = np.random.default_rng(0)
    for name, conv in layers.items():
        xa, dy = cap[name + "/x"], cap[name + "/dy"]
        hip = conv.weight.grad.detach().cpu().double()
        w = conv.weight.detach().cpu().clone().requires_grad_(True)
        F.conv3d(xa, w, padding=1).backward(dy)
        cpu32 = w.grad.double()
        Co, Ci = hip.shape[:2]
        xp = F.pad(xa[0].double(), (1, 1, 1, 1, 1, 1))
        D, H, W = xa.shape[2:]
        errs_h, errs_c, mags = [], [], []
        for _ in range(48):
            co, ci, t = int(rng.integers(Co)), int(rng.integers(Ci)), int(rng.integers(27))
            kz, ky, kx = t // 9, (t // 3) % 3, t % 3
            exact = (dy[0, co].double() * xp[ci, kz:kz + D, ky:ky + H, kx:kx + W]).sum().item()
            errs_h.append(abs(hip[co, ci, kz, ky, kx].item() - exact))
            errs_c.append(abs(cpu32[co, ci, kz, ky, kx].item() - exact))
            mags.append(abs(exact))
        scale = hip.abs().max().item()
        absdot = (dy[0].abs().double().sum(dim=(1, 2, 3)).max() * xa[0].double().mean()).item()
        print(f"{norm} {name} [{Ci}->{Co} @ {D}^3]: max|dW| {scale:.3e} (sum|dy| * mean x ~ {absdot:.3e}); vs exact fp64 on 48 entries: "
              f"HIP max err {max(errs_h):.2e} ({max(errs_h) / scale:.2e} of max|dW|), torch-CPU-fp32 max err {max(errs_c):.2e} "
              f"({max(errs_c) / scale:.2e}); HIP vs CPU whole tensor {((hip - cpu32).abs().max() / scale).item():.2e}", flush=True)


if __name__ == "__main__":
    main()
