"""Diagnostic: per-parameter gradient error of the HIP path vs the fp64 oracle, next to the reference's own fp32 error."""
import os, sys
import numpy as np, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bodyct-dram_amd"), os.path.join(ROOT, "tests")]
import models
from oracle import dram_oracle as O
from dram_amd.configs import SLIM
from test_gpu_parity import _fp64_oracle_grads, rel_err, _sub

z = np.load(os.path.join(ROOT, "tests/golden/dc3d_slim.npz"))
for tag, norm in [("slim_bn", "bn"), ("slim_ln", "ln")]:
    model = models.DC3D(**SLIM, norm_method=norm)
    sd = _sub(z, tag + "/sd/")
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model = model.cuda().train()
    x = torch.from_numpy(z[tag + "/x"]).cuda()
    d0, _ = model(x)
    (d0 * torch.from_numpy(z[tag + "/gout"]).cuda()).sum().backward()
    g64 = _fp64_oracle_grads(SLIM, sd, z[tag + "/x"], z[tag + "/gout"], norm)
    print(tag, "out err", rel_err(d0, z[tag + "/train_out"]))
    for k, p in model.named_parameters():
        key = tag + "/grad/" + k
        e_hip = max(rel_err(p.grad, g64[k]))
        e_ref = max(rel_err(z[key], g64[k])) if key in z.files else float("nan")
        print(f"  {k:45s} hip-vs-f64 {e_hip:.2e}  ref32-vs-f64 {e_ref:.2e}")
