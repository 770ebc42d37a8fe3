import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bodyct-dram_amd"), os.path.join(ROOT, "tests")]
import torch
import models
from oracle import dram_oracle as O
from dram_amd.configs import SLIM
from test_gpu_engine import _rel, _run
norm, N, shape = sys.argv[1], int(sys.argv[2]), tuple(int(v) for v in sys.argv[3:6])
torch.manual_seed(7)
model = models.DC3D(**SLIM, norm_method=norm)
model.init(models.HeNorm(mode="fan_in"))
g = torch.Generator().manual_seed(8)
with torch.no_grad():
    for m in model.modules():
        if isinstance(m, (torch.nn.BatchNorm3d, torch.nn.GroupNorm)) and m.weight is not None:
            m.weight.copy_(1.0 + 0.3 * torch.randn(m.weight.shape, generator=g))
            m.bias.copy_(0.2 * torch.randn(m.bias.shape, generator=g))
x = torch.rand((N, 1) + shape, generator=g)
gout = (torch.randn((N, 1) + shape, generator=g) / x.numel())
params, buffers = O.split_state_dict({k: v.clone().double() for k, v in model.state_dict().items()})
for p in params.values(): p.requires_grad_(True)
out = O.dc3d_forward(SLIM, params, buffers, x.double(), training=True, norm_method=norm)
(out * gout.double()).sum().backward()
model = model.cuda().train()
sd0 = {k: v.clone() for k, v in model.state_dict().items()}
ref = _run(model, x.cuda(), gout.cuda(), False, False)
model.load_state_dict(sd0)
got = _run(model, x.cuda(), gout.cuda(), True, False)
print("out per-op", _rel(ref[0], out), "fused", _rel(got[0], out))
for k in params:
    print(f"{k:45s} per-op-vs-oracle64 {_rel(ref[1][k], params[k].grad):.2e}   fused-vs-oracle64 {_rel(got[1][k], params[k].grad):.2e}")
