"""Per-tensor comparison of the fused engine against the per-op path (diagnostics)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bodyct-dram_amd"), os.path.join(ROOT, "tests")]
import torch
import models
from dram_amd.configs import SLIM
from test_gpu_engine import _run, _rel

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
model = model.cuda().train()
sd0 = {k: v.clone() for k, v in model.state_dict().items()}
x = torch.rand((N, 1) + shape, generator=g).cuda()
gout = (torch.randn((N, 1) + shape, generator=g) / x.numel()).cuda()
ref = _run(model, x, gout, False, True)
model.load_state_dict(sd0)
got = _run(model, x, gout, True, True)
model.load_state_dict(sd0)
ref2 = _run(model, x, gout, False, True)
print("out", _rel(got[0], ref[0]), "dx", _rel(got[2], ref[2]), " per-op twice:", _rel(ref2[0], ref[0]))
for k in ref[1]:
    print(f"{k:45s} fused-vs-perop {_rel(got[1][k], ref[1][k]):.2e}   perop-vs-perop {_rel(ref2[1][k], ref[1][k]):.2e}")
