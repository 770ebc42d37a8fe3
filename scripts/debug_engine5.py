import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bodyct-dram_amd"), os.path.join(ROOT, "tests")]
import torch
import models
from dram_amd import engine
from dram_amd.configs import SLIM
from test_gpu_engine import _rel
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
x = torch.rand((N, 1) + shape, generator=g).cuda()
gout = (torch.randn((N, 1) + shape, generator=g) / x.numel()).cuda()
model.fused = False
d0, _ = model(x)
fn = d0.grad_fn
print(type(fn).__name__)
nfn = fn.next_functions[0][0]
print(type(nfn).__name__)
xs, gamma, save_mean, save_rstd, rowcoef = nfn.saved_tensors
print("saved before backward: mean", save_mean.tolist(), "rstd", save_rstd.tolist())
m = xs.double().mean(dim=(1, 2, 3, 4)); v = xs.double().var(dim=(1, 2, 3, 4), unbiased=False)
print("expected              mean", m.tolist(), "rstd", (1 / torch.sqrt(v + 1e-5)).tolist())
rc0 = rowcoef.clone(); sm0 = save_mean.clone(); sr0 = save_rstd.clone(); x0 = xs.clone()
ok = []
def hook(grad_inputs, grad_outputs=None):
    pass
(d0 * gout).sum().backward(retain_graph=True)
print("after backward: rowcoef changed", (rowcoef - rc0).abs().max().item(), "mean", (save_mean - sm0).abs().max().item(),
      "rstd", (save_rstd - sr0).abs().max().item(), "x", (xs - x0).abs().max().item())
print("cfg", nfn.cfg if hasattr(nfn, "cfg") else None)
