"""Stage-by-stage comparison (diagnostics): raw conv outputs and their gradients, fused engine vs per-op path."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bodyct-dram_amd"), os.path.join(ROOT, "tests")]
import torch
import models
from dram_amd import engine, functional as HF
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

# per-op path: conv outputs and their gradients through hooks
ys, gys = {}, {}
convs = {n: m for n, m in model.named_modules() if isinstance(m, torch.nn.Conv3d) and m.kernel_size == (3, 3, 3)}
hooks = []
for n, m in convs.items():
    def fwd_hook(mod, inp, out, n=n):
        ys[n] = out.detach().clone()
        out.register_hook(lambda gr, n=n: gys.__setitem__(n, gr.detach().clone()))
    hooks.append(m.register_forward_hook(fwd_hook))
model.fused = False
d0, _ = model(x)
(d0 * gout).sum().backward()
for h in hooks:
    h.remove()
ref_grads = {k: p.grad.clone() for k, p in model.named_parameters()}
for p in model.parameters():
    p.grad = None

# fused: run forward/backward by hand, keeping every stage's d(raw output)
record = []
out = engine.forward(model, x, record)
stages = [it[1] for it in record if it[0] == "conv"]
names = list(convs)
assert len(stages) == len(names)
print("out", _rel(out, d0))
for n, s in zip(names, stages):
    print(f"fwd  {n:40s} y {_rel(s.y, ys[n]):.2e}")
orig = HF.call if hasattr(HF, "call") else None
import dram_amd.engine as E
real_call = E.call
captured = {}
def spy(name, *args):
    real_call(name, *args)
    if name == "dram_norm_bwd":
        captured[args[1]] = None       # x pointer -> mark
E.call = spy
grads, dx = engine.backward(model, record, gout * 1.0, True)
E.call = real_call
# after backward, each stage's incoming gradient buffer was overwritten in place with d(raw): not kept; compare grads only
for k, p in model.named_parameters():
    e = _rel(grads[p], ref_grads[k])
    print(f"grad {k:45s} {e:.2e}")
