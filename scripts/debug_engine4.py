import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bodyct-dram_amd"), os.path.join(ROOT, "tests")]
import torch
import models
from dram_amd import engine
import dram_amd.engine as E
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
# per-op: d(act) and d(raw) of every stage
dact, draw, names = {}, {}, []
hooks = []
for bn, blk in list(model.named_modules()):
    if hasattr(blk, "conv_blocks") and not isinstance(blk, torch.nn.Sequential):
        for j, seq in enumerate(blk.conv_blocks):
            nm = f"{bn}.{j}"
            names.append(nm)
            def fh_conv(mod, inp, out, nm=nm):
                out.register_hook(lambda gr, nm=nm: draw.__setitem__(nm, gr.detach().clone()))
            def fh_norm(mod, inp, out, nm=nm):
                out.register_hook(lambda gr, nm=nm: dact.__setitem__(nm, gr.detach().clone()))
            hooks += [seq[0].register_forward_hook(fh_conv), seq[1].register_forward_hook(fh_norm)]
model.fused = False
d0, _ = model(x)
(d0 * gout).sum().backward()
for h in hooks: h.remove()
ref = {k: p.grad.clone() for k, p in model.named_parameters()}
for p in model.parameters(): p.grad = None
record = []
out, _ = engine.forward(model, x, record)
stages = [it[1] for it in record if it[0] == "conv"]
ymap = {s.y.data_ptr(): nm for s, nm in zip(stages, names)}
real_call = E.call
def spy(name, *args):
    if name == "dram_norm_bwd":
        nm = ymap[args[1]]
        st = [s for s in stages if s.y.data_ptr() == args[1]][0]
        n = st.y.numel()
        buf = (ctypes.c_float * 0)
        gin = torch.empty_like(st.y)
        # args[0] = dy pointer: wrap as tensor via from_blob-like copy
        import numpy as np
        t = torch.empty_like(st.y)
        torch.cuda.synchronize()
        ctypes.cdll.LoadLibrary(torch.__path__[0] + "/lib/libamdhip64.so").hipMemcpy(ctypes.c_void_p(t.data_ptr()), ctypes.c_void_p(args[0]), ctypes.c_size_t(n * 4), 3)
        print(f"{nm:28s} d(act) fused-vs-perop {_rel(t, dact[nm]):.2e}", end="")
        real_call(name, *args)
        torch.cuda.synchronize()
        ctypes.cdll.LoadLibrary(torch.__path__[0] + "/lib/libamdhip64.so").hipMemcpy(ctypes.c_void_p(t.data_ptr()), ctypes.c_void_p(args[6]), ctypes.c_size_t(n * 4), 3)
        print(f"   d(raw) {_rel(t, draw[nm]):.2e}")
        return
    real_call(name, *args)
E.call = spy
grads, dx = engine.backward(model, record, gout * 1.0, True)
E.call = real_call
print("---- last stage inputs")
s = stages[-1]
y = s.y
N, C = y.shape[:2]
m = y.double().mean(dim=(1, 2, 3, 4)); v = y.double().var(dim=(1, 2, 3, 4), unbiased=False)
print("mean saved", s.mean.tolist(), "expected", m.tolist())
print("rstd saved", s.rstd.tolist(), "expected", (1 / torch.sqrt(v + 1e-5)).tolist())
gam, bet = s.norm.weight.double(), s.norm.bias.double()
a = (gam[None, :] / torch.sqrt(v + 1e-5)[:, None])
b = bet[None, :] - m[:, None] * a
c = s.coef.view(N, C, 2).double()
print("coef a err", (c[..., 0] - a).abs().max().item(), "b err", (c[..., 1] - b).abs().max().item())
print("kind", s.kind, "groups", s.groups, "batch_stats", s.batch_stats, "num_groups", s.norm.num_groups)
print("---- last stage: norm backward three ways")
from dram_amd import _lib
import torch.nn.functional as F
nm = names[-1]
yt = s.y.detach().clone().requires_grad_(True)
gam = s.norm.weight.detach().clone().requires_grad_(True); bet = s.norm.bias.detach().clone().requires_grad_(True)
act = torch.relu(F.group_norm(yt, 1, gam, bet, 1e-5))
act.backward(dact[nm])
print("torch autograd vs per-op d(raw):", _rel(draw[nm], yt.grad))
dx = torch.empty_like(s.y); dg = torch.empty_like(gam); db = torch.empty_like(bet)
Nn, Cc = s.y.shape[:2]; S = s.y.numel() // (Nn * Cc)
ws = torch.empty(_lib.lib.dram_norm_ws_bytes(Nn, Cc, S), dtype=torch.uint8, device="cuda")
_lib.call("dram_norm_bwd", dact[nm].contiguous().data_ptr(), s.y.data_ptr(), s.norm.weight.data_ptr(), s.mean.data_ptr(), s.rstd.data_ptr(),
          s.coef.data_ptr(), dx.data_ptr(), dg.data_ptr(), db.data_ptr(), s.kind, s.groups, 1, 1, Nn, Cc, S, ws.data_ptr(), ws.numel(),
          torch.cuda.current_stream().cuda_stream)
print("direct norm_bwd call vs torch:", _rel(dx, yt.grad), " dgamma", _rel(dg, gam.grad), " dbeta", _rel(db, bet.grad))
print("engine grads: dgamma", _rel(grads[s.norm.weight], gam.grad), "dbeta", _rel(grads[s.norm.bias], bet.grad))
print("per-op grads: dgamma", _rel(ref[nm.replace('.1','') + '.conv_blocks.1.1.weight'] if False else ref['us_modules.2.conv_blocks.1.1.weight'], gam.grad),
      "dbeta", _rel(ref['us_modules.2.conv_blocks.1.1.bias'], bet.grad))
print("---- op-level per-op norm_act on the same tensors")
from dram_amd import functional as HF
yt2 = s.y.detach().clone().requires_grad_(True)
g2 = s.norm.weight.detach().clone().requires_grad_(True); b2 = s.norm.bias.detach().clone().requires_grad_(True)
a2 = HF.norm_act(yt2, g2, b2, None, None, HF.NORM_GROUP, 1, True, 0.0, 1e-5, True)
print("fwd vs torch", _rel(a2, act))
a2.backward(dact[nm])
print("per-op op-level d(raw) vs torch", _rel(yt2.grad, yt.grad), "dgamma", _rel(g2.grad, gam.grad), "dbeta", _rel(b2.grad, bet.grad))
# the module itself
mod = s.norm
yt3 = s.y.detach().clone().requires_grad_(True)
mod.zero_grad()
a3 = mod(yt3, relu=True)
a3.backward(dact[nm])
print("module d(raw) vs torch", _rel(yt3.grad, yt.grad), "dbeta", _rel(mod.bias.grad, bet.grad))
# the conv stack as the per-op path runs it
blk = model.us_modules[2]
print("---- d(act) of the last stage against its closed form w[0,c] * gout")
wtop = model.top_layer.weight.detach().view(1, -1, 1, 1, 1)
expect = wtop * gout
print("per-op hook d(act) vs closed form", _rel(dact[nm], expect))
print("dbeta: torch", bet.grad.tolist()); print("dbeta: perop", ref['us_modules.2.conv_blocks.1.1.bias'].tolist()); print("dbeta: fused", grads[s.norm.bias].tolist())
print("---- pre-activations of the last stage, channel by channel")
model.fused = False
for p in model.parameters(): p.grad = None
d0b, _ = model(x)
nfn = d0b.grad_fn.next_functions[0][0]
xs, gamma_s, sm, sr, rc = nfn.saved_tensors
Nn, Cc = xs.shape[:2]
pre_p = torch.addcmul(rc.view(Nn, Cc, 2)[..., 1].view(Nn, Cc, 1, 1, 1), rc.view(Nn, Cc, 2)[..., 0].view(Nn, Cc, 1, 1, 1), xs)
cf = s.coef.view(Nn, Cc, 2)
pre_f = torch.addcmul(cf[..., 1].view(Nn, Cc, 1, 1, 1), cf[..., 0].view(Nn, Cc, 1, 1, 1), s.y)
for c in range(Cc):
    a, b = pre_p[:, c], pre_f[:, c]
    flips = ((a > 0) != (b > 0)).sum().item()
    print(f"ch {c}: flips {flips} of {a.numel()}  min|pre| per-op {a.abs().min().item():.3e} fused {b.abs().min().item():.3e}  "
          f"#|pre|<1e-5: {(a.abs() < 1e-5).sum().item()} / {(b.abs() < 1e-5).sum().item()}  coef a {rc.view(Nn,Cc,2)[0,c,0].item():.4f} b {rc.view(Nn,Cc,2)[0,c,1].item():.4f}"
          f"  y range [{xs[:, c].min().item():.4f}, {xs[:, c].max().item():.4f}] unique-ish {(xs[:, c] == 0).sum().item()} zeros")
