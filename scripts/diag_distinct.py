"""Diagnostic for tests/test_gpu_bench_shapes.py::test_full_width_batchnorm_distinct_chunks_sliced_vs_per_op: which gradients
differ between the per-op path, the fused engine (whole batch, lazy) and the fused engine in slices, and by how much."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bodyct-dram_amd"), os.path.join(ROOT, "tests")]
import torch
from dram_amd import engine
import test_gpu_bench_shapes as T

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
S = int(sys.argv[2]) if len(sys.argv) > 2 else 128
model = T._model("bn", seed=6, away_from_zero=True).to("cuda").train()
sd0 = {k: v.clone() for k, v in model.state_dict().items()}
x, gout = T._chunk(N, S, 25)
x, gout = x.cuda(), gout.cuda()
o_ref, g_ref, _ = T._device_step(model, x, gout, fused=False)
res = {}
for mode in ("whole", "sliced"):
    model.load_state_dict(sd0)
    engine.MEMORY_MODE = "manual"
    engine.MATERIALISE_BELOW = 0.0
    engine.KEEP_UPSAMPLED_BELOW = 0.0
    if mode == "sliced":
        orig = engine._slices
        def three(inp, n, budget, orig=orig):
            if not isinstance(inp, engine.Upsampled):
                return orig(inp, n, budget)
            per = 4 * inp.src.raw.shape[1] * inp.size[0] * inp.size[1] * inp.size[2]
            return orig(inp, n, 3 * per + 1)
        engine._slices = three
    else:
        engine.SLICE_UPSAMPLED_ABOVE = 1.0
    o, g, _ = T._device_step(model, x, gout, fused=True)
    res[mode] = (o, g)
    print(mode, "sliced stages", engine.LAST_PLAN.sliced_stages, "out", T._rel(o, o_ref))
    errs = sorted(((T._rel(g[k], g_ref[k]), k) for k in g_ref), reverse=True)
    for e, k in errs[:6]:
        print(f"   {e:.2e} {k}")
errs = sorted(((T._rel(res['sliced'][1][k], res['whole'][1][k]), k) for k in g_ref), reverse=True)
print("sliced vs whole:")
for e, k in errs[:6]:
    print(f"   {e:.2e} {k}")
# the per-op path against itself with another seed of nothing: run it twice (determinism)
model.load_state_dict(sd0)
o2, g2, _ = T._device_step(model, x, gout, fused=False)
print("per-op twice:", max(T._rel(g2[k], g_ref[k]) for k in g_ref))
