"""Value checks on the workload bench.py times: DC3D(st_dram_ref) at full channel widths on 128^3 chunks, through the fused
engine, in the memory mode ('tight': every large activation lazy, upsampled-input stages in slices of samples) and on the
kernels (Winograd-(z,y) forward / backward-data with 4 x-boxes per row, lazy backward-weights over 128^3 volumes, sample
offsets beyond 2^32 elements, statistics finalised over slices) that the 64 x 128^3 step runs.

  (a) anchor: ONE chunk of 128^3 against the oracle on the CPU (reference dram/models.py:120-147, dram/parts.py:177-187):
      output and every parameter gradient with GroupNorm(1, C) ('ln'); with BatchNorm the output against the oracle at
      2 x 64^3 and the whole step against the per-op path at 1 x 128^3;
  (b) scale: the same chunk replicated to the benchmark's 64 chunks, engine in 'tight' mode: every sample's output equals
      sample 0's = the N = 1 output, every parameter gradient is 64 x the N = 1 gradient -- with BatchNorm too (the
      statistics of 64 identical chunks are those of one), which is the mode bench.py runs;
  (c) what ran: the library's launch counters show the (z,y) forward kernel and the lazy backward-weights kernel.
"""
import numpy as np
import pytest
import torch

from oracle import dram_oracle as O
from dram_amd.configs import ST_DRAM_REF_MODEL

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def _model(norm, seed=0, away_from_zero=False):
    """DC3D(st_dram_ref) with HeNorm weights and non-trivial affine norm parameters.  `away_from_zero`: biases of +-3 sigma,
    so that no pre-activation lies within rounding of the ReLU threshold when two device paths are compared with each other
    (tests/test_gpu_engine.py explains the mask-flip mechanism)."""
    import models
    torch.manual_seed(seed)
    m = models.DC3D(**ST_DRAM_REF_MODEL, norm_method=norm)
    m.init(models.HeNorm(mode="fan_in"))
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, (torch.nn.BatchNorm3d, torch.nn.GroupNorm)) and mod.weight is not None:
                mod.weight.copy_(1.0 + 0.3 * torch.randn(mod.weight.shape, generator=g))
                if away_from_zero:
                    sign = torch.where(torch.arange(mod.bias.numel()) % 3 == 2, -1.0, 1.0)
                    mod.bias.copy_(3.0 * sign * mod.weight.abs())
                else:
                    mod.bias.copy_(0.2 * torch.randn(mod.bias.shape, generator=g))
    return m


def _chunk(n, s, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(n, 1, s, s, s, generator=g)
    gout = (0.5 + torch.rand(n, 1, s, s, s, generator=g)) / (s ** 3)       # positive: no cancelling sums in the last layers
    return x, gout


def _device_step(model, x, gout, fused=True):
    from dram_amd import functional as HF
    model.fused = fused
    for p in model.parameters():
        p.grad = None
    before = HF.conv_launch_counts()
    out, same = model(x)
    assert out is same
    (out * gout).sum().backward()
    torch.cuda.synchronize()
    delta = [a - b for a, b in zip(HF.conv_launch_counts(), before)]
    return out.detach(), {k: p.grad.detach().clone() for k, p in model.named_parameters()}, delta


class _ConvTap:
    """Records, for every 3x3x3 / 1x1x1 convolution the oracle runs, its input tensor and the gradient of its output (keyed by
    the weight tensor), so that individual weight-gradient entries can be recomputed EXACTLY (fp64 dot products over the
    volume).  Needed because torch's fp32 CPU convolution -- the oracle's arithmetic -- is itself off by up to 3e-3 of
    max|dW| on the backward-weights of 128^3 volumes (sums of 2 M products; scripts/diag_wgrad_accuracy_128.py measures it
    against exact fp64 values: torch-CPU 2e-4 on sampled entries, the HIP kernels 1e-7)."""

    def __init__(self):
        self.x, self.dy, self._orig = {}, {}, O.conv3d

    def __enter__(self):
        def conv3d(x, w, bias=None, pad=1):
            y = self._orig(x, w, bias, pad)
            if y.requires_grad:
                self.x[id(w)] = x.detach()
                y.register_hook(lambda g, k=id(w): self.dy.__setitem__(k, g.detach()))
            return y
        O.conv3d = conv3d
        return self

    def __exit__(self, *exc):
        O.conv3d = self._orig

    def exact_entries(self, w, n, rng):
        """[(index, exact fp64 value)] for `n` random entries of dW, from the oracle's own operands."""
        import torch.nn.functional as F
        x, dy = self.x[id(w)], self.dy[id(w)]
        k = w.shape[-1]
        D, H, W = x.shape[2:]
        out = []
        for _ in range(n):
            co, ci = int(rng.integers(w.shape[0])), int(rng.integers(w.shape[1]))
            kz, ky, kx = (int(rng.integers(k)) for _ in range(3))
            xp = F.pad(x[:, ci].double(), (k // 2,) * 6)
            val = (dy[:, co].double() * xp[:, kz:kz + D, ky:ky + H, kx:kx + W]).sum().item()
            out.append(((co, ci, kz, ky, kx), val))
        return out


def _oracle_step(model_cpu_sd, x, gout, norm, want_grads=True, tap=None):
    params, buffers = O.split_state_dict({k: v.clone() for k, v in model_cpu_sd.items()})
    for p in params.values():
        p.requires_grad_(want_grads)
    with torch.set_grad_enabled(want_grads):
        out = O.dc3d_forward(ST_DRAM_REF_MODEL, params, buffers, x, training=True, norm_method=norm, use_checkpoint=False)
        if want_grads:
            (out * gout).sum().backward()
    if tap is not None:
        return out.detach(), params
    return out.detach(), ({k: p.grad for k, p in params.items()} if want_grads else None)


def test_full_width_128_groupnorm_matches_the_oracle():
    """(a) 1 x 128^3, 'ln', fused engine vs the CPU oracle: output <= 1e-4; norm parameters and the head <= 5e-4 against the
    oracle's gradients; every conv weight gradient <= 1e-4 of max|dW| on 24 random entries against their EXACT values (fp64
    sums over the volume of the oracle's operands), and <= 1e-2 as a whole tensor against the oracle's fp32 gradient (the bound
    is torch-CPU's own error at this size, see _ConvTap)."""
    from dram_amd import functional as HF
    torch.set_num_threads(16)
    model = _model("ln")
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    x, gout = _chunk(1, 128, 21)
    with _ConvTap() as tap:
        ref_out, params = _oracle_step(sd, x, gout, "ln", tap=tap)
    model = model.to(DEV).train()
    out, grads, delta = _device_step(model, x.to(DEV), gout.to(DEV))
    assert delta[HF.K3_FWD_WZY] >= 16, delta           # the (z,y) kernel on the 128^3 / 64^3 / 32^3 levels, both directions
    # backward-weights: the (z,y) kernel ITSELF on every layer but the first (13 launches; the 16^3 level is 16 wide), counted
    # on its own -- a silent change of wgrad_plan would otherwise keep this green on the slower z-only kernel
    assert delta[HF.K3_WGRAD_WZY] >= 13 and delta[HF.K3_WGRAD_WZ_LAZY] + delta[HF.K3_WGRAD_WZ] == 0, delta
    assert _rel(out, ref_out) <= 1e-4, _rel(out, ref_out)
    rng = np.random.default_rng(0)
    worst_exact = worst_small = worst_whole = 0.0
    for k, p in params.items():
        got = grads[k].double().cpu()
        if p.dim() == 5:
            scale = p.grad.abs().max().item()
            for idx, val in tap.exact_entries(p, 24, rng):
                err = abs(got[idx].item() - val) / scale
                worst_exact = max(worst_exact, err)
                assert err <= 1e-4, (k, idx, got[idx].item(), val, scale)
            e = _rel(got, p.grad)
            worst_whole = max(worst_whole, e)
            assert e <= 1e-2, (k, e)
        else:
            e = _rel(got, p.grad)
            worst_small = max(worst_small, e)
            assert e <= 5e-4, (k, e)
    print(f"\nDC3D(st_dram_ref, ln) 1x128^3, engine vs oracle: out {_rel(out, ref_out):.2e}; conv weight gradients vs exact fp64 "
          f"entries {worst_exact:.1e}, vs torch-CPU fp32 {worst_whole:.1e}; norm / bias gradients {worst_small:.1e}")


def test_full_width_batchnorm_matches_the_oracle_and_the_per_op_path():
    """(a) 'bn': output vs the CPU oracle at 2 x 64^3 (<= 1e-4); at 1 x 128^3 the fused step equals the per-op step (the
    path the reference's block goldens pin) on outputs, gradients and BatchNorm buffers to 1e-4."""
    torch.set_num_threads(16)
    model = _model("bn", seed=3)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    x, gout = _chunk(2, 64, 22)
    ref_out, _ = _oracle_step(sd, x, gout, "bn", want_grads=False)
    model = model.to(DEV).train()
    out, _, _ = _device_step(model, x.to(DEV), gout.to(DEV))
    assert _rel(out, ref_out) <= 1e-4, _rel(out, ref_out)
    del model
    # fused vs per-op at the benchmark's chunk size, pre-activations kept away from the ReLU threshold
    model = _model("bn", seed=4, away_from_zero=True).to(DEV).train()
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    x, gout = _chunk(1, 128, 23)
    x, gout = x.to(DEV), gout.to(DEV)
    o_ref, g_ref, _ = _device_step(model, x, gout, fused=False)
    buf_ref = {k: v.clone() for k, v in model.state_dict().items() if "running" in k or "num_batches" in k}
    model.load_state_dict(sd0)
    o_got, g_got, delta = _device_step(model, x, gout, fused=True)
    assert _rel(o_got, o_ref) <= 2e-5, _rel(o_got, o_ref)
    errs = {k: _rel(g_got[k], g_ref[k]) for k in g_ref}
    print(f"\nDC3D(st_dram_ref, bn) 1x128^3, engine vs per-op: out {_rel(o_got, o_ref):.2e}, worst gradient {max(errs.values()):.2e}")
    bad = {k: v for k, v in errs.items() if v > 1e-4}
    assert not bad, bad
    for k, v in buf_ref.items():
        assert _rel(model.state_dict()[k].double(), v.double()) <= 1e-5, k


@pytest.mark.parametrize("norm", ["bn", "ln"])
def test_benchmark_batch_of_64_replicated_chunks(norm, monkeypatch):
    """(b) + (c): 64 copies of one 128^3 chunk through the engine exactly as bench.py runs it -- one batch, memory mode
    'tight' (chosen by the engine itself for this size), the widest stage in slices of samples, lazy backward-weights --
    must give 64 equal outputs, equal to the N = 1 run's, and 64 x its parameter gradients."""
    from dram_amd import engine
    from dram_amd import functional as HF
    torch.cuda.empty_cache()
    free, total = torch.cuda.mem_get_info()
    if free < 262 * 2 ** 30:
        pytest.skip(f"needs the 251 GB the benchmark step takes; {free / 2 ** 30:.0f} GB free")
    N = 64
    model = _model(norm, seed=5, away_from_zero=True).to(DEV).train()
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    x1, g1 = _chunk(1, 128, 24)
    x1, g1 = x1.to(DEV), g1.to(DEV)
    o1, grads1, _ = _device_step(model, x1, g1)
    plan1 = engine.LAST_PLAN
    model.load_state_dict(sd0)
    x = x1.expand(N, -1, -1, -1, -1).contiguous()
    gout = g1.expand(N, -1, -1, -1, -1).contiguous()
    assert x.numel() * 64 > 2 ** 32                       # a 64-channel activation of this batch: offsets beyond 2^32 elements
    torch.cuda.reset_peak_memory_stats()
    o, grads, delta = _device_step(model, x, gout)
    plan = engine.LAST_PLAN
    peak = torch.cuda.max_memory_allocated() / 2 ** 30
    # (c) the mode and the kernels of the benchmark
    assert plan.mode == "tight" and plan.sliced_stages >= 1 and plan1.sliced_stages == 0, (plan.mode, plan.sliced_stages)
    assert delta[HF.K3_FWD_WZY] >= 16 and delta[HF.K3_WGRAD_WZY] >= 13, delta      # ((z,y) backward-weights counted on its own)
    assert delta[HF.K3_WGRAD_WZ_LAZY] + delta[HF.K3_WGRAD_WZ] == 0, delta
    # (b) every sample equals sample 0 equals the single-chunk run
    worst_out = max(_rel(o[n], o[0]) for n in range(1, N))
    assert worst_out <= 1e-6, worst_out
    assert _rel(o[0], o1[0]) <= 2e-5, _rel(o[0], o1[0])
    errs = {k: _rel(grads[k], N * grads1[k]) for k in grads1}
    print(f"\n{norm}: 64 x 128^3 as one batch ('tight', {plan.sliced_stages} sliced stage(s), peak {peak:.0f} GB): samples differ by "
          f"{worst_out:.1e}, out vs N=1 {_rel(o[0], o1[0]):.1e}, worst gradient vs 64 x N=1 {max(errs.values()):.1e}")
    bad = {k: v for k, v in errs.items() if v > 1e-4}
    assert not bad, bad
    del o, grads, x, gout
    torch.cuda.empty_cache()


def test_full_width_batchnorm_distinct_chunks_sliced_vs_per_op(monkeypatch):
    """The hole the replicated-chunk test leaves (round-3 review): with 64 IDENTICAL samples any mis-pairing of per-sample
    data -- statistics partials of the wrong slice, a sample offset that lands on sample 0, lazy coefficients of another
    sample -- still gives the right answer.  Here 8 DISTINCT 128^3 chunks go through DC3D(st_dram_ref, 'bn') at full width,
    three ways:
      (p) the per-op path (one autograd node per op of the reference: what the reference's block goldens pin);
      (w) the fused engine, every activated tensor lazy (per-sample coefficients on load in forward, backward-weights,
          pool, upsample, head), no upsampled tensor kept, whole batch per launch;
      (s) the same with every up-stage forced into ragged slices of 3 + 3 + 2 samples (BatchNorm statistics finalised once
          over all slices' partials, backward-weights summed over slices, d(upsampled) never whole).
    (s) vs (w): every gradient to 5e-5 -- the only difference is the order in which three slices' dW are summed (measured:
    2e-5 on us_modules.2's first filter, 3e-6 on the other two sliced filters, 0 elsewhere).  (w) and (s) vs (p): outputs to
    2e-5, every gradient and BatchNorm buffer to 1e-4 -- except the norm parameters of us_modules.0's second stage at 2e-4: its
    gradient reaches it through four BatchNorm backwards whose (g - mean g) cancels strongly under the positive test gradient,
    and the two paths' statistics differ in the last bit (measured 1.00e-4 / 7.5e-5 there, identical for (w) and (s): the
    difference is between summation trees of the statistics, not between samples or slices).  Biases away from the ReLU
    threshold; the (z,y) kernels counted on their own."""
    from dram_amd import engine
    from dram_amd import functional as HF
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    if free < 120 * 2 ** 30:
        pytest.skip(f"needs ~100 GB (the per-op path keeps 3.5 KB per voxel); {free / 2 ** 30:.0f} GB free")
    N = 8
    model = _model("bn", seed=6, away_from_zero=True).to(DEV).train()
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    x, gout = _chunk(N, 128, 25)
    assert float((x[0] - x[1]).abs().max()) > 0.5                  # the chunks differ
    x, gout = x.to(DEV), gout.to(DEV)
    o_ref, g_ref, _ = _device_step(model, x, gout, fused=False)
    buf_ref = {k: v.clone() for k, v in model.state_dict().items() if "running" in k or "num_batches" in k}
    torch.cuda.empty_cache()
    monkeypatch.setattr(engine, "MEMORY_MODE", "manual")
    monkeypatch.setattr(engine, "MATERIALISE_BELOW", 0.0)
    monkeypatch.setattr(engine, "KEEP_UPSAMPLED_BELOW", 0.0)
    # (w) whole batch per launch
    model.load_state_dict(sd0)
    monkeypatch.setattr(engine, "SLICE_UPSAMPLED_ABOVE", 1.0)
    o_w, g_w, _ = _device_step(model, x, gout, fused=True)
    assert engine.LAST_PLAN.sliced_stages == 0
    # (s) slices of 3 + 3 + 2
    model.load_state_dict(sd0)
    slices = engine._slices

    def three_at_a_time(inp, n, budget):
        if not isinstance(inp, engine.Upsampled):
            return slices(inp, n, budget)
        per_sample = 4 * inp.src.raw.shape[1] * inp.size[0] * inp.size[1] * inp.size[2]
        return slices(inp, n, 3 * per_sample + 1)
    monkeypatch.setattr(engine, "_slices", three_at_a_time)
    o_got, g_got, delta = _device_step(model, x, gout, fused=True)
    plan = engine.LAST_PLAN
    assert plan.sliced_stages == 3, plan.sliced_stages              # us0, us1, us2 each as (0,3) (3,6) (6,8)
    # 11 forward + 10 backward-data launches on the (z,y) kernel, the up-stages' first convs three times each
    assert delta[HF.K3_FWD_WZY] >= 16 + 12, delta
    assert delta[HF.K3_WGRAD_WZY] >= 13 + 6 and delta[HF.K3_WGRAD_WZ_LAZY] + delta[HF.K3_WGRAD_WZ] == 0, delta
    # (s) vs (w)
    assert torch.equal(o_got, o_w)
    sw = {k: _rel(g_got[k], g_w[k]) for k in g_w}
    assert max(sw.values()) <= 5e-5, {k: v for k, v in sw.items() if v > 5e-5}
    # (w), (s) vs (p)
    worst_sample = max(_rel(o_got[n], o_ref[n]) for n in range(N))
    assert worst_sample <= 2e-5, worst_sample
    for name, g in (("whole", g_w), ("sliced", g_got)):
        errs = {k: _rel(g[k], g_ref[k]) for k in g_ref}
        print(f"\nDC3D(st_dram_ref, bn) 8 distinct 128^3 chunks, lazy, {name} vs per-op: worst sample out {worst_sample:.2e}, "
              f"worst gradient {max(errs.values()):.2e} ({max(errs, key=errs.get)}); sliced vs whole {max(sw.values()):.2e}")
        bad = {k: v for k, v in errs.items() if v > (2e-4 if k.startswith("us_modules.0.conv_blocks.1.1.") else 1e-4)}
        assert not bad, (name, bad)
    for k, v in buf_ref.items():
        assert _rel(model.state_dict()[k].double(), v.double()) <= 1e-5, k
