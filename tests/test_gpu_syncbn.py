"""Cross-rank BatchNorm ("sbn" under data parallelism): two processes share the GPU over gloo; each normalises its
half of a batch with HipSyncBatchNorm and must reproduce -- output, input gradient, parameter gradients, running
statistics -- what one process computes with HipBatchNorm3d on the whole batch."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _worker(rank, world, port, relu, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path[:0] = [ROOT, os.path.join(ROOT, "bodyct-dram_amd")]
    from dram_amd.modules import HipBatchNorm3d, HipSyncBatchNorm
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(7)
    N, C, shape = 6, 5, (4, 6, 8)
    x = torch.randn((N, C) + shape, generator=g) * 2.0 + 3.0
    gy = torch.randn((N, C) + shape, generator=g)
    w, b = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    lo, hi = (0, 2) if rank == 0 else (2, 6)            # uneven split: the combine must weight by count
    m = HipSyncBatchNorm(C).cuda().train()
    with torch.no_grad():
        m.weight.copy_(w); m.bias.copy_(b)
    xs = x[lo:hi].cuda().requires_grad_(True)
    y = m(xs, relu=relu)
    y.backward(gy[lo:hi].cuda())
    # parameter gradients: local sums, summed over ranks = the whole-batch gradient
    gw, gb = m.weight.grad.clone(), m.bias.grad.clone()
    dist.all_reduce(gw); dist.all_reduce(gb)
    ok = True
    if rank == 0:
        ref = HipBatchNorm3d(C).cuda().train()
        with torch.no_grad():
            ref.weight.copy_(w); ref.bias.copy_(b)
        xr = x.cuda().requires_grad_(True)
        yr = ref(xr, relu=relu)
        yr.backward(gy.cuda())
        close = lambda a, bb, tol=2e-5: ((a - bb).abs().max() <= tol * bb.abs().max().clamp_min(1e-6)).item()
        ok = close(y, yr[lo:hi].detach()) and close(xs.grad, xr.grad[lo:hi]) and close(gw, ref.weight.grad) and \
            close(gb, ref.bias.grad) and close(m.running_mean, ref.running_mean) and close(m.running_var, ref.running_var) \
            and int(m.num_batches_tracked) == 1
    out[rank] = bool(ok)
    dist.destroy_process_group()


@pytest.mark.parametrize("relu", [False, True])
def test_sync_batchnorm_two_ranks_match_whole_batch(relu):
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    port = 29650 + int(relu)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, relu, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert out.get(0) is True and out.get(1) is True


def test_sync_batchnorm_without_group_is_batchnorm():
    from dram_amd.modules import HipBatchNorm3d, HipSyncBatchNorm
    import parts
    assert isinstance(parts.normal_wrapper("sbn", 4), HipSyncBatchNorm)
    x = torch.randn(2, 4, 3, 5, 6, device="cuda")
    a, b = HipSyncBatchNorm(4).cuda().train(), HipBatchNorm3d(4).cuda().train()
    assert torch.equal(a(x), b(x))


def _engine_worker(rank, world, port, out):
    """Slim DC3D with "sbn" through the FUSED ENGINE on this rank's share of a batch (statistics combined over the two
    ranks inside the engine) against one process running the whole batch with "bn" through the same engine."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path[:0] = [ROOT, os.path.join(ROOT, "bodyct-dram_amd")]
    import models
    from dram_amd import engine
    from dram_amd.configs import SLIM
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def build(norm):
        torch.manual_seed(11)
        m = models.DC3D(**SLIM, norm_method=norm)
        m.init(models.HeNorm(mode="fan_in"))
        g = torch.Generator().manual_seed(12)
        with torch.no_grad():
            for mod in m.modules():
                if isinstance(mod, torch.nn.BatchNorm3d) and mod.weight is not None:
                    mod.weight.copy_(torch.rand(mod.weight.shape, generator=g) + 0.5)
                    mod.bias.copy_((torch.randint(0, 2, mod.bias.shape, generator=g).float() * 2 - 1) * 3.0)
            # channel 0 of the first stage: variance ~1e-9, far below eps = 1e-5 (its statistics must come from the partials'
            # M2, not from inverting rstd); momentum 1 on that layer so that running_var shows the batch variance itself
            m.ds_modules[0].conv_blocks[0][0].weight[0] *= 3e-5
            m.ds_modules[0].conv_blocks[0][1].momentum = 1.0
        return m.cuda().train()

    g = torch.Generator().manual_seed(13)
    N, shape = 4, (16, 16, 16)
    x = torch.randn((N, 1) + shape, generator=g)
    gout = torch.randn((N, 1) + shape, generator=g)
    lo, hi = (0, 1) if rank == 0 else (1, 4)            # uneven split: the combine must weight by count
    m = build("sbn")
    assert engine.supports(m) and m.fused
    before = engine.LAST_PLAN
    d0, _ = m(x[lo:hi].cuda())
    assert engine.LAST_PLAN is not before               # the engine ran (not the per-op path)
    (d0 * gout[lo:hi].cuda()).sum().backward()
    grads = {k: p.grad.clone() for k, p in m.named_parameters()}
    for v in grads.values():
        dist.all_reduce(v)                               # parameter gradients: local sums
    ok = True
    if rank == 0:
        ref = build("bn")
        r0, _ = ref(x.cuda())
        (r0 * gout.cuda()).sum().backward()
        rel = lambda a, b: ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()
        errs = {"out": rel(d0.detach(), r0.detach()[lo:hi])}
        for k, p in ref.named_parameters():
            errs[k] = rel(grads[k], p.grad)
        for k, v in ref.state_dict().items():
            if "running" in k:
                errs[k] = rel(m.state_dict()[k], v)
        k0 = "ds_modules.0.conv_blocks.0.1.running_var"
        tiny_ref, tiny_got = float(ref.state_dict()[k0][0]), float(m.state_dict()[k0][0])
        assert 0.0 < tiny_ref < 1e-7, tiny_ref
        errs["running_var of the tiny-variance channel"] = abs(tiny_got - tiny_ref) / tiny_ref
        bad = {k: e for k, e in errs.items() if not e <= 2e-4}
        ok = not bad
        if bad:
            print("sbn engine mismatches:", bad, flush=True)
    out[rank] = bool(ok)
    dist.destroy_process_group()


def test_sync_batchnorm_on_the_fused_engine_two_ranks():
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    procs = [ctx.Process(target=_engine_worker, args=(r, 2, 29657, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert out.get(0) is True and out.get(1) is True
