"""CPU checks of the train-step plumbing: the fused loss formulation (oracle.fused_loss_math) against the
reference's golden loss vector, that the product loss refuses CPU tensors, and the data-parallel gradient all-reduce (a sum: see DataParallelTrainer) over gloo with 2 ranks."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_device_loss_matches_reference_golden(golden_dir):
    from dram_amd.train_step import Batch
    from oracle.dram_oracle import fused_loss_math
    z = np.load(os.path.join(golden_dir, "loss.npz"))
    t = lambda k: torch.from_numpy(z[k])
    batch = Batch(t("images"), t("lobes"), t("lesions"), list(z["ctss"]), {k: 1.0 / 6 for k in range(6)}, band_width=1e-2)
    dense = t("dense").clone().requires_grad_(True)
    reg, seg = fused_loss_math(dense, batch, smoothing=0.1)
    assert abs(reg.item() - float(z["reg"])) <= 1e-5 * max(1.0, abs(float(z["reg"])))
    assert abs(seg.item() - float(z["seg"])) <= 1e-5 * max(1.0, abs(float(z["seg"])))
    (2.0 * reg + 1.0 * seg).backward()
    ref = z["gdense"]
    assert np.abs(dense.grad.numpy() - ref).max() <= 1e-4 * np.abs(ref).max()


def test_synthetic_batch_shapes():
    from dram_amd.train_step import synthetic_batch
    b = synthetic_batch(7, 12, 3, torch.device("cpu"))
    assert b.images.shape == (7, 1, 12, 12, 12) and b.targets.shape == (7, 2) and len(b) == 7
    assert float((b.images * (1 - b.lobes)).abs().max()) == 0.0          # zero outside the lobe
    assert torch.equal(b.lesions, ((b.images > 0.7) & (b.lobes > 0)).float())
    m = b.micro(2, 5)
    assert len(m) == 3 and m.ctss == [2.0, 3.0, 4.0]
    assert (b.targets[:, 0] <= b.targets[:, 1]).all()


def _dp_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path[:0] = [ROOT, os.path.join(ROOT, "bodyct-dram_amd")]
    from dram_amd.train_step import DataParallelTrainer
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3))
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    tr = DataParallelTrainer(model, opt, bucket_mb=0)     # bucket_mb=0 -> one bucket per parameter
    assert tr.world == world and len(tr.buckets) == 4
    for i, p in enumerate(model.parameters()):
        p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
    tr.allreduce_gradients()
    ok = all(torch.allclose(p.grad, torch.full_like(p, 3.0 * (i + 1))) for i, p in enumerate(model.parameters()))
    # a parameter without a gradient on this rank still takes part (zeros)
    params = list(model.parameters())
    params[0].grad = None if rank == 0 else torch.ones_like(params[0])
    tr.allreduce_gradients()
    ok = ok and torch.allclose(params[0].grad, torch.full_like(params[0], 1.0))
    out[rank] = bool(ok)
    dist.destroy_process_group()


def test_gradient_allreduce_two_ranks_gloo():
    world = 2
    port = 29500 + (os.getpid() % 2000)
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_dp_worker, args=(world, port, out), nprocs=world, join=True)
        assert dict(out) == {0: True, 1: True}


def test_device_loss_two_outputs_matches_reference_golden(golden_dir):
    """DC3DATGeneric case: regression term + pseudo label from dense_outs, segmentation term on refined."""
    from dram_amd.train_step import Batch
    from oracle.dram_oracle import fused_loss_math
    z = np.load(os.path.join(golden_dir, "loss2.npz"))
    t = lambda k: torch.from_numpy(z[k])
    batch = Batch(t("images"), t("lobes"), t("lesions"), list(z["ctss"]), {k: 1.0 / 6 for k in range(6)}, band_width=1e-2)
    dense, refined = t("dense").clone().requires_grad_(True), t("refined").clone().requires_grad_(True)
    reg, seg = fused_loss_math(dense, batch, refined=refined, smoothing=0.1)
    assert abs(reg.item() - float(z["reg"])) <= 1e-5 * max(1.0, abs(float(z["reg"])))
    assert abs(seg.item() - float(z["seg"])) <= 1e-5 * max(1.0, abs(float(z["seg"])))
    (2.0 * reg + 1.0 * seg).backward()
    assert np.abs(dense.grad.numpy() - z["gdense"]).max() <= 1e-4 * np.abs(z["gdense"]).max()
    assert np.abs(refined.grad.numpy() - z["grefined"]).max() <= 1e-4 * np.abs(z["grefined"]).max()


def test_device_loss_refuses_cpu_tensors(golden_dir):
    """No CPU / PyTorch fallback in the product path: the device loss raises on host tensors like every other op."""
    import pytest
    from dram_amd.train_step import Batch, DeviceIntRegRefineLoss
    z = np.load(os.path.join(golden_dir, "loss.npz"))
    t = lambda k: torch.from_numpy(z[k])
    batch = Batch(t("images"), t("lobes"), t("lesions"), list(z["ctss"]), {k: 1.0 / 6 for k in range(6)}, band_width=1e-2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        DeviceIntRegRefineLoss(1e-2, 0.1)(t("dense"), batch)
