"""GPU parity of the less-travelled factory branches against the reference's own
outputs (tests/golden/misc.npz, oracle/make_golden.py:gen_misc): act_wrapper "prelu" inside a
ConvPoolBlock5d and per channel, pooling_dense_features 'global_avg' / 'global_max' / lobe-masked."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 1e-4


def check(got, ref, what, tol=TOL):
    ref = torch.as_tensor(np.asarray(ref)).double()
    got = got.detach().cpu().double()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    err = ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()
    assert err <= tol, (what, err)


def _sub(z, prefix):
    return {k[len(prefix):]: z[k] for k in z.files if k.startswith(prefix)}


def test_convpool_block_with_prelu(golden_dir):
    import parts
    z = np.load(os.path.join(golden_dir, "misc.npz"))
    tag = "convpool_prelu/"
    blk = parts.ConvPoolBlock5d([3, 4], [4, 6], 0, (3, 3), False, (1, 1), 2, 2, 0, dropout=0.0, norm_method="bn",
                                act_method="prelu")
    blk.load_state_dict({k: torch.from_numpy(v) for k, v in _sub(z, tag + "sd/").items()})
    blk = blk.cuda()
    x = torch.from_numpy(z[tag + "in0"]).cuda().requires_grad_(True)
    y, pooled = blk(x)
    check(y, z[tag + "out/y"], "y")
    check(pooled, z[tag + "out/pooled"], "pooled")
    t = lambda k: torch.from_numpy(z[tag + k]).cuda()
    ((y * t("gout/y")).sum() + (pooled * t("gout/pooled")).sum()).backward()
    check(x.grad, z[tag + "gin0"], "gin")
    for k, p in blk.named_parameters():
        check(p.grad, z[tag + "gparam/" + k], "gparam/" + k)


def test_prelu_per_channel(golden_dir):
    import parts
    z = np.load(os.path.join(golden_dir, "misc.npz"))
    tag = "prelu_c/"
    act = parts.act_wrapper("prelu", 5, 0.1)
    act.load_state_dict({k: torch.from_numpy(v) for k, v in _sub(z, tag + "sd/").items()})
    act = act.cuda()
    x = torch.from_numpy(z[tag + "in0"]).cuda().requires_grad_(True)
    y = act(x)
    check(y, z[tag + "out/y"], "y", 1e-6)
    (y * torch.from_numpy(z[tag + "gout/y"]).cuda()).sum().backward()
    check(x.grad, z[tag + "gin0"], "gin", 1e-6)
    check(act.weight.grad, z[tag + "gparam/weight"], "da", 1e-5)
    with pytest.raises(NotImplementedError):
        parts.act_wrapper("gelu")


@pytest.mark.parametrize("method", ["global_avg", "global_max", "avg"])
def test_pooling_dense_features(golden_dir, method):
    import models
    z = np.load(os.path.join(golden_dir, "misc.npz"))
    dense = torch.from_numpy(z["pool/dense"]).cuda().requires_grad_(True)
    lungs = torch.from_numpy(z["pool/lungs"]).cuda()
    out = models.pooling_dense_features(dense, lungs, method)
    check(out, z[f"pool/{method}/out"], method + " out", 1e-6)
    (out * torch.from_numpy(z[f"pool/{method}/gout"]).cuda()).sum().backward()
    if method == "global_max":      # ties: the gradient goes to the first maximum, bit for bit
        assert torch.equal(dense.grad.cpu(), torch.from_numpy(z[f"pool/{method}/gin"]))
    else:
        check(dense.grad, z[f"pool/{method}/gin"], method + " gin", 1e-6)


def test_pooling_dense_features_per_channel_mask():
    """`lungs.expand_as(dense_outs)` (reference models.py:45) also accepts a [B,C,D,H,W] mask: one mask per channel.
    Checked against the reference's formula evaluated in fp64 on the host."""
    import models
    g = torch.Generator().manual_seed(21)
    dense = torch.randn(2, 3, 5, 6, 7, generator=g)
    lungs = (torch.rand(2, 3, 5, 6, 7, generator=g) > 0.4).float()
    gout = torch.randn(2, 3, generator=g)
    d64 = dense.double().requires_grad_(True)
    ref = (d64 * lungs.double()).view(2, 3, -1).sum(-1) / lungs.double().view(2, 3, -1).sum(-1)   # models.py:45-47
    (ref * gout.double()).sum().backward()
    dg = dense.cuda().requires_grad_(True)
    out = models.pooling_dense_features(dg, lungs.cuda(), "avg")
    check(out, ref.detach(), "per-channel mask out", 1e-6)
    (out * gout.cuda()).sum().backward()
    check(dg.grad, d64.grad, "per-channel mask gin", 1e-6)
    with pytest.raises(ValueError):
        models.pooling_dense_features(dg, lungs.cuda()[:, :2], "avg")
