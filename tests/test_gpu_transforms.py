"""GPU parity of the on-device OneShot transforms against the reference's own classes run on CPU
tensors (tests/golden/transforms.npz, oracle/make_golden.py:gen_transforms): outputs bit-exact for the pure index
transforms and nearest resize, 1e-5 for trilinear; gradients through flip / rot90 / trilinear resize."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _close(got, ref, tol):
    ref = torch.as_tensor(ref)
    got = got.detach().cpu()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert (got - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())


def test_flip_and_rot90_match_reference(golden_dir):
    from dram_amd import transforms as T
    z = np.load(os.path.join(golden_dir, "transforms.npz"))
    n = len([k for k in z.files if k.startswith("pf/") and k.endswith("/cfg")])
    assert n == 7 + 18
    for i in range(n):
        cfg = z[f"pf/{i}/cfg"]
        axes = tuple(int(a) for a in cfg[2:2 + int(z[f"pf/{i}/naxes"])])
        t = T.Flip3DOneShot(flip_axis=axes) if cfg[0] == 0 else T.Rotate903DOneShot(rotate_axis=axes, rotate_times=int(cfg[1]))
        x = torch.from_numpy(z["x"]).cuda().requires_grad_(True)
        res = t({"#image": x, "meta": 1})
        assert res["meta"] == 1
        out = res["#image"]
        assert torch.equal(out.detach().cpu(), torch.from_numpy(z[f"pf/{i}/out"])), (i, cfg)
        (out * torch.from_numpy(z[f"pf/{i}/gout"]).cuda()).sum().backward()
        assert torch.equal(x.grad.cpu(), torch.from_numpy(z[f"pf/{i}/gin"])), (i, cfg)


def test_rescale_matches_reference(golden_dir):
    from dram_amd import transforms as T
    z = np.load(os.path.join(golden_dir, "transforms.npz"))
    n = len([k for k in z.files if k.startswith("rs/") and k.endswith("/mode")])
    for i in range(n):
        mode = "size" if int(z[f"rs/{i}/mode"]) == 0 else "factor"
        sf = tuple(z[f"rs/{i}/sf"]) if mode == "factor" else tuple(int(v) for v in z[f"rs/{i}/sf"])
        t = T.Rescale3DOneShot(None, sf, mode=mode)
        x = torch.from_numpy(z["x"]).cuda().requires_grad_(True)
        res = t({"#image": x, "#reference": torch.from_numpy(z["lab"]).cuda()})
        _close(res["#image"], z[f"rs/{i}/out"], 1e-5)
        assert torch.equal(res["#reference"].cpu(), torch.from_numpy(z[f"rs/{i}/lab"])), (i, mode, sf)
        (res["#image"] * torch.from_numpy(z[f"rs/{i}/gout"]).cuda()).sum().backward()
        _close(x.grad, z[f"rs/{i}/gin"], 1e-5)


def test_random_parameter_choice_and_protocol():
    import random
    from dram_amd import transforms as T
    random.seed(3)
    np.random.seed(3)
    f, r = T.Flip3DOneShot(), T.Rotate903DOneShot()
    assert 1 <= len(f.flip_axis) <= 3 and all(2 <= a <= 4 for a in f.flip_axis)
    assert len(r.rotate_axis) == 2 and 1 <= r.rotate_times <= 3
    s = T.Rescale3DOneShot([70, 80, 90], None, mode='size')
    assert all(v in (70, 80, 90) for v in s.scale_factor)
    x = torch.rand(1, 1, 6, 6, 6, device="cuda")
    y = r(f({"#image": x}))["#image"]
    assert y.numel() == x.numel() and abs(float(y.sum()) - float(x.sum())) < 1e-3
    with pytest.raises(NotImplementedError):
        s({"#other": x})
    assert T.Identity()({"a": 1}) == {"a": 1}


def test_rotate3dx_matches_reference(golden_dir):
    """Rotate3DXOneShot (affine_grid + grid_sample with their defaults) against the reference's class at three angles:
    output and the gradient w.r.t. the input (scatter adjoint through float atomics) at 1e-5."""
    from dram_amd import transforms as T
    z = np.load(os.path.join(golden_dir, "transforms.npz"))
    n = len([k for k in z.files if k.startswith("rx/") and k.endswith("/theta")])
    assert n == 3
    for i in range(n):
        t = T.Rotate3DXOneShot()
        t.theta = np.array([float(z[f"rx/{i}/theta"])])
        x = torch.from_numpy(z["x"]).cuda().requires_grad_(True)
        out = t({"#image": x})["#image"]
        _close(out, z[f"rx/{i}/out"], 1e-5)
        (out * torch.from_numpy(z[f"rx/{i}/gout"]).cuda()).sum().backward()
        _close(x.grad, z[f"rx/{i}/gin"], 1e-5)
    np.random.seed(4)
    assert 0.0 <= float(T.Rotate3DXOneShot().theta[0]) <= np.pi
