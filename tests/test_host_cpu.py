"""CPU-side checks of the host logic and of the C ABI surface (no kernel is launched)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "dram_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(?:int|size_t|const char\*)\s+(dram_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = m.group(2).strip()
        n = 0 if args in ("", "void") else len(args.split(","))
        out[m.group(1)] = n
    return out


def test_library_exports_every_declared_symbol():
    decl = _header_functions()
    assert len(decl) >= 29
    lib = ctypes.CDLL(os.path.join(ROOT, "bodyct-dram_amd", "libdram_hip.so"))
    for name in decl:
        assert hasattr(lib, name), f"libdram_hip.so does not export {name}"
    assert lib.dram_abi_version() == 1


def test_ctypes_signatures_match_header():
    from dram_amd import _lib
    decl = _header_functions()
    assert set(decl) == set(_lib.SIGNATURES), set(decl) ^ set(_lib.SIGNATURES)
    for name, nargs in decl.items():
        assert len(_lib.SIGNATURES[name][1]) == nargs, name


def test_argument_errors_are_reported_without_a_gpu():
    from dram_amd import _lib
    with pytest.raises(_lib.DramHipError, match="null pointer"):
        _lib.call("dram_conv3d_k3_fwd", None, None, None, None, 1, 1, 1, 4, 4, 4, None)
    with pytest.raises(_lib.DramHipError, match="not divisible"):
        _lib.call("dram_norm_fwd_train", 16, None, None, 16, 16, 16, 16, None, None, 0.1, 1e-5, 1, 3, 1, 2, 4, 8, 16, 1 << 20, None)
    assert _lib.lib.dram_conv3d_k3_wgrad_ws_bytes(4, 64, 64, 128, 128, 128) > 0


def test_kernel_choice_queries():
    """dram_conv3d_k3_fwd_choice / dram_conv3d_k3_wgrad_choice are pure functions of the shape (no GPU needed): the
    benchmark's layers (DC3D st_dram_ref at 128^3) get the kernels DESIGN.md names, and the two rules the old host-side
    copy of use_wzy had dropped are there."""
    from dram_amd import functional as HF
    S = (128, 128, 128)
    assert HF.conv_fwd_kernel_name(S, 64, 32, fused=True) == "conv3d_k3_fwd_wzy_kernel"           # ds0.1 forward
    assert HF.conv_fwd_kernel_name(S, 64, 192, fused=True) == "conv3d_k3_fwd_wzy_kernel"          # us2.0 forward
    # us2.0 backward-data: 64 -> 128 + 64 channels over two tensors (boundary at 128: a multiple of 32)
    assert HF.conv_fwd_kernel_name(S, 192, 64, dst_split=(128, 64, 128, 128, 128)) == "conv3d_k3_fwd_wzy_kernel"
    # a split at a channel that is not a multiple of 32 cannot take the (z,y) kernel (one destination tensor per wave)
    assert "fwd_wz_kernel" in HF.conv_fwd_kernel_name((8, 8, 32), 192, 64, dst_split=(112, 80, 8, 8, 32))
    # ds0.1 backward-data 64 -> 32: Cout % 64 != 0
    name = HF.conv_fwd_kernel_name(S, 32, 64)
    assert name in ("conv3d_k3_fwd_wz_kernel<32, 4, 1, false>", "conv3d_k3_fwd_wzy_kernel"), name
    # first layer
    assert HF.conv_fwd_kernel_name(S, 32, 1, fused=True) == "conv3d_k3_fwd_c1w_kernel"
    assert HF.conv_fwd_kernel_name((80, 80, 80), 32, 1, fused=True) == "conv3d_k3_fwd_c1_kernel"
    # the 16^3 level of 128^3 chunks and the reference's own 80^3 pyramid (80 / 40 / 20): 16-wide (z,y) boxes; 10^3: z-only
    assert HF.conv_fwd_kernel_name((16, 16, 16), 512, 256, fused=True) == "conv3d_k3_fwd_wzy16_kernel"
    for s_, co_, ci_ in ((80, 64, 192), (40, 128, 384), (20, 256, 768)):
        assert HF.conv_fwd_kernel_name((s_,) * 3, co_, ci_, fused=True) == "conv3d_k3_fwd_wzy16_kernel", s_
    assert HF.conv_fwd_kernel_name((10, 10, 10), 512, 256, fused=True) == "conv3d_k3_fwd_wz_kernel<10, 10, 2, true>"
    assert HF.conv_fwd_kernel_name((5, 7, 32), 128, 64) == "conv3d_k3_fwd_wz_kernel<32, 4, 2, false>"      # either box pads too much
    wz = HF.conv_wgrad_kernel_name(64, S, 64, 128, 64, lazy=True)
    assert wz in ("conv3d_k3_wgrad_wz_kernel<16, 2, 4, 2, true>", "conv3d_k3_wgrad_wzy_kernel<true>"), wz
    assert HF.conv_wgrad_kernel_name(64, S, 32, 1) == "conv3d_k3_wgrad_c1_kernel"
    assert "wgrad_wz" not in HF.conv_wgrad_kernel_name(2, (5, 7, 11), 16, 8)              # odd width: direct kernel
    assert HF.conv_wgrad_kernel_name(2, (5, 7, 11), 16, 8, lazy=True) == HF.conv_wgrad_kernel_name(2, (5, 7, 11), 16, 8)
    counts = HF.conv_launch_counts()
    assert len(counts) == HF.K3_KINDS and all(c >= 0 for c in counts)


def test_fwd_choice_depends_on_the_source_but_the_partial_count_does_not():
    """A (z,y)-shaped forward launch whose source the (z,y) kernel refuses (a base pointer off 16-byte alignment; a cropped
    skip whose window starts at x % 4 != 0 or whose rows are not multiples of 16 bytes) runs the z-only kernel ON THE SAME
    32x4x2 BOXES: dram_conv3d_k3_stats_parts(shape) stays what the launch writes (round-3 advisor finding: at 56x32x56 the
    z-only kernel would otherwise pick 8x16 boxes, 1568 partials against the 1792 the buffer was sized for)."""
    import ctypes
    from dram_amd import _lib

    def name(dhw, co, ci, **src):
        buf = ctypes.create_string_buffer(96)
        kind = _lib.lib.dram_conv3d_k3_fwd_choice_src(ci, co, *dhw, co, 0, 0, 0, 0, 1, src.get("c2", 0), src.get("d2", 0),
                                                      src.get("h2", 0), src.get("w2", 0), src.get("ox", 0), src.get("mis", 0),
                                                      buf, len(buf))
        assert kind >= 0
        return kind, buf.value.decode()

    dhw = (56, 32, 56)
    parts = _lib.lib.dram_conv3d_k3_stats_parts(64, 64, *dhw)
    assert parts == 2 * 8 * 28 * 4                                       # 32x4x2 boxes, 4 partials each
    assert name(dhw, 64, 64) == (2, "conv3d_k3_fwd_wzy_kernel")
    assert name(dhw, 64, 64, c2=32, d2=56, h2=32, w2=56, ox=0) == (2, "conv3d_k3_fwd_wzy_kernel")
    for src in (dict(mis=1), dict(c2=32, d2=58, h2=35, w2=62, ox=3), dict(c2=32, d2=56, h2=32, w2=58, ox=0)):
        assert name(dhw, 64, 64, **src) == (1, "conv3d_k3_fwd_wz_kernel<32, 4, 2, true>"), src
    # a shape the (z,y) kernel does not serve keeps the z-only kernel's own box choice
    assert name((56, 32, 56), 32, 64)[1] == "conv3d_k3_fwd_wz_kernel<8, 16, 1, true>"
    # a 16-wide (z,y) shape: its refused launches run the z-only kernel on 16x8 positions; the slot count covers both boxings
    # (20^3: 2 x 5 x 5 (z,y) boxes against 2 x 3 x 10 z-only ones)
    assert _lib.lib.dram_conv3d_k3_stats_parts(64, 64, 20, 20, 20) == 2 * 3 * 10 * 4
    assert name((20, 20, 20), 64, 64) == (2, "conv3d_k3_fwd_wzy16_kernel")
    assert name((20, 20, 20), 64, 64, mis=1) == (1, "conv3d_k3_fwd_wz_kernel<16, 8, 2, true>")


def test_cpu_tensors_fail_loudly():
    import parts
    blk = parts.ConvBlock5d([2, 3], [3, 4], 0, 3, False, 1, 0.0)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        blk(torch.zeros(1, 2, 4, 4, 4))


def test_normal_wrapper_and_act_wrapper():
    import parts
    assert isinstance(parts.normal_wrapper("bn", 4), torch.nn.BatchNorm3d)
    bnt = parts.normal_wrapper("bnt", 4)
    assert isinstance(bnt, torch.nn.BatchNorm3d) and not bnt.track_running_stats and bnt.affine
    assert not parts.normal_wrapper("bntna", 4).affine
    ln = parts.normal_wrapper("ln", 6)
    assert isinstance(ln, torch.nn.GroupNorm) and ln.num_groups == 1
    assert parts.normal_wrapper("in", 6).num_groups == 6
    assert not parts.normal_wrapper("lnna", 6).affine
    assert isinstance(parts.normal_wrapper(None, 4), parts.Identity)
    assert isinstance(parts.normal_wrapper("nope", 4), parts.Identity)
    assert isinstance(parts.act_wrapper("relu"), torch.nn.ReLU)
    with pytest.raises(NotImplementedError):
        parts.act_wrapper("tanh")
    for name in ("torch", "nn", "np", "F", "functools", "math", "checkpoint", "checkpoint_sequential"):
        assert hasattr(parts, name)   # models.py relies on the star import (reference parts.py:1-8)


def test_dc3d_structure_matches_reference(golden_dir):
    """Same seed -> same parameters as the reference's DC3D (checksums from the reference run),
    same 86 state-dict keys."""
    import models
    from dram_amd.configs import ST_DRAM_REF_MODEL
    z = np.load(os.path.join(golden_dir, "dc3d_full.npz"))
    torch.manual_seed(0)
    model = models.DC3D(**ST_DRAM_REF_MODEL)
    model.init(models.HeNorm(mode="fan_in"))
    sd = model.state_dict()
    ref_keys = sorted(k[len("full_bn/sdsum/"):] for k in z.files if k.startswith("full_bn/sdsum/"))
    assert sorted(sd.keys()) == ref_keys and len(ref_keys) == 86
    assert sum(p.numel() for p in model.parameters()) == 16317921
    for k in ref_keys:
        v = sd[k].double()
        got = np.array([v.sum().item(), (v * v).sum().item()])
        np.testing.assert_allclose(got, z["full_bn/sdsum/" + k], rtol=1e-9, atol=1e-9, err_msg=k)
    assert model.trace_path is None and model.dummy.requires_grad and not model.dummy.is_cuda
    assert model.in_ch_list == ST_DRAM_REF_MODEL["in_ch_list"]


def test_dc3d_constructor_asserts():
    import models
    with pytest.raises(AssertionError):
        models.DC3D(1, [1, 2], [2, 2], [2], 1, [1, 1], [0, 0], 0.0)


def test_public_signatures_match_reference(golden_dir):
    """Every public callable of the reference's parts.py / models.py exists here with the same call signature, and so
    does every public method (tests/golden/signatures.json, written by oracle/make_golden.py from the reference)."""
    import inspect
    import json
    import models
    import parts
    from oracle.make_golden import PCM_DGL_INTERNALS
    ref = json.load(open(os.path.join(golden_dir, "signatures.json")))
    assert len(ref) == 14
    for name, entry in ref.items():
        mod = parts if entry["module"] == "parts" else models
        obj = getattr(mod, name)
        got = str(inspect.signature(obj.__init__ if inspect.isclass(obj) else obj))
        assert got == entry["signature"], (name, got, entry["signature"])
        for meth, sig in entry.get("methods", {}).items():
            if name == "PCM" and meth in PCM_DGL_INTERNALS:
                continue                                   # DGL message passing: replaced by the grid stencil kernels
            assert hasattr(obj, meth), (name, meth)
            if name == "PCM" and meth == "init_graph":
                continue                                   # same name, returns the stencil offsets; arguments optional here
            assert str(inspect.signature(getattr(obj, meth))) == sig, (name, meth)


def test_package_configs_equal_the_oracle_copies():
    """The product package carries its own model configurations (dram_amd/configs.py) so that nothing under
    bodyct-dram_amd/, bench.py's GPU path or the GPU tests needs oracle/ for a constant; the oracle's copies
    (used to generate the goldens) must be the same dicts."""
    from dram_amd import configs as C
    from oracle import dram_oracle as O
    from oracle import make_golden as G
    assert C.ST_DRAM_REF_MODEL == O.ST_DRAM_REF_MODEL
    assert C.ST_DRAM_REF_ATT_MODEL == O.ST_DRAM_REF_ATT_MODEL
    assert C.SLIM == G.SLIM and C.SLIM_ATT == G.SLIM_ATT


def test_checkpoint_wrapper_both_branches():
    """parts.checkpoint_wrapper (reference dram/parts.py:57-64): segments > 0 -> torch.utils.checkpoint (the module's
    forward runs again during backward), otherwise a plain call; same values and gradients either way."""
    import parts

    class Counting(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.lin = torch.nn.Linear(5, 3)
            self.calls = 0

        def forward(self, a, b):
            self.calls += 1
            return torch.tanh(self.lin(a)) * b

    torch.manual_seed(0)
    m = Counting()
    a0, b0 = torch.randn(4, 5), torch.randn(4, 3)
    results = {}
    for segments in (0, 2):
        m.calls = 0
        m.zero_grad()
        a, b = a0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
        out = parts.checkpoint_wrapper(m, segments, a, b)
        assert m.calls == 1
        out.sum().backward()
        assert m.calls == (2 if segments > 0 else 1)          # recomputed in backward only when checkpointed
        results[segments] = (out.detach().clone(), a.grad.clone(), b.grad.clone(), m.lin.weight.grad.clone())
    for x, y in zip(results[0], results[2]):
        assert torch.equal(x, y)
    ref = torch.tanh(m.lin(a0)) * b0
    assert torch.equal(results[0][0], ref.detach())


def test_one_hip_runtime_whatever_the_import_order():
    """libdram_hip.so must bind to the HIP runtime torch brought (the one that owns the buffers and streams it is
    handed).  Imported before torch it used to pull /opt/rocm's libamdhip64 next to torch's bundled copy: two runtimes
    in one process, every launch failing with "no ROCm-capable device" (seen in round 2 through
    `python __graft_entry__.py smoke`, whose build() imports the package first)."""
    import subprocess
    import sys
    code = ("import sys, re; sys.path.insert(0, %r); import dram_amd; import torch; "
            "libs = sorted(set(re.findall(r'\\S*libamdhip64\\S*', open('/proc/self/maps').read()))); "
            "print(len(libs), libs)" % os.path.join(ROOT, "bodyct-dram_amd"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.split()[0] == "1", out.stdout


def test_engine_applicability_and_footprint():
    """Host logic of the fused engine (dram_amd/engine.py) that needs no device: which networks it covers (the others
    take the per-op path), and the raw-output footprint that picks its memory mode."""
    import models
    from dram_amd import engine
    from dram_amd.configs import SLIM, ST_DRAM_REF_MODEL
    assert engine.supports(models.DC3D(**SLIM))
    for norm in ("bnt", "bntna", "ln", "lnna", "in"):
        assert engine.supports(models.DC3D(**SLIM, norm_method=norm)), norm
    assert engine.supports(models.DC3D(**SLIM, norm_method="sbn"))             # cross-rank statistics: combined inside the engine
    assert not engine.supports(models.DC3D(**SLIM, norm_method=None))          # no norm (and conv biases)
    assert not engine.supports(models.DC3D(**SLIM, act_method="prelu"))
    assert not engine.supports(models.DC3D(**dict(SLIM, dropout=0.1)))
    full = models.DC3D(**ST_DRAM_REF_MODEL)
    gib = lambda n: engine._raw_output_bytes(full, torch.empty(n, 1, 128, 128, 128, device="meta")) / 2 ** 30
    # 295.5 channel-planes at full resolution per sample: 32+64 | (64+128)/8 | (128+256)/64 | (256+512)/512 | 512/64 | 256/8 | 128
    assert abs(gib(1) - 295.5 * 128 ** 3 * 4 / 2 ** 30) < 1e-6
    assert 2.2 * gib(32) < 0.62 * 288 < 2.2 * gib(64)        # "speed" up to 32 chunks, "tight" for the whole batch of 64


def test_isa_has_no_store_data_hazard():
    """The hazard class behind round 3's silent data fault -- a store of more than 64 bits whose data registers the NEXT VALU
    instruction overwrites; hipcc 7.2 leaves the pair unseparated when a buffer store's soffset is an SGPR, and on gfx950 the
    store then picks up the new value (conv3d_k3_fwd_c1w_kernel) -- must not occur in ANY kernel of the
    built library.  scripts/isa_hazards.py disassembles every gfx950 code object of libdram_hip.so (host tools only); the scanner
    itself was checked against a build with the kernel's pinned wait states removed: 56 hits, all in that kernel."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("isa_hazards", os.path.join(ROOT, "scripts", "isa_hazards.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main(os.path.join(ROOT, "bodyct-dram_amd", "libdram_hip.so")) == 0
