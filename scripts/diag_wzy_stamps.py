"""Diagnostic (not part of the product): s_memtime shares of the Winograd-(z,y) forward kernel's chunk loop, per wave and
chunk, for the plain / statistics-only / lazy-operand variants.
Build first:  cd bodyct-dram_amd/csrc && for f in *.hip; do hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DDRAM_WZY_STAMPS -c $f -o /tmp/st_$f.o; done;
              hipcc --offload-arch=gfx950 -shared -fPIC /tmp/st_*.o -o ../../scripts/libdram_hip_stamp.so"""
import ctypes, os
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
lib = ctypes.CDLL(os.path.join(ROOT, "scripts", os.environ.get("DRAM_STAMP_LIB", "libdram_hip_stamp.so")))
P, I = ctypes.c_void_p, ctypes.c_int
lib.dram_conv3d_k3_fwd_fused.argtypes = [P, I, P, I, P, I, P, I, I, I, I, I, I, I, P, P, P, P, I, I, I, I, I, I, P]
lib.dram_conv3d_k3_pack_weights.argtypes = [P, P, I, I, I, P]
lib.dram_conv3d_k3_packed_floats.restype = ctypes.c_size_t
lib.dram_conv3d_k3_stats_parts.argtypes = [I, I, I, I, I]
for (N, Ci, Co, S) in [(4, 64, 64, 128), (4, 192, 64, 128), (16, 256, 256, 32)]:
    x = torch.rand(N, Ci, S, S, S, device="cuda") - 0.5
    w = torch.randn(Co, Ci, 3, 3, 3, device="cuda")
    wt = torch.empty(lib.dram_conv3d_k3_packed_floats(Co, Ci), device="cuda"); y = torch.empty(N, Co, S, S, S, device="cuda")
    coef = torch.rand(N * Ci * 2, device="cuda") + 0.5
    nparts = lib.dram_conv3d_k3_stats_parts(Ci, Co, S, S, S)
    parts = torch.empty(N * Co * nparts * 3, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    lib.dram_conv3d_k3_pack_weights(w.data_ptr(), wt.data_ptr(), Co, Ci, 0, st)
    for tag, cf, pt in (("plain", None, None), ("stats", None, parts), ("lazy", coef, None), ("both", coef, parts)):
        for rep in range(2):
            lib.dram_debug_wzy_stamps(None, 1)
            lib.dram_conv3d_k3_fwd_fused(x.data_ptr(), Ci, None if cf is None else cf.data_ptr(), 1, None, 0, None, 0, 0, 0, 0, 0, 0, 0,
                                         wt.data_ptr(), None, y.data_ptr(), None if pt is None else pt.data_ptr(),
                                         nparts if pt is not None else 0, N, Co, S, S, S, st)
            torch.cuda.synchronize()
        out32 = (ctypes.c_ulonglong * 32)()
        lib.dram_debug_wzy_stamps(out32, 0)
        for role, base in (("waves 0-3", 0), ("waves 4-7", 16)):
            out = [out32[base + q] for q in range(16)]
            n0, nb = out[7], out[8]
            nch = n0 + nb
            names = ["it0", "it1-2", "it3-6", "it7-11", "pre-barrier", "barrier", "it0 at boundary (epilogue)"]
            per = [out[0] / max(n0, 1), out[1] / nch, out[2] / nch, out[3] / nch, out[4] / nch, out[5] / nch, out[6] / max(nb, 1)]
            tot = sum(out[q] for q in range(7)) / nch
            print(f"[{N},{Ci}->{Co},{S}^3] {tag:5s} {role}: cycles per wave and chunk {tot:.0f}: " +
                  ", ".join(f"{names[q]} {per[q]:.0f}" for q in range(7)), flush=True)
            ne = max(out[15], 1)
            en = ["A^T along y", "exchange write + barrier", "read + combine + barrier", "statistics", "stores"]
            print(f"        epilogue per wave and item: " + ", ".join(f"{en[q]} {out[10 + q] / ne:.0f}" for q in range(5)), flush=True)
