"""Diagnostic (not part of the product): s_memtime shares of the Winograd-(z,y) forward kernel's chunk loop.
Build first:  cd bodyct-dram_amd/csrc && for f in *.hip; do hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DDRAM_WZY_STAMPS -c $f -o /tmp/st_$f.o; done;
              hipcc --offload-arch=gfx950 -shared -fPIC /tmp/st_*.o -o ../../scripts/libdram_hip_stamp.so"""
import ctypes, os
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
lib = ctypes.CDLL(os.path.join(ROOT, "scripts", "libdram_hip_stamp.so"))
P, I = ctypes.c_void_p, ctypes.c_int
lib.dram_conv3d_k3_fwd.argtypes = [P, P, P, P, I, I, I, I, I, I, P]
lib.dram_conv3d_k3_pack_weights.argtypes = [P, P, I, I, I, P]
lib.dram_conv3d_k3_packed_floats.restype = ctypes.c_size_t
for (N, Ci, Co, S) in [(4, 64, 64, 128), (4, 192, 64, 128), (8, 384, 128, 64)]:
    x = torch.rand(N, Ci, S, S, S, device="cuda") - 0.5
    w = torch.randn(Co, Ci, 3, 3, 3, device="cuda")
    wt = torch.empty(lib.dram_conv3d_k3_packed_floats(Co, Ci), device="cuda"); y = torch.empty(N, Co, S, S, S, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    lib.dram_conv3d_k3_pack_weights(w.data_ptr(), wt.data_ptr(), Co, Ci, 0, st)
    for rep in range(2):
        lib.dram_debug_wzy_stamps(None, 1)
        lib.dram_conv3d_k3_fwd(x.data_ptr(), wt.data_ptr(), None, y.data_ptr(), N, Ci, Co, S, S, S, st)
        torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 16)()
    lib.dram_debug_wzy_stamps(out, 0)
    n0, nb = out[7], out[8]
    nch = n0 + nb
    names = ["it0", "it1-2", "it3-6", "it7-11", "pre-barrier", "barrier", "it0 at boundary (epilogue)"]
    per = [out[0] / max(n0, 1), out[1] / nch, out[2] / nch, out[3] / nch, out[4] / nch, out[5] / nch, out[6] / max(nb, 1)]
    tot = sum(out[q] for q in range(7)) / nch
    print(f"[{N},{Ci}->{Co},{S}^3] wave-chunks {nch} (boundaries {nb}); cycles per chunk {tot:.0f}: " +
          ", ".join(f"{names[q]} {per[q]:.0f}" for q in range(7)), flush=True)
