"""Diagnostic (not part of the product): s_memtime shares of the fwd conv kernel's chunk loop."""
import ctypes, os
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
lib = ctypes.CDLL(os.path.join(ROOT, "scripts", "libdram_hip_stamp.so"))
P, I = ctypes.c_void_p, ctypes.c_int
lib.dram_conv3d_k3_fwd.argtypes = [P, P, P, P, I, I, I, I, I, I, P]
lib.dram_conv3d_k3_pack_weights.argtypes = [P, P, I, I, I, P]
for (N, Ci, Co, S) in [(4, 64, 64, 128), (4, 192, 64, 128), (8, 384, 128, 64)]:
    x = torch.rand(N, Ci, S, S, S, device="cuda") - 0.5
    w = torch.randn(Co, Ci, 3, 3, 3, device="cuda")
    wt = torch.empty(27 * Ci * Co, device="cuda"); y = torch.empty(N, Co, S, S, S, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    lib.dram_conv3d_k3_pack_weights(w.data_ptr(), wt.data_ptr(), Co, Ci, 0, st)
    for rep in range(2):
        lib.dram_debug_stamps(None, 1)
        lib.dram_conv3d_k3_fwd(x.data_ptr(), wt.data_ptr(), None, y.data_ptr(), N, Ci, Co, S, S, S, st)
        torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 16)()
    lib.dram_debug_stamps(out, 0)
    waves = out[8]; chunks = out[6]
    names = ["prologue", "load issue", "compute", "vmcnt wait", "store", "barrier", None, "epilogue"]
    tot = sum(out[q] for q in (0, 1, 2, 3, 4, 5, 7))
    print(f"[{N},{Ci}->{Co},{S}^3] per wave: total {tot / waves:.0f} cycles, chunks {chunks / waves:.1f}; per chunk: " +
          ", ".join(f"{names[q]} {out[q] / chunks:.0f}" for q in (1, 2, 3, 4, 5)) +
          f"; per block: prologue {out[0] / waves:.0f}, epilogue {out[7] / waves:.0f}; shares: " +
          ", ".join(f"{names[q]} {100.0 * out[q] / tot:.1f}%" for q in (0, 1, 2, 3, 4, 5, 7)))
