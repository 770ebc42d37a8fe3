"""Diagnostic (not part of the product): run the wgrad kernel from a -DDRAM_STAMP build and print where a box
iteration spends its cycles (s_memtime stamps, shares only)."""
import ctypes, os, sys
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
lib = ctypes.CDLL(os.path.join(ROOT, "scripts", "libdram_hip_stamp.so"))
P, I, Z = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
lib.dram_conv3d_k3_wgrad.argtypes = [P, P, P, P, Z, I, I, I, I, I, I, P]
lib.dram_conv3d_k3_wgrad_ws_bytes.restype = Z
lib.dram_conv3d_k3_wgrad_ws_bytes.argtypes = [I] * 6
for (N, Ci, Co, S) in [(4, 64, 64, 128), (8, 384, 128, 64)]:
    x = torch.rand(N, Ci, S, S, S, device="cuda") - 0.5
    dy = torch.rand(N, Co, S, S, S, device="cuda") - 0.5
    dw = torch.empty(Co, Ci, 27, device="cuda")
    nb = lib.dram_conv3d_k3_wgrad_ws_bytes(N, Ci, Co, S, S, S)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    lib.dram_conv3d_k3_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), nb, N, Ci, Co, S, S, S, st)
    torch.cuda.synchronize()
    lib.dram_debug_stamps(None, 1)
    lib.dram_conv3d_k3_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), nb, N, Ci, Co, S, S, S, st)
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 16)()
    lib.dram_debug_stamps(out, 0)
    names = ["load_box issue", "compute", "barrier1 wait", "vmcnt wait", "store+barrier2"]
    for half, base in (("waves 0-3", 0), ("waves 4-7", 6)):
        n = max(out[base + 5], 1)
        tot = sum(out[base + q] for q in range(5))
        print(f"[{N},{Ci}->{Co},{S}^3] {half}: boxes/wave-sum {n}, cycles/box {tot / n:.0f}: " +
              ", ".join(f"{names[q]} {out[base + q] / n:.0f} ({100.0 * out[base + q] / tot:.1f}%)" for q in range(5)))
