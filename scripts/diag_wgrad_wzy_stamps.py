"""Diagnostic (not part of the product): s_memtime shares of the Winograd-(z,y) backward-weights kernel's box loop, per wave and
box.  Build first:  cd bodyct-dram_amd/csrc && for f in *.hip; do hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DDRAM_WZY_STAMPS
-c $f -o /tmp/st_${f%.hip}.o; done && hipcc --offload-arch=gfx950 -shared -fPIC /tmp/st_*.o -o ../../scripts/libdram_hip_stamp.so"""
import ctypes, os
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
lib = ctypes.CDLL(os.path.join(ROOT, "scripts", "libdram_hip_stamp.so"))
P, I, Z = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
lib.dram_conv3d_k3_wgrad_fused.argtypes = [P, I, P, I, P, I, P, I, I, I, I, I, I, I, P, P, P, Z, I, I, I, I, I, P]
lib.dram_conv3d_k3_wgrad_ws_bytes.restype = Z
lib.dram_conv3d_k3_wgrad_ws_bytes.argtypes = [I, I, I, I, I, I]
for (N, Ci, Co, S) in [(4, 64, 64, 128), (4, 192, 64, 128), (16, 256, 256, 32)]:
    x = torch.rand(N, Ci, S, S, S, device="cuda") - 0.5
    dy = torch.rand(N, Co, S, S, S, device="cuda") - 0.5
    dw = torch.empty(Co, Ci, 3, 3, 3, device="cuda")
    coef = torch.rand(N * Ci * 2, device="cuda") + 0.5
    nb = lib.dram_conv3d_k3_wgrad_ws_bytes(N, Ci, Co, S, S, S)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for tag, cf in (("plain", None), ("lazy", coef)):
        for rep in range(2):
            lib.dram_debug_wgrad_stamps(None, 1)
            rc = lib.dram_conv3d_k3_wgrad_fused(x.data_ptr(), Ci, None if cf is None else cf.data_ptr(), 1, None, 0, None, 0, 0, 0, 0, 0, 0, 0,
                                                dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), nb, N, Co, S, S, S, st)
            assert rc == 0, rc
            torch.cuda.synchronize()
        out = (ctypes.c_ulonglong * 8)()
        lib.dram_debug_wgrad_stamps(out, 0)
        for half, who in ((0, "waves 0-3 (fetch + transform + one co tile)"), (1, "waves 4-7 (three co tiles)")):
            o = out[4 * half:4 * half + 4]
            nbox = max(o[3], 1)
            print(f"[{N},{Ci}->{Co},{S}^3] {tag:5s} {who}: cycles per wave and box: MFMAs + rides {o[0] / nbox:.0f}, fetch wait {o[1] / nbox:.0f}, "
                  f"barrier {o[2] / nbox:.0f}  (sum {sum(o[:3]) / nbox:.0f}; the SIMD pair's MFMAs: 6144)", flush=True)
