"""First-layer (Cin = 1) forward: conv3d_k3_fwd_c1_kernel against the generic direct kernel (DRAM_CONV_DIRECT=1), with and
without the statistics epilogue, HIP-event timed; GB/s = algorithmic bytes (4 B read + 4*Cout B written per voxel) / time."""
import os, subprocess, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bodyct-dram_amd")]


def child():
    import torch
    from dram_amd import functional as HF
    from dram_amd import _lib
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    p = lambda t: None if t is None else t.data_ptr()
    for N, Co, S in ((64, 32, 128), (10, 32, 80), (16, 40, 64)):
        x = torch.rand(N, 1, S, S, S, device=dev)
        w = torch.randn(Co, 1, 3, 3, 3, device=dev) / 27 ** 0.5
        wt = HF._pack(w, 0)
        y = torch.empty(N, Co, S, S, S, device=dev)
        nparts = _lib.lib.dram_conv3d_k3_stats_parts(1, Co, S, S, S)
        parts = torch.empty(N * Co * nparts * 3, device=dev)
        name = HF.conv_fwd_kernel_name((S, S, S), Co, 1, fused=True)
        out = []
        for stt in (None, parts):
            fn = lambda: _lib.call("dram_conv3d_k3_fwd_fused", p(x), 1, None, 0, None, 0, None, 0, 0, 0, 0, 0, 0, 0, p(wt), None,
                                   p(y), p(stt), nparts if stt is not None else 0, N, Co, S, S, S, st)
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            out.append(f"{'stats' if stt is not None else 'plain'} {ms:7.3f} ms {4.0 * (1 + Co) * N * S ** 3 / ms / 1e6:7.0f} GB/s")
        print(f"[{N},1->{Co},{S}^3] {name}: " + " | ".join(out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        for env in ({}, {"DRAM_CONV_DIRECT": "1"}):
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, **env))
