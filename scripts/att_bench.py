"""Step time of DC3DATGeneric(st_dram_ref_att) -- the model process_pipeline.py loads -- on one GPU:
TRAIN_BATCH_SIZE 10 chunks of RESAMPLE_SIZE 80^3 (st_dram_ref_att.py:40-45), attention grid 64^3, 18
neighbours; prints ms per train step and the share of the PCM kernels (rocprofv3 gives the split).
    python scripts/att_bench.py [--n 10] [--size 80] [--steps 3]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bodyct-dram_amd")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10)
    ap.add_argument("--size", type=int, default=80)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--no-fused", action="store_true", help="the per-op path instead of the fused engine")
    ap.add_argument("--dc3d", action="store_true", help="DC3D(st_dram_ref) instead of the attention model (same shape: the reference's own training batch)")
    a = ap.parse_args()
    import models
    from dram_amd.train_step import DataParallelTrainer, synthetic_batch
    cfg = dict(n_layers=3, in_ch_list=[1, 64, 128, 256, 768, 384, 192], base_ch_list=[32, 64, 128, 256, 256, 128, 64],
               end_ch_list=[64, 128, 256, 512, 256, 128, 64], kernel_sizes=[(3, 3)] * 7, stacking=3,
               padding_list=[(1, 1)] * 7, checkpoint_layers=[0, 1, 0, 1, 0, 1, 0], dropout=0.0, upsample_ksize=(3, 3, 3),
               upsample_sf=(2, 2, 2), out_ch=1, at_spatial_size=(64, 64, 64), at_f_dim=8, at_g_dim=8, at_g_iter=1,
               at_k_size=3, at_merge_type="scaled_dot_product_relu", at_self_loop=False, at_layers=[-1, 0, 1],
               at_p_enc_dim=0, at_geo_f_dim=0)
    torch.manual_seed(0)
    if a.dc3d:
        from dram_amd.configs import ST_DRAM_REF_MODEL
        m = models.DC3D(**ST_DRAM_REF_MODEL)
    else:
        m = models.DC3DATGeneric(**cfg)
    m.init(models.HeNorm(mode="fan_in"))
    m = m.cuda().train()
    m.fused = not a.no_fused
    tr = DataParallelTrainer(m, torch.optim.Adam(m.parameters(), lr=1e-4))
    batch = synthetic_batch(a.n, a.size, 100, torch.device("cuda"))
    tr.step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        reg, seg = tr.step(batch)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    if a.dc3d:
        print({"model": "DC3D(st_dram_ref)", "fused_engine": bool(m.fused), "peak_gb": torch.cuda.max_memory_allocated() / 2 ** 30, "chunks": a.n,
               "size": a.size, "ms_per_step": dt * 1e3, "voxels_per_s": a.n * a.size ** 3 / dt, "reg": float(reg), "seg": float(seg)})
        return
    # the attention alone
    att = m.attention_module
    cam = torch.randn(a.n, 1, 64, 64, 64, device="cuda", requires_grad=True)
    f = torch.randn(a.n, 17, 64, 64, 64, device="cuda", requires_grad=True)
    for _ in range(2):
        att(cam, f).sum().backward()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        att(cam, f).sum().backward()
    torch.cuda.synchronize()
    dpcm = (time.perf_counter() - t0) / 10
    print({"model": "DC3DATGeneric(st_dram_ref_att)", "fused_engine": bool(m.fused), "peak_gb": torch.cuda.max_memory_allocated() / 2 ** 30, "chunks": a.n, "size": a.size, "ms_per_step": dt * 1e3,
           "voxels_per_s": a.n * a.size ** 3 / dt, "pcm_fwd_bwd_ms": dpcm * 1e3, "reg": float(reg), "seg": float(seg)})


if __name__ == "__main__":
    main()
