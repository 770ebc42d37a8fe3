"""BASELINE config 5: whole-volume inference on a synthetic 300x512x512 CT (5 lobes), 1 GPU:
time of the GPU path and mask Dice against the CPU oracle."""
import argparse, os, sys, time
import numpy as np, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bodyct-dram_amd")]
import models
from dram_amd.inference import LobeInference, dice, synthetic_ct
from oracle import dram_oracle as O
from dram_amd.configs import ST_DRAM_REF_MODEL
ap = argparse.ArgumentParser(); ap.add_argument("--shape", default="300,512,512"); ap.add_argument("--no-oracle", action="store_true")
args = ap.parse_args()
shape = tuple(int(v) for v in args.shape.split(","))
scan, lobe, spacing = synthetic_ct(shape, (1.0, 0.7, 0.7), seed=7)
torch.manual_seed(0)
model = models.DC3D(**ST_DRAM_REF_MODEL); model.init(models.HeNorm(mode="fan_in"))
params, buffers = O.split_state_dict({k: v.clone() for k, v in model.state_dict().items()})
model = model.cuda().eval()
inf = LobeInference(model)
scan_d, lobe_d = torch.from_numpy(scan).cuda(), torch.from_numpy(lobe).cuda()
res = inf.run(scan_d, lobe_d, spacing); torch.cuda.synchronize()
t0 = time.perf_counter(); reps = 5
for _ in range(reps):
    res = inf.run(scan_d, lobe_d, spacing)
torch.cuda.synchronize(); t_gpu = (time.perf_counter() - t0) / reps
print(f"GPU path: {t_gpu * 1e3:.1f} ms per scan {shape} ({len(res['chunks'])} lobes as one batch), threshold {res['threshold']:.4f}, "
      f"lesion ratio {res['lesion_ratio']:.5f}, ctss {res['ctss']}", flush=True)
if not args.no_oracle:
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    t0 = time.perf_counter()
    htp_ref, mask_ref, th_ref, ratio_ref = O.evaluate_scan(ST_DRAM_REF_MODEL, params, buffers, scan, lobe, spacing)
    t_cpu = time.perf_counter() - t0
    d = dice(res["mask"].cpu().numpy(), mask_ref, 1e-5)
    print(f"CPU oracle: {t_cpu:.1f} s; mask Dice GPU vs CPU = {d:.6f}; max |htp diff| = "
          f"{np.abs(res['htp'].cpu().numpy() - htp_ref).max():.2e}; thresholds {res['threshold']:.4f} / {th_ref:.4f}")
