"""Sweep of the copy-kernel variants behind dram_calibrate_hbm_copy (one child process per variant: the choice is read once)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    sys.path[:0] = [ROOT, os.path.join(ROOT, "bodyct-dram_amd")]
    import torch
    sys.path.insert(0, ROOT)
    import bench
    print(sys.argv[1], bench.measured_ceilings(torch.device("cuda", 0)))
else:
    for v in (0, 1, 2, 4, 5, 6):
        subprocess.call([sys.executable, os.path.abspath(__file__), str(v)], env=dict(os.environ, DRAM_CALIB_COPY_VARIANT=str(v)))
