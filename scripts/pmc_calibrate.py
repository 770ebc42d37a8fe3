"""Known-byte-count launches to calibrate rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths
our kernels use (MI355X_MICROARCH.md: FETCH_SIZE reads 1/2 for 16 B/lane streams, other widths uncalibrated).
relu_fwd_kernel: 4 B/lane loads+stores of n floats.  row_affine_act_kernel<true>: 16 B/lane."""
import os, sys
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bodyct-dram_amd")]
from dram_amd import _lib
n = 1 << 28   # 1 GiB read + 1 GiB written per launch
x = torch.rand(n, device="cuda"); y = torch.empty_like(x)
st = torch.cuda.current_stream().cuda_stream
coef = torch.tensor([1.0, 0.0] * 64, device="cuda")
for _ in range(2):
    _lib.call("dram_relu_fwd", x.data_ptr(), y.data_ptr(), n, st)
    # 64 rows of n/64 floats through the float4 path: y = relu(1*x + 0)
    sm = torch.empty(64, device="cuda"); sr = torch.empty(64, device="cuda"); rc = torch.empty(128, device="cuda")
    _lib.call("dram_bn_fwd_eval", x.data_ptr(), None, None, torch.zeros(64, device="cuda").data_ptr(),
              torch.ones(64, device="cuda").data_ptr(), y.data_ptr(), sm.data_ptr(), sr.data_ptr(), rc.data_ptr(),
              0.0, 1, 1, 64, n // 64, st)
torch.cuda.synchronize()
print("calibration launches done: each kernel reads 1 GiB and writes 1 GiB")
