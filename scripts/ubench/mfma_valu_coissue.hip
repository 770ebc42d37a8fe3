// Micro-benchmark: can fp32 VALU FMAs run beside fp32 MFMAs on the same SIMD at full rate?
// 8 waves per block (2 per SIMD), 1 block per CU.  mode 0: all waves MFMA; mode 1: all waves VALU;
// mode 2: waves 0-3 MFMA, waves 4-7 VALU (one of each per SIMD).  Register-only loops.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512, 2) void k(float* out, int iters, int mode, float a0, float b0) {
    const int wave = threadIdx.x >> 6;
    const bool do_mfma = mode == 0 || (mode == 2 && wave < 4);
    const bool do_valu = mode == 1 || (mode == 2 && wave >= 4);
    float res = 0.f;
    if (do_mfma) {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        float a = a0 + threadIdx.x, b = b0 - threadIdx.x;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
            }
        }
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) res += acc[i][r];
    }
    if (do_valu) {
        float acc[32];
        for (int i = 0; i < 32; ++i) acc[i] = (float)i;
        float x = a0 + threadIdx.x * 1e-3f, w = b0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
#pragma unroll
                for (int i = 0; i < 32; ++i) acc[i] = fmaf(x, w, acc[i]);
                asm volatile("" : "+v"(x));
            }
        }
        for (int i = 0; i < 32; ++i) res += acc[i];
    }
    out[blockIdx.x * 512 + threadIdx.x] = res;
}

int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    const int iters = 20000;
    for (int mode = 0; mode < 3; ++mode) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, 100, mode, 1.0f, 0.5f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, iters, mode, 1.0f, 0.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double mf_waves = mode == 0 ? 8 : (mode == 2 ? 4 : 0), va_waves = mode == 1 ? 8 : (mode == 2 ? 4 : 0);
        const double mfma_flops = 256.0 * mf_waves * iters * 32.0 * 4096.0;          // 32 MFMAs/iter x 4096 flop
        const double valu_flops = 256.0 * va_waves * iters * 8.0 * 32.0 * 64.0 * 2.0; // 256 fma/iter/lane
        printf("mode %d: %.3f ms  MFMA %.1f TF/s  VALU %.1f TF/s  total %.1f TF/s\n", mode, ms,
               mfma_flops / ms / 1e9, valu_flops / ms / 1e9, (mfma_flops + valu_flops) / ms / 1e9);
    }
    return 0;
}
