// Micro-benchmark: v_mfma_f32_32x32x2_f32 issue rate with 2 waves per SIMD, for accumulator reuse distances 1, 2, 4, 8
// (MFMA i accumulates into acc[i % DIST]).  Build: hipcc --offload-arch=gfx950 -O3 mfma_dep.hip -o mfma_dep
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int DIST, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void k(float* out, int iters) {
    f32x16 acc[8];
    for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int m = 0; m < 16; ++m)
            acc[m % DIST] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m % DIST], 0, 0, 0);
    }
    float s = 0.f;
    for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int DIST, int WAVES>
void run(float* d, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<DIST, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, d, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<DIST, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 256.0 * WAVES * iters * 16 * 2.0 * 32 * 32 * 2;
    printf("waves/CU %d  acc reuse distance %d: %.3f ms  %.1f TFLOP/s (%.3f of 157.3)\n", WAVES, DIST, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3);
}

int main() {
    float* d; hipMalloc(&d, 256 * 1024 * 4);
    const int it = 20000;
    run<1, 4>(d, it); run<2, 4>(d, it); run<4, 4>(d, it); run<8, 4>(d, it);
    run<1, 8>(d, it); run<2, 8>(d, it); run<4, 8>(d, it); run<8, 8>(d, it);
    run<2, 16>(d, it); run<4, 16>(d, it);
    return 0;
}
