// Micro-benchmark: register-only issue rate of the two exact-fp32 MFMA shapes on gfx950, at 1 and 2 waves per SIMD
// and with 4 / 27 independent accumulators -- the ceiling for the conv kernels' inner loops.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int PRIO = 0>
__global__ __launch_bounds__(512) void k32(float* out, int iters, float a0, float b0) {
    if (PRIO && threadIdx.x >= 256) __builtin_amdgcn_s_setprio(2);   // second wave of each SIMD issues first
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x, b = b0 - threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float res = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) res += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = res;
}

template <int NACC>
__global__ __launch_bounds__(512) void k16(float* out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x, b = b0 - threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float res = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) res += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = res;
}

template <typename K>
static void run(const char* name, K kern, int threads, double flops_per_mfma, int nacc, float* out) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, 100, 1.0f, 0.5f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, iters, 1.0f, 0.5f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = 256.0 * (threads / 64) * iters * 4.0 * nacc * flops_per_mfma;
    printf("%-34s %d waves/SIMD: %.1f TF/s\n", name, threads / 256, flops / ms / 1e9);
}

int main() {
    float* out; (void)hipMalloc(&out, 256 * 512 * 4);
    run("32x32x2  4 accumulators", k32<4>, 512, 4096.0, 4, out);
    run("32x32x2  4 accumulators", k32<4>, 256, 4096.0, 4, out);
    run("32x32x2  4 acc, wave priority", k32<4, 1>, 512, 4096.0, 4, out);
    run("32x32x2  8 accumulators", k32<8>, 512, 4096.0, 8, out);
    run("32x32x2  2 accumulators", k32<2>, 512, 4096.0, 2, out);
    run("16x16x4  4 accumulators", k16<4>, 512, 2048.0, 4, out);
    run("16x16x4 27 accumulators", k16<27>, 512, 2048.0, 27, out);
    run("16x16x4 27 accumulators", k16<27>, 256, 2048.0, 27, out);
    return 0;
}
