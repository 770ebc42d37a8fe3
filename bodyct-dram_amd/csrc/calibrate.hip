// Measured ceilings of the device the library runs on ("calibrate both peaks on the box with a copy
// kernel and an FMA loop and quote achieved/peak against the measured ceilings as well as the vendor ones"), gfx950.
//
//   dram_calibrate_hbm_copy   dst[i] = src[i], 16 bytes per lane and instruction, grid-stride: the streaming rate a kernel of
//                             this library can reach when it reads and writes every byte once (bytes moved = 2 x nbytes).
//   dram_calibrate_mfma_f32   a register-only loop of v_mfma_f32_32x32x2_f32 on 8 independent accumulator tiles, two waves
//                             per SIMD on every CU: the issue rate of the exact-fp32 matrix instruction the conv kernels are
//                             built on (FLOPs = blocks x 8 waves x iters x 32 MFMAs x 4096).
// Both only launch; bench.py brackets them with HIP events on the launch stream.  No result is consumed: `sink` receives one
// float per thread so that the loop is not removed.
#include "common.h"

namespace dram {

typedef float calib_f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void calibrate_copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t n16) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
}

__global__ __launch_bounds__(512, 1) void calibrate_mfma_kernel(float* __restrict__ sink, int iters, float a0, float b0) {
    calib_f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const float a = a0 + (float)threadIdx.x, b = b0 - (float)threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float res = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) res += acc[i][r];
    sink[(size_t)blockIdx.x * 512 + threadIdx.x] = res;
}

}  // namespace dram

using namespace dram;

extern "C" int dram_calibrate_hbm_copy(const void* src, void* dst, size_t nbytes, void* stream) {
    DRAM_REQUIRE(src && dst && nbytes >= 16 && nbytes % 16 == 0, "calibrate_hbm_copy: need two buffers of a multiple of 16 bytes");
    DRAM_REQUIRE(((((unsigned long long)src) | ((unsigned long long)dst)) & 15ull) == 0, "calibrate_hbm_copy: buffers must be 16-byte aligned");
    const size_t n16 = nbytes / 16;
    const size_t want = (n16 + 255) / 256;
    const unsigned grid = (unsigned)(want < 256 * 32 ? want : 256 * 32);
    hipLaunchKernelGGL(calibrate_copy_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float4*)src, (float4*)dst, n16);
    return check_launch("calibrate_hbm_copy");
}

// sink: blocks * 512 floats.  Returns the FLOPs of the launch in *flops (may be NULL).
extern "C" int dram_calibrate_mfma_f32(float* sink, int blocks, int iters, double* flops, void* stream) {
    DRAM_REQUIRE(sink && blocks > 0 && iters > 0, "calibrate_mfma_f32: bad arguments");
    hipLaunchKernelGGL(calibrate_mfma_kernel, dim3((unsigned)blocks), dim3(512), 0, (hipStream_t)stream, sink, iters, 1.0f, 0.5f);
    if (flops) *flops = (double)blocks * 8.0 * (double)iters * 32.0 * 4096.0;
    return check_launch("calibrate_mfma_f32");
}
