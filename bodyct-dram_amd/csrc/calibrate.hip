// Measured ceilings of the device the library runs on ("calibrate both peaks on the box with a copy
// kernel and an FMA loop and quote achieved/peak against the measured ceilings as well as the vendor ones"), gfx950.
//
//   dram_calibrate_hbm_copy   dst[i] = src[i], one non-temporal 16-byte load + store per thread: the streaming rate a kernel
//                             can reach when it reads and writes every byte once (bytes moved = 2 x nbytes).
//   dram_calibrate_mfma_f32   a register-only loop of v_mfma_f32_32x32x2_f32 on 8 independent accumulator tiles, two waves
//                             per SIMD on every CU: the issue rate of the exact-fp32 matrix instruction the conv kernels are
//                             built on (FLOPs = blocks x 8 waves x iters x 32 MFMAs x 4096).
// Both only launch; bench.py brackets them with HIP events on the launch stream.  No result is consumed: `sink` receives one
// float per thread so that the loop is not removed.
#include "common.h"
#include <stdlib.h>

namespace dram {

typedef float calib_f32x16 __attribute__((ext_vector_type(16)));
typedef float calib_f32x4 __attribute__((ext_vector_type(4)));

// U independent 16-byte loads per thread in flight before the first store; one pass, no loop (a block = 256 x U x 16 B)
template <int U, bool NT>
__global__ __launch_bounds__(256) void calibrate_copy_kernel(const calib_f32x4* __restrict__ src, calib_f32x4* __restrict__ dst, size_t n16) {
    const size_t base = (size_t)blockIdx.x * (256 * U) + threadIdx.x;
    calib_f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const size_t i = base + (size_t)u * 256;
        if (i < n16) v[u] = NT ? __builtin_nontemporal_load(src + i) : src[i];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const size_t i = base + (size_t)u * 256;
        if (i < n16) {
            if (NT) __builtin_nontemporal_store(v[u], dst + i);
            else dst[i] = v[u];
        }
    }
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
    // Measured on MI355X (scripts/calib_sweep.py, 1 GiB, read + written bytes / time): one 16-byte load per thread, non-temporal:
    // 6.64 TB/s; the same with default cache policy 6.24; four loads in flight per thread 5.68 / 6.21 (nt); eight 4.33 / 4.38.
    static const int variant = getenv("DRAM_CALIB_COPY_VARIANT") ? atoi(getenv("DRAM_CALIB_COPY_VARIANT")) : 5;   // (sweeps only)
    const int U = (variant & 3) == 1 ? 1 : ((variant & 3) == 2 ? 8 : 4);
    const size_t blocks = (n16 + (size_t)256 * U - 1) / ((size_t)256 * U);
    DRAM_REQUIRE(blocks <= 0x7fffffffull, "calibrate_hbm_copy: buffer too large");
    const dim3 grid((unsigned)blocks), blk(256);
    hipStream_t st = (hipStream_t)stream;
    const bool nt = (variant & 4) != 0;
    if (U == 1) { if (nt) hipLaunchKernelGGL((calibrate_copy_kernel<1, true>), grid, blk, 0, st, (const calib_f32x4*)src, (calib_f32x4*)dst, n16); else hipLaunchKernelGGL((calibrate_copy_kernel<1, false>), grid, blk, 0, st, (const calib_f32x4*)src, (calib_f32x4*)dst, n16); }
    else if (U == 8) { if (nt) hipLaunchKernelGGL((calibrate_copy_kernel<8, true>), grid, blk, 0, st, (const calib_f32x4*)src, (calib_f32x4*)dst, n16); else hipLaunchKernelGGL((calibrate_copy_kernel<8, false>), grid, blk, 0, st, (const calib_f32x4*)src, (calib_f32x4*)dst, n16); }
    else { if (nt) hipLaunchKernelGGL((calibrate_copy_kernel<4, true>), grid, blk, 0, st, (const calib_f32x4*)src, (calib_f32x4*)dst, n16); else hipLaunchKernelGGL((calibrate_copy_kernel<4, false>), grid, blk, 0, st, (const calib_f32x4*)src, (calib_f32x4*)dst, n16); }
    return check_launch("calibrate_hbm_copy");
}

// sink: blocks * 512 floats.  Returns the FLOPs of the launch in *flops (may be NULL).
extern "C" int dram_calibrate_mfma_f32(float* sink, int blocks, int iters, double* flops, void* stream) {
    DRAM_REQUIRE(sink && blocks > 0 && iters > 0, "calibrate_mfma_f32: bad arguments");
    hipLaunchKernelGGL(calibrate_mfma_kernel, dim3((unsigned)blocks), dim3(512), 0, (hipStream_t)stream, sink, iters, 1.0f, 0.5f);
    if (flops) *flops = (double)blocks * 8.0 * (double)iters * 32.0 * 4096.0;
    return check_launch("calibrate_mfma_f32");
}
