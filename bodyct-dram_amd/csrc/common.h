// Shared helpers for libdram_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/dram_hip.h"

namespace dram {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: HIP launch failed: %s", what, hipGetErrorString(e));
        return DRAM_EHIP;
    }
    return DRAM_OK;
}

#define DRAM_REQUIRE(cond, ...)            \
    do {                                   \
        if (!(cond)) {                     \
            dram::set_error(__VA_ARGS__);  \
            return DRAM_EINVAL;            \
        }                                  \
    } while (0)

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline int cdiv(int a, int b) { return (a + b - 1) / b; }
inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// 64-lane wavefront reductions (gfx950: wave = 64)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Block-wide sum for 256-thread blocks; `red` is >= 4 floats of LDS. Result in every thread.
__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

}  // namespace dram
