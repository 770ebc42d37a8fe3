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

// Opt-in to more than 64 KB of dynamic LDS for a kernel.  The attribute belongs to the (function, device)
// pair, so the "already done" flag is kept per device of the calling thread (a process that drives several
// devices sets it once on each); racing threads at worst set it twice.
constexpr int DRAM_MAX_DEVICES = 64;
struct LdsAttrOnce {
    unsigned char done[DRAM_MAX_DEVICES] = {};
};
inline int ensure_dynamic_lds(const void* fn, size_t bytes, LdsAttrOnce& once, const char* who) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= DRAM_MAX_DEVICES) dev = -1;
    if (dev >= 0 && once.done[dev]) return DRAM_OK;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) {
        set_error("%s: hipFuncSetAttribute(MaxDynamicSharedMemorySize=%zu) failed: %s", who, bytes, hipGetErrorString(e));
        return DRAM_EHIP;
    }
    if (dev >= 0) once.done[dev] = 1;
    return DRAM_OK;
}

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
