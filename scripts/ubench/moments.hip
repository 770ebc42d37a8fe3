// Micro-benchmark: per-(row, chunk) {mean, M2} reduction variants, to find why row_moments_kernel reads at
// 2.5 TB/s while the other row kernels reach 5.5-6.2 TB/s.
//   A  as shipped: 8 float4 per thread in registers, block mean, then squared deviations (two block sums)
//   B  one pass around a pivot (first element of the chunk): sum d, sum d^2, one fused block reduction
//   C  like B, wave-level partials only (each wave writes its own {sum d, sum d^2}; no __syncthreads)
//   D  plain sum (upper bound of a read-only pass)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

constexpr int CHUNK = 8192;
__device__ __forceinline__ float wave_sum(float v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

__global__ __launch_bounds__(256) void kA(const float* __restrict__ x, float* __restrict__ part, int64_t S, int nch) {
    __shared__ float red[4];
    const float* p = x + (int64_t)blockIdx.y * S + (int64_t)blockIdx.x * CHUNK;
    float v[32];
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float4 t = ld4(p + (q * 256 + threadIdx.x) * 4);
        v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
        s += (t.x + t.y) + (t.z + t.w);
    }
    const float mean = block_sum_256(s, red) / CHUNK;
    float m2 = 0.f;
#pragma unroll
    for (int q = 0; q < 32; ++q) { const float d = v[q] - mean; m2 += d * d; }
    m2 = block_sum_256(m2, red);
    if (threadIdx.x == 0) { float* o = part + ((size_t)blockIdx.y * nch + blockIdx.x) * 2; o[0] = mean; o[1] = m2; }
}

__global__ __launch_bounds__(256) void kB(const float* __restrict__ x, float* __restrict__ part, int64_t S, int nch) {
    __shared__ float red[8];
    const float* p = x + (int64_t)blockIdx.y * S + (int64_t)blockIdx.x * CHUNK;
    const float K = p[0];
    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float4 t = ld4(p + (q * 256 + threadIdx.x) * 4);
        const float a = t.x - K, b = t.y - K, c = t.z - K, d = t.w - K;
        s += (a + b) + (c + d);
        ss += (a * a + b * b) + (c * c + d * d);
    }
    s = wave_sum(s); ss = wave_sum(ss);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[w] = s; red[4 + w] = ss; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float S1 = red[0] + red[1] + red[2] + red[3], S2 = red[4] + red[5] + red[6] + red[7];
        float* o = part + ((size_t)blockIdx.y * nch + blockIdx.x) * 2;
        o[0] = K + S1 / CHUNK; o[1] = S2 - S1 * S1 / CHUNK;
    }
}

__global__ __launch_bounds__(256) void kC(const float* __restrict__ x, float* __restrict__ part, int64_t S, int nch) {
    const float* p = x + (int64_t)blockIdx.y * S + (int64_t)blockIdx.x * CHUNK;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const float* pw = p + w * (CHUNK / 4);
    const float K = pw[0];
    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float4 t = ld4(pw + (q * 64 + l) * 4);
        const float a = t.x - K, b = t.y - K, c = t.z - K, d = t.w - K;
        s += (a + b) + (c + d);
        ss += (a * a + b * b) + (c * c + d * d);
    }
    s = wave_sum(s); ss = wave_sum(ss);
    if (l == 0) {
        float* o = part + (((size_t)blockIdx.y * nch + blockIdx.x) * 4 + w) * 2;
        o[0] = K + s / (CHUNK / 4); o[1] = ss - s * s / (CHUNK / 4);
    }
}

__global__ __launch_bounds__(256) void kD(const float* __restrict__ x, float* __restrict__ part, int64_t S, int nch) {
    const float* p = x + (int64_t)blockIdx.y * S + (int64_t)blockIdx.x * CHUNK;
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float4 t = ld4(p + (q * 256 + threadIdx.x) * 4);
        s += (t.x + t.y) + (t.z + t.w);
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) part[(((size_t)blockIdx.y * nch + blockIdx.x) * 4 + (threadIdx.x >> 6)) * 2] = s;
}

int main() {
    const int rows = 512;
    const int64_t S = 128LL * 128 * 128;
    const int nch = (int)(S / CHUNK);
    float *x, *part;
    hipMalloc(&x, rows * S * 4);
    hipMalloc(&part, (size_t)rows * nch * 8 * 4);
    hipMemset(x, 0, rows * S * 4);
    typedef void (*K)(const float*, float*, int64_t, int);
    K ks[4] = {kA, kB, kC, kD};
    const char* names[4] = {"A two-pass registers", "B pivot one-pass", "C pivot per-wave", "D plain sum"};
    for (int k = 0; k < 4; ++k) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(ks[k], dim3(nch, rows), dim3(256), 0, 0, x, part, S, nch);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(ks[k], dim3(nch, rows), dim3(256), 0, 0, x, part, S, nch);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        printf("%-24s %.3f ms  %.0f GB/s\n", names[k], ms, rows * S * 4.0 / ms / 1e6);
    }
    return 0;
}
