// What an LDS-DMA instruction (buffer_load_dwordx4 ... lds / buffer_load_dword ... lds) costs the ISSUING wave on gfx950,
// by address pattern: the 64 lanes' 16-byte pieces contiguous (1 KB), in rows of 160 / 96 / 64 bytes at a large stride,
// or every lane in its own 128-byte line; and a plain buffer_load_dword (register destination) with 2 rows of 128 bytes.
// One block of 512 threads per CU (as the conv kernels run), every wave issues NI instructions back to back, s_memtime
// around the batch (issue only) and around batch + wait.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 scripts/ubench/lds_dma_patterns.hip -o scripts/ubench/lds_dma_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((address_space(3))) void* lds_ptr_t;
constexpr int NI = 8;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

// pattern: 0 contiguous 1 KB; 1 rows of 160 B (10 lanes); 2 rows of 96 B (6 lanes); 3 rows of 64 B (4 lanes); 4 one line per lane
__device__ __forceinline__ unsigned lane_offset(int pattern, int lane, int i, int wave, int block) {
    const unsigned region = ((unsigned)block * 8u + wave) * NI + i;     // distinct memory per instruction
    const unsigned base = region * 65536u;
    switch (pattern) {
        case 0: return base + 16u * lane;
        case 1: return base + (lane / 10) * 4096u + 112u + 16u * (lane % 10);
        case 2: return base + (lane / 6) * 4096u + 48u + 16u * (lane % 6);
        case 3: return base + (lane / 4) * 4096u + 16u * (lane % 4);
        default: return base + 512u * lane;
    }
}

template <int BYTES>
__global__ __launch_bounds__(512, 1) void k(const float* src, unsigned bytes, int pattern, int waves_active, long long* out, float* sink) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t srd = make_rsrc(src, bytes);
    unsigned off[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) off[i] = lane_offset(pattern, lane, i, wave, blockIdx.x) % (bytes - 4096u);
    __syncthreads();
    long long t0 = 0, t1 = 0, t2 = 0;
    float acc = 0.f;
    if (wave < waves_active) {
        t0 = __builtin_readcyclecounter();
        if constexpr (BYTES > 0) {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                if constexpr (BYTES == 16)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lds_ptr_t)(lds + (wave * NI + i) * 256), 16, (int)off[i], 0, 0, 0);
                else
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lds_ptr_t)(lds + (wave * NI + i) * 256), 4, (int)off[i], 0, 0, 0);
            }
        } else {
            float v[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                // two rows of 128 bytes, 4 bytes in front of a line boundary (the old patch loads)
                const unsigned o = (off[i] & ~4095u) + (lane >> 5) * 8192u + 124u + 4u * (lane & 31);
                v[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd, (int)(o % (bytes - 4096u)), 0, 0));
            }
            t1 = __builtin_readcyclecounter();
#pragma unroll
            for (int i = 0; i < NI; ++i) acc += v[i];
        }
        if (BYTES > 0) t1 = __builtin_readcyclecounter();
        __builtin_amdgcn_s_waitcnt(0);
        asm volatile("s_waitcnt vmcnt(0)");
        t2 = __builtin_readcyclecounter();
    }
    __syncthreads();
    if (lane == 0 && wave < waves_active) {
        atomicAdd((unsigned long long*)&out[0], (unsigned long long)(t1 - t0));
        atomicAdd((unsigned long long*)&out[1], (unsigned long long)(t2 - t0));
        atomicAdd((unsigned long long*)&out[2], 1ull);
    }
    if (acc == 123.456f) sink[0] = acc + lds[threadIdx.x];
}

int main() {
    const size_t bytes = 1ull << 31;
    float* src; long long* out; float* sink;
    hipMalloc(&src, bytes); hipMemset(src, 0, bytes); hipMalloc(&out, 64); hipMalloc(&sink, 64);
    const char* names[] = {"contiguous 1 KB", "rows of 160 B", "rows of 96 B", "rows of 64 B", "a line per lane"};
    for (int wa : {1, 4, 8})
        for (int kind = 0; kind < 3; ++kind)
            for (int p = 0; p < (kind == 2 ? 1 : 5); ++p) {
                long long h[3];
                for (int rep = 0; rep < 2; ++rep) {       // second repetition: the same addresses again (L2-warm)
                    hipMemset(out, 0, 64);
                    if (kind == 0) hipLaunchKernelGGL(k<16>, dim3(256), dim3(512), 65536, 0, src, (unsigned)(bytes - 1), p, wa, out, sink);
                    else if (kind == 1) hipLaunchKernelGGL(k<4>, dim3(256), dim3(512), 65536, 0, src, (unsigned)(bytes - 1), p, wa, out, sink);
                    else hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 65536, 0, src, (unsigned)(bytes - 1), p, wa, out, sink);
                    hipDeviceSynchronize();
                    hipMemcpy(h, out, 24, hipMemcpyDeviceToHost);
                    printf("%d waves/CU issuing, %-28s %-16s %s: issue %6.0f cycles per instruction, until data %7.0f per batch of %d\n", wa,
                           kind == 0 ? "lds-dma 16 B/lane" : kind == 1 ? "lds-dma 4 B/lane" : "buffer_load_dword (regs)",
                           kind == 2 ? "2 rows of 128 B" : names[p], rep ? "warm" : "cold", (double)h[0] / h[2] / NI, (double)h[1] / h[2], NI);
                }
            }
    return 0;
}
