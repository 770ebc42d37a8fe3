// 1x1x1 convolution with bias (the regression head), per-channel sums and the lobe-masked
// mean, fp32 NCDHW, gfx950.
//
// Replaces the ATen dispatches of DC3D.top_layer = nn.Conv3d(64, out_ch, 1) (reference
// dram/models.py:109-110,145) and of pooling_dense_features (models.py:37-49).
//
// the head is 64 -> 1 (a per-voxel 64-term dot product, 0.5 FLOP/B): HBM-bound, so
// it is a streaming VALU kernel, not an MFMA GEMM.  Each thread owns 4 consecutive voxels
// (float4 loads, 1 KiB per wave instruction) and walks the channel planes.
#include "common.h"

namespace dram {

constexpr int MAXCO = 8;  // output channels handled per pass

// `coef` (optional): x holds a RAW conv output and the operand is act(a*x + b) per (n,c) row, ReLU if `relu`
// ("normalise + ReLU on load", see csrc/conv3d_k3.hip ConvArgs::coef1; same fmaf / fmaxf as everywhere)
template <bool VEC>
__global__ __launch_bounds__(256) void conv1x1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ y,
                                                          int Cin, int Cout, int co0, int nco, int64_t S,
                                                          const float* __restrict__ coef, int relu) {
    const int n = blockIdx.y;
    const int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) * (VEC ? 4 : 1);
    if (e >= S) return;
    const float* xp = x + (int64_t)n * Cin * S + e;
    float acc[MAXCO][4];
#pragma unroll
    for (int o = 0; o < MAXCO; ++o) {
        const float b = (bias && o < nco) ? bias[co0 + o] : 0.f;
        acc[o][0] = acc[o][1] = acc[o][2] = acc[o][3] = b;
    }
    for (int c = 0; c < Cin; ++c) {
        float xv[4];
        if (VEC) {
            const float4 t = *reinterpret_cast<const float4*>(xp + (int64_t)c * S);
            xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w;
        } else {
            xv[0] = xp[(int64_t)c * S]; xv[1] = xv[2] = xv[3] = 0.f;
        }
        if (coef) {
            const float ca = coef[2 * ((int64_t)n * Cin + c)], cb = coef[2 * ((int64_t)n * Cin + c) + 1];
            const float lo = relu ? 0.f : -INFINITY;
#pragma unroll
            for (int u = 0; u < 4; ++u) xv[u] = fmaxf(fmaf(ca, xv[u], cb), lo);
        }
#pragma unroll
        for (int o = 0; o < MAXCO; ++o) {
            if (o < nco) {
                const float wv = w[(size_t)(co0 + o) * Cin + c];
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[o][u] = fmaf(wv, xv[u], acc[o][u]);
            }
        }
    }
#pragma unroll
    for (int o = 0; o < MAXCO; ++o) {
        if (o < nco) {
            float* yp = y + ((int64_t)n * Cout + co0 + o) * S + e;
            if (VEC) *reinterpret_cast<float4*>(yp) = make_float4(acc[o][0], acc[o][1], acc[o][2], acc[o][3]);
            else yp[0] = acc[o][0];
        }
    }
}

// dx[n,c,s] = sum_o w[o,c] * dy[n,o,s]   (Cout <= MAXCO per pass; accumulate over passes)
template <bool VEC>
__global__ __launch_bounds__(256) void conv1x1_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                            float* __restrict__ dx, int Cin, int Cout, int co0,
                                                            int nco, int64_t S, int accumulate) {
    const int n = blockIdx.y;
    const int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) * (VEC ? 4 : 1);
    if (e >= S) return;
    float g[MAXCO][4];
#pragma unroll
    for (int o = 0; o < MAXCO; ++o) {
        g[o][0] = g[o][1] = g[o][2] = g[o][3] = 0.f;
        if (o < nco) {
            const float* p = dy + ((int64_t)n * Cout + co0 + o) * S + e;
            if (VEC) {
                const float4 t = *reinterpret_cast<const float4*>(p);
                g[o][0] = t.x; g[o][1] = t.y; g[o][2] = t.z; g[o][3] = t.w;
            } else {
                g[o][0] = p[0];
            }
        }
    }
    for (int c = 0; c < Cin; ++c) {
        float r[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int o = 0; o < MAXCO; ++o) {
            if (o < nco) {
                const float wv = w[(size_t)(co0 + o) * Cin + c];
#pragma unroll
                for (int u = 0; u < 4; ++u) r[u] = fmaf(wv, g[o][u], r[u]);
            }
        }
        float* q = dx + ((int64_t)n * Cin + c) * S + e;
        if (VEC) {
            float4 t = make_float4(r[0], r[1], r[2], r[3]);
            if (accumulate) {
                const float4 old = *reinterpret_cast<const float4*>(q);
                t.x += old.x; t.y += old.y; t.z += old.z; t.w += old.w;
            }
            *reinterpret_cast<float4*>(q) = t;
        } else {
            q[0] = accumulate ? q[0] + r[0] : r[0];
        }
    }
}

// Partial weight/bias gradients.  grid (nblk, N); block handles VPB voxels of sample n.
// part[((n*nblk + blk) * Cout + o) * (Cin+1) + c] = sum_s dy[o,s]*x[c,s];  column Cin holds sum_s dy[o,s].
constexpr int WG_VPT = 8;                 // voxels per thread
constexpr int WG_VPB = 256 * WG_VPT;      // voxels per block

__global__ __launch_bounds__(256) void conv1x1_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            float* __restrict__ part, int Cin, int Cout, int64_t S,
                                                            int nblk, const float* __restrict__ coef, int relu) {
    __shared__ float red[4];
    const int n = blockIdx.y, blk = blockIdx.x;
    const int64_t base = (int64_t)blk * WG_VPB;
    for (int o = 0; o < Cout; ++o) {
        float g[WG_VPT];
        float gs = 0.f;
#pragma unroll
        for (int u = 0; u < WG_VPT; ++u) {
            const int64_t e = base + u * 256 + threadIdx.x;
            g[u] = e < S ? dy[((int64_t)n * Cout + o) * S + e] : 0.f;
            gs += g[u];
        }
        float* prow = part + (((size_t)n * nblk + blk) * Cout + o) * (Cin + 1);
        gs = block_sum_256(gs, red);
        if (threadIdx.x == 0) prow[Cin] = gs;
        for (int c = 0; c < Cin; ++c) {
            float s = 0.f;
            const float ca = coef ? coef[2 * ((int64_t)n * Cin + c)] : 1.f, cb = coef ? coef[2 * ((int64_t)n * Cin + c) + 1] : 0.f;
            const float lo = (coef && relu) ? 0.f : -INFINITY;
#pragma unroll
            for (int u = 0; u < WG_VPT; ++u) {
                const int64_t e = base + u * 256 + threadIdx.x;
                if (e < S) {
                    float xv = x[((int64_t)n * Cin + c) * S + e];
                    if (coef) xv = fmaxf(fmaf(ca, xv, cb), lo);
                    s = fmaf(g[u], xv, s);
                }
            }
            s = block_sum_256(s, red);
            if (threadIdx.x == 0) prow[c] = s;
        }
    }
}

// The same for 16-byte aligned rows (S % 4 == 0, aligned pointers): a thread holds two float4 of dy and accumulates WG_CT
// input channels in registers from 16-byte loads that are all in flight together, and the block reduces the WG_CT sums
// once per channel tile (wave shuffles + one LDS exchange) -- 2 barriers per 16 channels instead of 2 per channel, which
// is what held the scalar kernel at 3.6 TB/s.  Same partial layout, same fixed-order final sum.
constexpr int WG_CT = 16;
__global__ __launch_bounds__(256) void conv1x1_wgrad_vec_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                float* __restrict__ part, int Cin, int Cout, int64_t S,
                                                                int nblk, const float* __restrict__ coef, int relu) {
    __shared__ float red[4][WG_CT + 1];
    const int n = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t base = (int64_t)blk * WG_VPB;
    const int64_t e0 = base + 4 * tid, e1 = base + 1024 + 4 * tid;         // two float4 per thread
    const bool ok0 = e0 < S, ok1 = e1 < S;
    const float lo = (coef && relu) ? 0.f : -INFINITY;
    for (int o = 0; o < Cout; ++o) {
        const float* gp = dy + ((int64_t)n * Cout + o) * S;
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 g0 = ok0 ? *reinterpret_cast<const float4*>(gp + e0) : z4;
        const float4 g1 = ok1 ? *reinterpret_cast<const float4*>(gp + e1) : z4;
        float* prow = part + (((size_t)n * nblk + blk) * Cout + o) * (Cin + 1);
        {
            float gs = wave_sum(((g0.x + g0.y) + (g0.z + g0.w)) + ((g1.x + g1.y) + (g1.z + g1.w)));
            if (lane == 0) red[wv][WG_CT] = gs;
        }
        for (int c0 = 0; c0 < Cin; c0 += WG_CT) {
            float acc[WG_CT];
            float4 xa[WG_CT], xb[WG_CT];
#pragma unroll
            for (int k = 0; k < WG_CT; ++k) {
                const int c = c0 + k < Cin ? c0 + k : Cin - 1;             // (tail channels: loaded, not stored)
                const float* xp = x + ((int64_t)n * Cin + c) * S;
                xa[k] = ok0 ? *reinterpret_cast<const float4*>(xp + e0) : z4;
                xb[k] = ok1 ? *reinterpret_cast<const float4*>(xp + e1) : z4;
            }
#pragma unroll
            for (int k = 0; k < WG_CT; ++k) {
                float4 a = xa[k], b = xb[k];
                if (coef) {   // the input is a RAW conv output: act(a*x + b) on load; out-of-range slots have g = 0
                    const int c = c0 + k < Cin ? c0 + k : Cin - 1;
                    const float ca = coef[2 * ((int64_t)n * Cin + c)], cb = coef[2 * ((int64_t)n * Cin + c) + 1];
                    a.x = fmaxf(fmaf(ca, a.x, cb), lo); a.y = fmaxf(fmaf(ca, a.y, cb), lo);
                    a.z = fmaxf(fmaf(ca, a.z, cb), lo); a.w = fmaxf(fmaf(ca, a.w, cb), lo);
                    b.x = fmaxf(fmaf(ca, b.x, cb), lo); b.y = fmaxf(fmaf(ca, b.y, cb), lo);
                    b.z = fmaxf(fmaf(ca, b.z, cb), lo); b.w = fmaxf(fmaf(ca, b.w, cb), lo);
                }
                float sacc = g0.x * a.x;
                sacc = fmaf(g0.y, a.y, sacc); sacc = fmaf(g0.z, a.z, sacc); sacc = fmaf(g0.w, a.w, sacc);
                sacc = fmaf(g1.x, b.x, sacc); sacc = fmaf(g1.y, b.y, sacc); sacc = fmaf(g1.z, b.z, sacc); sacc = fmaf(g1.w, b.w, sacc);
                acc[k] = wave_sum(sacc);
            }
            __syncthreads();                                               // (the previous tile's sums have been read)
            if (lane == 0) {
#pragma unroll
                for (int k = 0; k < WG_CT; ++k) red[wv][k] = acc[k];
            }
            __syncthreads();
            if (tid < WG_CT && c0 + tid < Cin) prow[c0 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
            if (c0 == 0 && tid == WG_CT) prow[Cin] = (red[0][WG_CT] + red[1][WG_CT]) + (red[2][WG_CT] + red[3][WG_CT]);
        }
        __syncthreads();
    }
}

// Several output channels (the 1x1x1 "reshape" convs of DC3DATGeneric: 64 / 128 -> 8, models.py:488-494): the kernels above
// walk the output channels in their OUTER loop and read x once per output channel -- right for the 64 -> 1 head, 8x the
// traffic for 8 outputs (rocprofv3 of the attention-model step: 6.2 % of it, 0.5-2.4 ms per launch for 0.3 ms of bytes).
// Here a block owns WM_VPB voxels of sample n, a tile of WM_CT input channels and a tile of WM_CO output channels: per pair of
// float4 positions it loads the WM_CO dy and WM_CT x vectors ONCE (all in flight together), keeps the WM_CO x WM_CT sums per
// thread in registers over the block's voxels, and reduces them across the block once at the end.  Same partial layout
// (with its own, larger voxel tile), same fixed-order fp64 final sum.  grid (nblk, N, ci tiles * co tiles).
constexpr int WM_CO = 8, WM_CT = 8;
constexpr int WM_VPB = 8192;              // voxels per block: four rounds of two float4 per thread
__global__ __launch_bounds__(256) void conv1x1_wgrad_multi_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                  float* __restrict__ part, int Cin, int Cout, int64_t S,
                                                                  int nblk, int ci_tiles, const float* __restrict__ coef, int relu) {
    __shared__ float red[4][WM_CO * WM_CT + WM_CO];
    const int n = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int c0 = (blockIdx.z % ci_tiles) * WM_CT, o0 = (blockIdx.z / ci_tiles) * WM_CO;
    const float lo = (coef && relu) ? 0.f : -INFINITY;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float acc[WM_CO][WM_CT];
    float gsum[WM_CO];
#pragma unroll
    for (int o = 0; o < WM_CO; ++o) {
        gsum[o] = 0.f;
#pragma unroll
        for (int k = 0; k < WM_CT; ++k) acc[o][k] = 0.f;
    }
    float ca[WM_CT], cb[WM_CT];
#pragma unroll
    for (int k = 0; k < WM_CT; ++k) {
        const int c = c0 + k < Cin ? c0 + k : Cin - 1;                     // (tail channels: loaded, not stored)
        ca[k] = coef ? coef[2 * ((int64_t)n * Cin + c)] : 1.f;
        cb[k] = coef ? coef[2 * ((int64_t)n * Cin + c) + 1] : 0.f;
    }
    for (int64_t e = (int64_t)blk * WM_VPB + 4 * tid; e < (int64_t)(blk + 1) * WM_VPB && e < S; e += 1024) {
        float4 g[WM_CO], xv[WM_CT];
#pragma unroll
        for (int o = 0; o < WM_CO; ++o)
            g[o] = o0 + o < Cout ? *reinterpret_cast<const float4*>(dy + ((int64_t)n * Cout + o0 + o) * S + e) : z4;
#pragma unroll
        for (int k = 0; k < WM_CT; ++k) {
            const int c = c0 + k < Cin ? c0 + k : Cin - 1;
            xv[k] = *reinterpret_cast<const float4*>(x + ((int64_t)n * Cin + c) * S + e);
        }
        if (coef) {       // the input is a RAW conv output: act(a*x + b) on load
#pragma unroll
            for (int k = 0; k < WM_CT; ++k) {
                xv[k].x = fmaxf(fmaf(ca[k], xv[k].x, cb[k]), lo); xv[k].y = fmaxf(fmaf(ca[k], xv[k].y, cb[k]), lo);
                xv[k].z = fmaxf(fmaf(ca[k], xv[k].z, cb[k]), lo); xv[k].w = fmaxf(fmaf(ca[k], xv[k].w, cb[k]), lo);
            }
        }
#pragma unroll
        for (int o = 0; o < WM_CO; ++o) {
            gsum[o] += (g[o].x + g[o].y) + (g[o].z + g[o].w);
#pragma unroll
            for (int k = 0; k < WM_CT; ++k) {
                float a = acc[o][k];
                a = fmaf(g[o].x, xv[k].x, a); a = fmaf(g[o].y, xv[k].y, a); a = fmaf(g[o].z, xv[k].z, a); a = fmaf(g[o].w, xv[k].w, a);
                acc[o][k] = a;
            }
        }
    }
#pragma unroll
    for (int o = 0; o < WM_CO; ++o) {
#pragma unroll
        for (int k = 0; k < WM_CT; ++k) {
            const float v = wave_sum(acc[o][k]);
            if (lane == 0) red[wv][o * WM_CT + k] = v;
        }
        const float gv = wave_sum(gsum[o]);
        if (lane == 0) red[wv][WM_CO * WM_CT + o] = gv;
    }
    __syncthreads();
    if (tid < WM_CO * WM_CT + WM_CO) {
        const float v = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
        const bool is_bias = tid >= WM_CO * WM_CT;
        const int o = is_bias ? tid - WM_CO * WM_CT : tid / WM_CT, k = is_bias ? 0 : tid % WM_CT;
        if (o0 + o < Cout) {
            float* prow = part + (((size_t)n * nblk + blk) * Cout + o0 + o) * (Cin + 1);
            if (is_bias) { if (c0 == 0) prow[Cin] = v; }
            else if (c0 + k < Cin) prow[c0 + k] = v;
        }
    }
}

// out[j] = sum over `count` partial vectors of length L: one 256-thread block per j, fp64, fixed order
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ part, int count, int L,
                                                           float* __restrict__ dw, float* __restrict__ dbias, int Cin,
                                                           int Cout) {
    __shared__ double red[4];
    const int j = blockIdx.x;
    double s = 0.0;
    for (int p = threadIdx.x; p < count; p += 256) s += part[(size_t)p * L + j];
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        s = red[0] + red[1] + red[2] + red[3];
        const int o = j / (Cin + 1), c = j % (Cin + 1);
        if (c < Cin) { if (dw) dw[(size_t)o * Cin + c] = (float)s; }
        else if (dbias) dbias[o] = (float)s;
    }
}

// ---------------------------------------------------------------- per-channel sum (conv bias gradient)
constexpr int CS_CHUNK = 8192;
__global__ __launch_bounds__(256) void row_sum_kernel(const float* __restrict__ x, float* __restrict__ part, int64_t S,
                                                      int nchunks) {
    __shared__ float red[4];
    const int64_t row = blockIdx.y;
    const int64_t beg = (int64_t)blockIdx.x * CS_CHUNK;
    const int len = (int)((S - beg) < CS_CHUNK ? (S - beg) : CS_CHUNK);
    const float* p = x + row * S + beg;
    float s = 0.f;
    for (int e = threadIdx.x; e < len; e += 256) s += p[e];
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) part[(size_t)row * nchunks + blockIdx.x] = s;
}
__global__ void channel_sum_finalize_kernel(const float* __restrict__ part, float* __restrict__ out, int N, int C,
                                            int nchunks) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0;
    for (int n = 0; n < N; ++n)
        for (int ch = 0; ch < nchunks; ++ch) s += part[((size_t)n * C + c) * nchunks + ch];
    out[c] = (float)s;
}

// ---------------------------------------------------------------- masked mean
// part[(row*nchunks+chunk)*2] = {sum x*m, sum m} over the chunk
__global__ __launch_bounds__(256) void masked_sum_kernel(const float* __restrict__ x, const float* __restrict__ m,
                                                         float* __restrict__ part, int C, int64_t S, int nchunks) {
    __shared__ float red[4];
    const int64_t row = blockIdx.y;
    const int n = (int)(row / C);
    const int64_t beg = (int64_t)blockIdx.x * CS_CHUNK;
    const int len = (int)((S - beg) < CS_CHUNK ? (S - beg) : CS_CHUNK);
    const float* p = x + row * S + beg;
    const float* q = m + (int64_t)n * S + beg;
    float s = 0.f, ms = 0.f;
    for (int e = threadIdx.x; e < len; e += 256) {
        const float mv = q[e];
        s = fmaf(p[e], mv, s);
        ms += mv;
    }
    s = block_sum_256(s, red);
    ms = block_sum_256(ms, red);
    if (threadIdx.x == 0) {
        part[((size_t)row * nchunks + blockIdx.x) * 2] = s;
        part[((size_t)row * nchunks + blockIdx.x) * 2 + 1] = ms;
    }
}
__global__ void masked_mean_finalize_kernel(const float* __restrict__ part, float* __restrict__ out,
                                            float* __restrict__ msum, int N, int C, int nchunks) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= N * C) return;
    double s = 0.0, ms = 0.0;
    for (int ch = 0; ch < nchunks; ++ch) {
        s += part[((size_t)row * nchunks + ch) * 2];
        ms += part[((size_t)row * nchunks + ch) * 2 + 1];
    }
    out[row] = (float)(s / ms);
    if (row % C == 0) msum[row / C] = (float)ms;
}
__global__ __launch_bounds__(256) void masked_mean_bwd_kernel(const float* __restrict__ dout,
                                                              const float* __restrict__ m,
                                                              const float* __restrict__ msum, float* __restrict__ dx,
                                                              int C, int64_t S) {
    const int64_t row = blockIdx.y;
    const int n = (int)(row / C);
    const float g = dout[row] / msum[n];
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < S; e += stride) dx[row * S + e] = g * m[(int64_t)n * S + e];
}

static inline bool vec4_ok(const void* a, const void* b, int64_t S) {
    return S % 4 == 0 && ((uintptr_t)a % 16 == 0) && ((uintptr_t)b % 16 == 0);
}

}  // namespace dram

using namespace dram;

static int conv1x1_fwd_run(const char* who, const float* x, const float* coef, int relu, const float* w, const float* bias,
                           float* y, int N, int Cin, int Cout, int64_t S, void* stream) {
    DRAM_REQUIRE(x && w && y, "%s: null pointer", who);
    DRAM_REQUIRE(N > 0 && N <= 65535 && Cin > 0 && Cout > 0 && S > 0, "%s: bad dimensions", who);
    hipStream_t st = (hipStream_t)stream;
    const bool vec = vec4_ok(x, y, S);
    for (int co0 = 0; co0 < Cout; co0 += MAXCO) {
        const int nco = (Cout - co0) < MAXCO ? (Cout - co0) : MAXCO;
        if (vec) {
            dim3 grid((unsigned)cdiv64(S / 4, 256), N);
            hipLaunchKernelGGL(conv1x1_fwd_kernel<true>, grid, dim3(256), 0, st, x, w, bias, y, Cin, Cout, co0, nco, S, coef, relu);
        } else {
            dim3 grid((unsigned)cdiv64(S, 256), N);
            hipLaunchKernelGGL(conv1x1_fwd_kernel<false>, grid, dim3(256), 0, st, x, w, bias, y, Cin, Cout, co0, nco, S, coef, relu);
        }
    }
    return check_launch(who);
}

extern "C" int dram_conv3d_k1_fwd(const float* x, const float* w, const float* bias, float* y, int N, int Cin,
                                  int Cout, int64_t S, void* stream) {
    return conv1x1_fwd_run("conv3d_k1_fwd", x, nullptr, 0, w, bias, y, N, Cin, Cout, S, stream);
}

// 1x1x1 conv of act(coef * x): the norm (+ReLU) of the producing layer applied on load (coef = per-row {a, b})
extern "C" int dram_conv3d_k1_fwd_lazy(const float* x, const float* coef, int relu, const float* w, const float* bias, float* y,
                                       int N, int Cin, int Cout, int64_t S, void* stream) {
    return conv1x1_fwd_run("conv3d_k1_fwd_lazy", x, coef, relu, w, bias, y, N, Cin, Cout, S, stream);
}

extern "C" size_t dram_conv3d_k1_bwd_ws_bytes(int N, int Cin, int Cout, int64_t S) {
    if (N <= 0 || Cin <= 0 || Cout <= 0 || S <= 0) return 0;
    const size_t nblk = (size_t)cdiv64(S, WG_VPB);
    return (size_t)N * nblk * Cout * (Cin + 1) * sizeof(float);
}

static int conv1x1_bwd_run(const float* dy, const float* x, const float* coef, int relu, const float* w, float* dx, float* dw,
                           float* dbias, void* ws, size_t ws_bytes, int N, int Cin, int Cout, int64_t S, void* stream) {
    DRAM_REQUIRE(dy && x && w, "conv3d_k1_bwd: null pointer");
    DRAM_REQUIRE(N > 0 && N <= 65535 && Cin > 0 && Cout > 0 && S > 0, "conv3d_k1_bwd: bad dimensions");
    hipStream_t st = (hipStream_t)stream;
    if (dx) {
        const bool vec = vec4_ok(dy, dx, S);
        for (int co0 = 0; co0 < Cout; co0 += MAXCO) {
            const int nco = (Cout - co0) < MAXCO ? (Cout - co0) : MAXCO;
            if (vec) {
                dim3 grid((unsigned)cdiv64(S / 4, 256), N);
                hipLaunchKernelGGL(conv1x1_dgrad_kernel<true>, grid, dim3(256), 0, st, dy, w, dx, Cin, Cout, co0, nco, S, co0 > 0);
            } else {
                dim3 grid((unsigned)cdiv64(S, 256), N);
                hipLaunchKernelGGL(conv1x1_dgrad_kernel<false>, grid, dim3(256), 0, st, dy, w, dx, Cin, Cout, co0, nco, S, co0 > 0);
            }
        }
    }
    if (dw || dbias) {
        DRAM_REQUIRE(ws, "conv3d_k1_bwd: workspace is null");
        if (ws_bytes < dram_conv3d_k1_bwd_ws_bytes(N, Cin, Cout, S)) {
            set_error("conv3d_k1_bwd: workspace too small");
            return DRAM_EWS;
        }
        int nblk = (int)cdiv64(S, WG_VPB);
        float* part = (float*)ws;
        static_assert(WG_VPB == 2048, "conv1x1_wgrad_vec_kernel: two float4 per thread");
        static_assert(WM_VPB % WG_VPB == 0, "the multi-output kernel's partials fit the workspace sized for WG_VPB");
        const bool vec = (S % 4) == 0 && ((((uintptr_t)dy) | ((uintptr_t)x)) & 15) == 0 && getenv("DRAM_K1_WGRAD_SCALAR") == nullptr;
        const int ci_tiles = cdiv(Cin, WM_CT), co_tiles = cdiv(Cout, WM_CO);
        if (vec && Cout > 1 && ci_tiles * co_tiles <= 65535 && getenv("DRAM_K1_WGRAD_NOMULTI") == nullptr) {
            nblk = (int)cdiv64(S, WM_VPB);
            hipLaunchKernelGGL(conv1x1_wgrad_multi_kernel, dim3(nblk, N, ci_tiles * co_tiles), dim3(256), 0, st, dy, x, part, Cin, Cout,
                               S, nblk, ci_tiles, coef, relu);
        } else if (vec)
            hipLaunchKernelGGL(conv1x1_wgrad_vec_kernel, dim3(nblk, N), dim3(256), 0, st, dy, x, part, Cin, Cout, S, nblk, coef, relu);
        else
            hipLaunchKernelGGL(conv1x1_wgrad_kernel, dim3(nblk, N), dim3(256), 0, st, dy, x, part, Cin, Cout, S, nblk, coef, relu);
        const int L = Cout * (Cin + 1);
        hipLaunchKernelGGL(sum_partials_kernel, dim3(L), dim3(256), 0, st, part, N * nblk, L, dw, dbias, Cin, Cout);
    }
    return check_launch("conv3d_k1_bwd");
}

extern "C" int dram_conv3d_k1_bwd(const float* dy, const float* x, const float* w, float* dx, float* dw, float* dbias,
                                  void* ws, size_t ws_bytes, int N, int Cin, int Cout, int64_t S, void* stream) {
    return conv1x1_bwd_run(dy, x, nullptr, 0, w, dx, dw, dbias, ws, ws_bytes, N, Cin, Cout, S, stream);
}

// backward of dram_conv3d_k1_fwd_lazy: dx is the gradient w.r.t. the ACTIVATED input act(coef * x)
extern "C" int dram_conv3d_k1_bwd_lazy(const float* dy, const float* x, const float* coef, int relu, const float* w, float* dx,
                                       float* dw, float* dbias, void* ws, size_t ws_bytes, int N, int Cin, int Cout, int64_t S,
                                       void* stream) {
    return conv1x1_bwd_run(dy, x, coef, relu, w, dx, dw, dbias, ws, ws_bytes, N, Cin, Cout, S, stream);
}

extern "C" size_t dram_channel_sum_ws_bytes(int N, int C, int64_t S) {
    if (N <= 0 || C <= 0 || S <= 0) return 0;
    return (size_t)N * C * cdiv64(S, CS_CHUNK) * sizeof(float);
}

extern "C" int dram_channel_sum(const float* dy, float* dbias, void* ws, size_t ws_bytes, int N, int C, int64_t S,
                                void* stream) {
    DRAM_REQUIRE(dy && dbias && ws, "channel_sum: null pointer");
    DRAM_REQUIRE(N > 0 && C > 0 && S > 0 && (int64_t)N * C <= 65535, "channel_sum: bad dimensions");
    if (ws_bytes < dram_channel_sum_ws_bytes(N, C, S)) {
        set_error("channel_sum: workspace too small");
        return DRAM_EWS;
    }
    const int nch = (int)cdiv64(S, CS_CHUNK);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(row_sum_kernel, dim3(nch, N * C), dim3(256), 0, st, dy, (float*)ws, S, nch);
    hipLaunchKernelGGL(channel_sum_finalize_kernel, dim3(cdiv(C, 64)), dim3(64), 0, st, (const float*)ws, dbias, N, C, nch);
    return check_launch("channel_sum");
}

extern "C" size_t dram_masked_mean_ws_bytes(int N, int C, int64_t S) {
    if (N <= 0 || C <= 0 || S <= 0) return 0;
    return (size_t)N * C * cdiv64(S, CS_CHUNK) * 2 * sizeof(float);
}

extern "C" int dram_masked_mean_fwd(const float* x, const float* mask, float* out, float* msum, void* ws,
                                    size_t ws_bytes, int N, int C, int64_t S, void* stream) {
    DRAM_REQUIRE(x && mask && out && msum && ws, "masked_mean_fwd: null pointer");
    DRAM_REQUIRE(N > 0 && C > 0 && S > 0 && (int64_t)N * C <= 65535, "masked_mean_fwd: bad dimensions");
    if (ws_bytes < dram_masked_mean_ws_bytes(N, C, S)) {
        set_error("masked_mean_fwd: workspace too small");
        return DRAM_EWS;
    }
    const int nch = (int)cdiv64(S, CS_CHUNK);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(masked_sum_kernel, dim3(nch, N * C), dim3(256), 0, st, x, mask, (float*)ws, C, S, nch);
    hipLaunchKernelGGL(masked_mean_finalize_kernel, dim3(cdiv(N * C, 64)), dim3(64), 0, st, (const float*)ws, out, msum, N, C, nch);
    return check_launch("masked_mean_fwd");
}

extern "C" int dram_masked_mean_bwd(const float* dout, const float* mask, const float* msum, float* dx, int N, int C,
                                    int64_t S, void* stream) {
    DRAM_REQUIRE(dout && mask && msum && dx, "masked_mean_bwd: null pointer");
    DRAM_REQUIRE(N > 0 && C > 0 && S > 0 && (int64_t)N * C <= 65535, "masked_mean_bwd: bad dimensions");
    const unsigned gx = (unsigned)(cdiv64(S, 256) < 1024 ? cdiv64(S, 256) : 1024);
    hipLaunchKernelGGL(masked_mean_bwd_kernel, dim3(gx, N * C), dim3(256), 0, (hipStream_t)stream, dout, mask, msum, dx, C, S);
    return check_launch("masked_mean_bwd");
}
