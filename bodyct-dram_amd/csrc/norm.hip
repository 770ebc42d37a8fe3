// BatchNorm3d / GroupNorm (+ fused ReLU), forward and backward, fp32 NCDHW, gfx950.
//
// Replaces the ATen batch_norm / group_norm / relu_ dispatches (and their backward
// kernels) behind normal_wrapper and act_wrapper, reference dram/parts.py:17-35,48-54.
//
// All of these are HBM-bound.  In NCDHW one (n,c) "row" is S = D*H*W contiguous
// floats, and both norm families reduce over whole rows:
//    BatchNorm: statistic c      <- rows {(n,c) : n}
//    GroupNorm: statistic (n,g)  <- rows {(n,c) : c in group g}
// so one set of row kernels serves both:
//    row_moments      x            -> per (row, chunk) {mean, M2}      (1 read)
//    *_finalize       chunk stats  -> mean/rstd per statistic, per-row {a,b}   (tiny, fp64 Chan combine)
//    row_affine_act   y = act(a*x+b)                                   (1 read + 1 write)
//    row_bwd_reduce   per (row, chunk) {sum dy', sum dy'*xhat}          (2 reads)
//    *_bwd_finalize   -> dgamma, dbeta, per-row {p,q,r}                 (tiny)
//    row_bwd_apply    dx = p*dy' + q*x + r                              (2 reads + 1 write)
// with dy' = dy * [a*x+b > 0] when the ReLU is fused (the mask is recomputed from x,
// so the activated output never has to be read back).
//
// Numerics: chunk statistics are computed two-pass from registers (mean, then sum of
// squared deviations), and combined with Chan's formula in fp64 -> no E[x^2]-E[x]^2
// cancellation even at N*S = 1.3e8 elements per channel (SURVEY "BN at scale").
#include "common.h"

namespace dram {

#ifndef DRAM_NORM_Q
#define DRAM_NORM_Q 4
#endif
constexpr int NQ = DRAM_NORM_Q;       // float4 per thread and tensor, all in flight together
constexpr int CHUNK = 256 * 4 * NQ;   // floats per (row, chunk) work item: 256 threads x NQ float4

// Streaming 16-byte accesses of the row kernels: non-temporal (every tensor here is far larger than the caches and is touched once
// per pass), four per thread and tensor in flight.  Measured on [16,64,128^3] (scripts/bench_norm.py, same-box A/B of builds):
// dram_norm_bwd 7.48 ms with default-policy accesses and eight per thread, 7.26 non-temporal, 7.11 non-temporal with four (7.22
// with two); dram_row_affine_act 3.22 / 3.04 / 2.91 / 2.87 ms.  -DDRAM_NORM_TEMPORAL restores the default cache policy.
#ifndef DRAM_NORM_TEMPORAL
typedef float norm_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4(const float* p) {
    const norm_f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const norm_f32x4*>(p));
    return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void st4(float* p, float4 v) {
    const norm_f32x4 w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, reinterpret_cast<norm_f32x4*>(p));
}
#else
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
#endif

// grid: (nchunks, rows).  part[(row*nchunks + chunk)*2] = {mean, M2} of that chunk.
template <bool VEC>
__global__ __launch_bounds__(256) void row_moments_kernel(const float* __restrict__ x, float* __restrict__ part,
                                                          int64_t S, int nchunks) {
    __shared__ float red[4];
    const int64_t row = blockIdx.y;
    const int chunk = blockIdx.x;
    const int64_t beg = (int64_t)chunk * CHUNK;
    const int len = (int)((S - beg) < CHUNK ? (S - beg) : CHUNK);
    const float* p = x + row * S + beg;
    float v[32];
    int cnt = 0;
    float s = 0.f;
    if (VEC) {
        // All 8 loads first, from clamped (always valid) addresses, and only then the arithmetic: a load inside
        // `if (e < len)` makes hipcc emit branch + load + s_waitcnt vmcnt(0) per slot, i.e. 8 serialised memory
        // latencies per thread (measured: 2.5 TB/s; 6.2 without the guards).  len % 4 == 0 on this path.
        float4 t[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int e = (q * 256 + threadIdx.x) * 4;
            t[q] = ld4(p + (e < len ? e : 0));
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int e = (q * 256 + threadIdx.x) * 4;
            const bool in = e < len;
            v[4 * q] = in ? t[q].x : 0.f; v[4 * q + 1] = in ? t[q].y : 0.f;
            v[4 * q + 2] = in ? t[q].z : 0.f; v[4 * q + 3] = in ? t[q].w : 0.f;
            s += (v[4 * q] + v[4 * q + 1]) + (v[4 * q + 2] + v[4 * q + 3]);
        }
    } else {
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            const int e = q * 256 + threadIdx.x;
            v[q] = e < len ? p[e] : 0.f;
            s += v[q];
        }
    }
    (void)cnt;
    const float mean = block_sum_256(s, red) / (float)len;
    float m2 = 0.f;
    if (VEC) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int e = (q * 256 + threadIdx.x) * 4;
            if (e < len) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float d = v[4 * q + u] - mean;
                    m2 += d * d;
                }
            }
        }
    } else {
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            const int e = q * 256 + threadIdx.x;
            if (e < len) {
                const float d = v[q] - mean;
                m2 += d * d;
            }
        }
    }
    m2 = block_sum_256(m2, red);
    if (threadIdx.x == 0) {
        float* o = part + ((size_t)row * nchunks + chunk) * 2;
        o[0] = mean;
        o[1] = m2;
    }
}

__device__ __forceinline__ int chunk_len(int64_t S, int chunk) {
    const int64_t beg = (int64_t)chunk * CHUNK;
    return (int)((S - beg) < CHUNK ? (S - beg) : CHUNK);
}

// One block per statistic.  BatchNorm: stat = c, members = rows (n,c) for all n.
// GroupNorm: stat = (n,g), members = rows (n, g*cpg .. (g+1)*cpg-1).
// Chan combine over members x chunks in fp64 (two sweeps: mean, then M2).
__global__ __launch_bounds__(256) void norm_finalize_kernel(const float* __restrict__ part, int nchunks, int64_t S,
                                                            int kind, int N, int C, int G,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps,
                                                            float* __restrict__ save_mean,
                                                            float* __restrict__ save_rstd,
                                                            float* __restrict__ rowcoef, float* running_mean,
                                                            float* running_var, float momentum) {
    __shared__ double dred[4];
    __shared__ double bc[2];
    const int stat = blockIdx.x;
    const int cpg = C / G;
    int nmem;     // member rows
    int64_t row0; // first row
    int64_t rstride;
    if (kind == DRAM_NORM_BATCH) { nmem = N; row0 = stat; rstride = C; }
    else { nmem = cpg; row0 = (int64_t)(stat / G) * C + (int64_t)(stat % G) * cpg; rstride = 1; }
    const int64_t items = (int64_t)nmem * nchunks;
    const double total = (double)nmem * (double)S;

    double s = 0.0;
    for (int64_t it = threadIdx.x; it < items; it += 256) {
        const int64_t m = it / nchunks;
        const int ch = (int)(it % nchunks);
        const float* o = part + ((size_t)(row0 + m * rstride) * nchunks + ch) * 2;
        s += (double)o[0] * (double)chunk_len(S, ch);
    }
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) dred[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) bc[0] = (dred[0] + dred[1] + dred[2] + dred[3]) / total;
    __syncthreads();
    const double mean = bc[0];
    double m2 = 0.0;
    for (int64_t it = threadIdx.x; it < items; it += 256) {
        const int64_t m = it / nchunks;
        const int ch = (int)(it % nchunks);
        const float* o = part + ((size_t)(row0 + m * rstride) * nchunks + ch) * 2;
        const double d = (double)o[0] - mean;
        m2 += (double)o[1] + d * d * (double)chunk_len(S, ch);
    }
    m2 = wave_sum_d(m2);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) dred[threadIdx.x >> 6] = m2;
    __syncthreads();
    if (threadIdx.x == 0) bc[1] = (dred[0] + dred[1] + dred[2] + dred[3]);
    __syncthreads();
    const double var = bc[1] / total;  // biased
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float meanf = (float)mean;
    if (threadIdx.x == 0) {
        save_mean[stat] = meanf;
        save_rstd[stat] = rstd;
        if (kind == DRAM_NORM_BATCH && running_mean) {
            const double unb = total > 1.0 ? bc[1] / (total - 1.0) : var;
            running_mean[stat] = (1.f - momentum) * running_mean[stat] + momentum * meanf;
            running_var[stat] = (1.f - momentum) * running_var[stat] + momentum * (float)unb;
        }
    }
    // per-row coefficients y_pre = a*x + b
    for (int m = threadIdx.x; m < nmem; m += 256) {
        const int64_t row = row0 + (int64_t)m * rstride;
        const int c = (int)(row % C);
        const float g = gamma ? gamma[c] : 1.f;
        const float bt = beta ? beta[c] : 0.f;
        const float a = g * rstd;
        rowcoef[2 * row] = a;
        rowcoef[2 * row + 1] = bt - meanf * a;
    }
}

// eval-mode BatchNorm coefficients from the running statistics
__global__ void bn_eval_coef_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                    const float* __restrict__ rm, const float* __restrict__ rv, float eps, int N,
                                    int C, float* __restrict__ save_mean, float* __restrict__ save_rstd,
                                    float* __restrict__ rowcoef) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= N * C) return;
    const int c = row % C;
    const float rstd = 1.f / sqrtf(rv[c] + eps);
    const float a = (gamma ? gamma[c] : 1.f) * rstd;
    rowcoef[2 * row] = a;
    rowcoef[2 * row + 1] = (beta ? beta[c] : 0.f) - rm[c] * a;
    if (row < C) {
        save_mean[c] = rm[c];
        save_rstd[c] = rstd;
    }
}

// y = act(a*x + b); grid (nchunks, rows)
template <bool VEC>
__global__ __launch_bounds__(256) void row_affine_act_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                             const float* __restrict__ rowcoef, int64_t S, int relu) {
    const int64_t row = blockIdx.y;
    const int64_t beg = (int64_t)blockIdx.x * CHUNK;
    const int len = (int)((S - beg) < CHUNK ? (S - beg) : CHUNK);
    const float a = rowcoef[2 * row], b = rowcoef[2 * row + 1];
    const float* p = x + row * S + beg;
    float* o = y + row * S + beg;
    if (VEC) {
        float4 tv[NQ];       // loads first, unconditional (see row_moments_kernel)
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int e = (q * 256 + threadIdx.x) * 4;
            tv[q] = ld4(p + (e < len ? e : 0));
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int e = (q * 256 + threadIdx.x) * 4;
            float4 t = tv[q];
            t.x = fmaf(a, t.x, b); t.y = fmaf(a, t.y, b); t.z = fmaf(a, t.z, b); t.w = fmaf(a, t.w, b);
            if (relu) { t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f); t.z = fmaxf(t.z, 0.f); t.w = fmaxf(t.w, 0.f); }
            if (e < len) st4(o + e, t);
        }
    } else {
        for (int e = threadIdx.x; e < len; e += 256) {
            float t = fmaf(a, p[e], b);
            if (relu) t = fmaxf(t, 0.f);
            o[e] = t;
        }
    }
}

// per (row, chunk): {sum dy', sum dy' * xhat}, xhat = (x - mean)*rstd of the row's statistic
template <bool VEC>
__global__ __launch_bounds__(256) void row_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                             const float* __restrict__ rowcoef,
                                                             const float* __restrict__ save_mean,
                                                             const float* __restrict__ save_rstd,
                                                             float* __restrict__ part, int64_t S, int nchunks,
                                                             int kind, int C, int G, int relu) {
    __shared__ float red[4];
    const int64_t row = blockIdx.y;
    const int chunk = blockIdx.x;
    const int64_t beg = (int64_t)chunk * CHUNK;
    const int len = (int)((S - beg) < CHUNK ? (S - beg) : CHUNK);
    const int c = (int)(row % C), n = (int)(row / C);
    const int stat = kind == DRAM_NORM_BATCH ? c : n * G + c / (C / G);
    const float mean = save_mean[stat], rstd = save_rstd[stat];
    const float a = rowcoef[2 * row], b = rowcoef[2 * row + 1];
    const float* px = x + row * S + beg;
    const float* pd = dy + row * S + beg;
    float s1 = 0.f, s2 = 0.f;
    if (VEC) {
        float4 xq[NQ], dq[NQ];    // loads first, unconditional (see row_moments_kernel)
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int e = (q * 256 + threadIdx.x) * 4;
            const int ec = e < len ? e : 0;
            xq[q] = ld4(px + ec);
            dq[q] = ld4(pd + ec);
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int e = (q * 256 + threadIdx.x) * 4;
            const bool in = e < len;
            const float xs[4] = {xq[q].x, xq[q].y, xq[q].z, xq[q].w};
            const float ds[4] = {dq[q].x, dq[q].y, dq[q].z, dq[q].w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float d = in ? ds[u] : 0.f;
                if (relu && !(fmaf(a, xs[u], b) > 0.f)) d = 0.f;
                s1 += d;
                s2 += d * ((xs[u] - mean) * rstd);
            }
        }
    } else {
        for (int e = threadIdx.x; e < len; e += 256) {
            float d = pd[e];
            const float xv = px[e];
            if (relu && !(fmaf(a, xv, b) > 0.f)) d = 0.f;
            s1 += d;
            s2 += d * ((xv - mean) * rstd);
        }
    }
    s1 = block_sum_256(s1, red);
    s2 = block_sum_256(s2, red);
    if (threadIdx.x == 0) {
        float* o = part + ((size_t)row * nchunks + chunk) * 2;
        o[0] = s1;
        o[1] = s2;
    }
}

// rowsum[row] = {sum over chunks of s1, s2} in fp64 -> stored as 2 doubles
__global__ void row_sum_chunks_kernel(const float* __restrict__ part, double* __restrict__ rowsum, int rows,
                                      int nchunks) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows) return;
    double s1 = 0.0, s2 = 0.0;
    const float* o = part + (size_t)row * nchunks * 2;
    for (int ch = 0; ch < nchunks; ++ch) { s1 += o[2 * ch]; s2 += o[2 * ch + 1]; }
    rowsum[2 * row] = s1;
    rowsum[2 * row + 1] = s2;
}

// One thread per channel: dgamma[c] = sum_n s2[n,c], dbeta[c] = sum_n s1[n,c]; BatchNorm also emits the
// per-row dx coefficients {p,q,r}: dx = p*dy' + q*x + r.
__global__ void bn_bwd_finalize_kernel(const double* __restrict__ rowsum, const float* __restrict__ gamma,
                                       const float* __restrict__ save_mean, const float* __restrict__ save_rstd,
                                       float* __restrict__ dgamma, float* __restrict__ dbeta,
                                       float* __restrict__ pqr, int N, int C, int64_t S, int batch_stats) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s1 = 0.0, s2 = 0.0;
    for (int n = 0; n < N; ++n) {
        s1 += rowsum[2 * ((size_t)n * C + c)];
        s2 += rowsum[2 * ((size_t)n * C + c) + 1];
    }
    if (dgamma) dgamma[c] = (float)s2;
    if (dbeta) dbeta[c] = (float)s1;
    const double g = gamma ? (double)gamma[c] : 1.0;
    const double rstd = save_rstd[c], mean = save_mean[c];
    const double M = (double)N * (double)S;
    const double k1 = g * rstd;
    double k2 = 0.0, k3 = 0.0;
    if (batch_stats) { k2 = -k1 * s2 / M; k3 = -k1 * s1 / M; }
    // dx = k1*dy' + k2*xhat + k3, xhat = (x-mean)*rstd
    const float p = (float)k1, q = (float)(k2 * rstd), r = (float)(k3 - k2 * rstd * mean);
    for (int n = 0; n < N; ++n) {
        float* o = pqr + 3 * ((size_t)n * C + c);
        o[0] = p; o[1] = q; o[2] = r;
    }
}

// One thread per (n,g): A = sum_c gamma*s1, B = sum_c gamma*s2 over the group's channels.
__global__ void gn_bwd_finalize_rows_kernel(const double* __restrict__ rowsum, const float* __restrict__ gamma,
                                            const float* __restrict__ save_mean, const float* __restrict__ save_rstd,
                                            float* __restrict__ pqr, int N, int C, int G, int64_t S) {
    const int stat = blockIdx.x * blockDim.x + threadIdx.x;
    if (stat >= N * G) return;
    const int n = stat / G, g = stat % G, cpg = C / G;
    double A = 0.0, B = 0.0;
    for (int cc = 0; cc < cpg; ++cc) {
        const int c = g * cpg + cc;
        const double gm = gamma ? (double)gamma[c] : 1.0;
        A += gm * rowsum[2 * ((size_t)n * C + c)];
        B += gm * rowsum[2 * ((size_t)n * C + c) + 1];
    }
    const double rstd = save_rstd[stat], mean = save_mean[stat];
    const double M = (double)cpg * (double)S;
    const double k2 = -rstd * B / M, k3 = -rstd * A / M;
    for (int cc = 0; cc < cpg; ++cc) {
        const int c = g * cpg + cc;
        const double gm = gamma ? (double)gamma[c] : 1.0;
        float* o = pqr + 3 * ((size_t)n * C + c);
        o[0] = (float)(rstd * gm);
        o[1] = (float)(k2 * rstd);
        o[2] = (float)(k3 - k2 * rstd * mean);
    }
}

__global__ void gn_bwd_finalize_params_kernel(const double* __restrict__ rowsum, float* __restrict__ dgamma,
                                              float* __restrict__ dbeta, int N, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s1 = 0.0, s2 = 0.0;
    for (int n = 0; n < N; ++n) {
        s1 += rowsum[2 * ((size_t)n * C + c)];
        s2 += rowsum[2 * ((size_t)n * C + c) + 1];
    }
    if (dgamma) dgamma[c] = (float)s2;
    if (dbeta) dbeta[c] = (float)s1;
}

// dx = p*dy' + q*x + r
template <bool VEC>
__global__ __launch_bounds__(256) void row_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ rowcoef,
                                                            const float* __restrict__ pqr, float* __restrict__ dx,
                                                            int64_t S, int relu) {
    const int64_t row = blockIdx.y;
    const int64_t beg = (int64_t)blockIdx.x * CHUNK;
    const int len = (int)((S - beg) < CHUNK ? (S - beg) : CHUNK);
    const float a = rowcoef[2 * row], b = rowcoef[2 * row + 1];
    const float p = pqr[3 * row], qq = pqr[3 * row + 1], r = pqr[3 * row + 2];
    const float* px = x + row * S + beg;
    const float* pd = dy + row * S + beg;
    float* o = dx + row * S + beg;
    if (VEC) {
        float4 xq[NQ], dq[NQ];    // loads first, unconditional (see row_moments_kernel)
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int e = (q * 256 + threadIdx.x) * 4;
            const int ec = e < len ? e : 0;
            xq[q] = ld4(px + ec);
            dq[q] = ld4(pd + ec);
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int e = (q * 256 + threadIdx.x) * 4;
            const float xs[4] = {xq[q].x, xq[q].y, xq[q].z, xq[q].w};
            float ds[4] = {dq[q].x, dq[q].y, dq[q].z, dq[q].w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (relu && !(fmaf(a, xs[u], b) > 0.f)) ds[u] = 0.f;
                ds[u] = fmaf(p, ds[u], fmaf(qq, xs[u], r));
            }
            if (e < len) st4(o + e, make_float4(ds[0], ds[1], ds[2], ds[3]));
        }
    } else {
        for (int e = threadIdx.x; e < len; e += 256) {
            float d = pd[e];
            const float xv = px[e];
            if (relu && !(fmaf(a, xv, b) > 0.f)) d = 0.f;
            o[e] = fmaf(p, d, fmaf(qq, xv, r));
        }
    }
}

__global__ void relu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = i; e < n; e += stride) y[e] = fmaxf(x[e], 0.f);
}
__global__ void relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx,
                                int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = i; e < n; e += stride) dx[e] = y[e] > 0.f ? dy[e] : 0.f;
}

// ---- cross-rank BatchNorm ("sbn"): the statistics / the backward sums leave the device between two stages ----
// per channel: mean and M2 (sum of squared deviations) of the local rows, fp64 (Chan combine over rows x chunks)
__global__ __launch_bounds__(256) void bn_stats_finalize_kernel(const float* __restrict__ part, int nchunks, int64_t S, int N,
                                                                int C, double* __restrict__ mean_m2) {
    __shared__ double dred[4];
    __shared__ double bc;
    const int c = blockIdx.x;
    const int64_t items = (int64_t)N * nchunks;
    const double total = (double)N * (double)S;
    double s = 0.0;
    for (int64_t it = threadIdx.x; it < items; it += 256) {
        const int64_t n = it / nchunks;
        const int ch = (int)(it % nchunks);
        s += (double)part[((size_t)(n * C + c) * nchunks + ch) * 2] * (double)chunk_len(S, ch);
    }
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) dred[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) bc = (dred[0] + dred[1] + dred[2] + dred[3]) / total;
    __syncthreads();
    const double mean = bc;
    double m2 = 0.0;
    for (int64_t it = threadIdx.x; it < items; it += 256) {
        const int64_t n = it / nchunks;
        const int ch = (int)(it % nchunks);
        const float* o = part + ((size_t)(n * C + c) * nchunks + ch) * 2;
        const double d = (double)o[0] - mean;
        m2 += (double)o[1] + d * d * (double)chunk_len(S, ch);
    }
    m2 = wave_sum_d(m2);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) dred[threadIdx.x >> 6] = m2;
    __syncthreads();
    if (threadIdx.x == 0) {
        mean_m2[2 * c] = mean;
        mean_m2[2 * c + 1] = dred[0] + dred[1] + dred[2] + dred[3];
    }
}

// sums[2c] = sum_n s1[n,c], sums[2c+1] = sum_n s2[n,c]
__global__ void bn_bwd_sums_kernel(const double* __restrict__ rowsum, double* __restrict__ sums, int N, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s1 = 0.0, s2 = 0.0;
    for (int n = 0; n < N; ++n) {
        s1 += rowsum[2 * ((size_t)n * C + c)];
        s2 += rowsum[2 * ((size_t)n * C + c) + 1];
    }
    sums[2 * c] = s1;
    sums[2 * c + 1] = s2;
}

// per-row dx coefficients from (global) sums over `count` elements per channel
__global__ void bn_pqr_from_sums_kernel(const double* __restrict__ sums, double count, const float* __restrict__ gamma,
                                        const float* __restrict__ save_mean, const float* __restrict__ save_rstd,
                                        float* __restrict__ pqr, int N, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double g = gamma ? (double)gamma[c] : 1.0;
    const double rstd = save_rstd[c], mean = save_mean[c];
    const double k1 = g * rstd;
    const double k2 = -k1 * sums[2 * c + 1] / count, k3 = -k1 * sums[2 * c] / count;
    const float p = (float)k1, q = (float)(k2 * rstd), r = (float)(k3 - k2 * rstd * mean);
    for (int n = 0; n < N; ++n) {
        float* o = pqr + 3 * ((size_t)n * C + c);
        o[0] = p; o[1] = q; o[2] = r;
    }
}


// ---- statistics from the conv epilogue (csrc/conv3d_k3.hip stats_epilogue): per (row, part) {mean, M2, count} ----
// Chan's parallel update of (count, mean, M2) in fp64.
struct Moments {
    double n, mean, m2;
};
__device__ __forceinline__ Moments chan_merge(Moments a, Moments b) {
    if (b.n == 0.0) return a;
    if (a.n == 0.0) return b;
    Moments r;
    r.n = a.n + b.n;
    const double d = b.mean - a.mean;
    r.mean = a.mean + d * (b.n / r.n);
    r.m2 = a.m2 + b.m2 + d * d * (a.n * b.n / r.n);
    return r;
}
__device__ __forceinline__ Moments wave_merge(Moments m) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        Moments b;
        b.n = __shfl_xor(m.n, o, 64);
        b.mean = __shfl_xor(m.mean, o, 64);
        b.m2 = __shfl_xor(m.m2, o, 64);
        // both lanes of a pair must get the same bits: merge in a fixed (low lane, high lane) order
        const bool low = ((threadIdx.x & 63) & o) == 0;
        m = low ? chan_merge(m, b) : chan_merge(b, m);
    }
    return m;
}
// block-wide merge for 256 threads; result valid in thread 0
__device__ __forceinline__ Moments block_merge_256(Moments m, double* sh /* 12 doubles */) {
    m = wave_merge(m);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { sh[3 * w] = m.n; sh[3 * w + 1] = m.mean; sh[3 * w + 2] = m.m2; }
    __syncthreads();
    Moments r = {sh[0], sh[1], sh[2]};
#pragma unroll
    for (int k = 1; k < 4; ++k) r = chan_merge(r, Moments{sh[3 * k], sh[3 * k + 1], sh[3 * k + 2]});
    return r;
}

// level 1: grid (groups, rows): group g of a row merges parts [g*per, (g+1)*per) -> part2[(row*groups + g)*3] (doubles)
constexpr int PARTS_PER_GROUP = 2048;
__global__ __launch_bounds__(256) void parts_reduce_kernel(const float* __restrict__ part, double* __restrict__ part2,
                                                           int nparts, int groups) {
    __shared__ double sh[12];
    const int64_t row = blockIdx.y;
    const int g = blockIdx.x;
    const int beg = g * PARTS_PER_GROUP;
    const int end = (beg + PARTS_PER_GROUP) < nparts ? (beg + PARTS_PER_GROUP) : nparts;
    const float* p = part + (row * (int64_t)nparts) * 3;
    Moments m = {0.0, 0.0, 0.0};
    for (int i = beg + threadIdx.x; i < end; i += 256) {
        const Moments b = {(double)p[3 * (int64_t)i + 2], (double)p[3 * (int64_t)i], (double)p[3 * (int64_t)i + 1]};
        m = chan_merge(m, b);
    }
    m = block_merge_256(m, sh);
    if (threadIdx.x == 0) {
        double* o = part2 + (row * groups + g) * 3;
        o[0] = m.n; o[1] = m.mean; o[2] = m.m2;
    }
}

// level 2: one block per statistic (as norm_finalize_kernel), members x groups triples
__global__ __launch_bounds__(256) void norm_finalize_parts_kernel(const double* __restrict__ part2, int groups, int kind, int N,
                                                                  int C, int G, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, float eps,
                                                                  float* __restrict__ save_mean, float* __restrict__ save_rstd,
                                                                  float* __restrict__ rowcoef, float* running_mean,
                                                                  float* running_var, float momentum, double expect_count) {
    __shared__ double sh[12];
    __shared__ double bc[3];
    const int stat = blockIdx.x;
    const int cpg = C / G;
    int nmem;
    int64_t row0, rstride;
    if (kind == DRAM_NORM_BATCH) { nmem = N; row0 = stat; rstride = C; }
    else { nmem = cpg; row0 = (int64_t)(stat / G) * C + (int64_t)(stat % G) * cpg; rstride = 1; }
    const int64_t items = (int64_t)nmem * groups;
    Moments m = {0.0, 0.0, 0.0};
    for (int64_t it = threadIdx.x; it < items; it += 256) {
        const int64_t mem = it / groups;
        const int g = (int)(it % groups);
        const double* o = part2 + ((row0 + mem * rstride) * groups + g) * 3;
        m = chan_merge(m, Moments{o[0], o[1], o[2]});
    }
    m = block_merge_256(m, sh);
    if (threadIdx.x == 0) { bc[0] = m.n; bc[1] = m.mean; bc[2] = m.m2; }
    __syncthreads();
    const double total = bc[0] > 0.0 ? bc[0] : 1.0;
    const double mean = bc[1];
    const double var = bc[2] / total;  // biased
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float meanf = (float)mean;
    if (threadIdx.x == 0) {
        // a count that differs from members x spatial size means a partial went missing: poison instead of a silent bias
        const bool bad = expect_count > 0.0 && bc[0] != expect_count;
        save_mean[stat] = bad ? NAN : meanf;
        save_rstd[stat] = bad ? NAN : rstd;
        if (kind == DRAM_NORM_BATCH && running_mean) {
            const double unb = total > 1.0 ? bc[2] / (total - 1.0) : var;
            running_mean[stat] = (1.f - momentum) * running_mean[stat] + momentum * meanf;
            running_var[stat] = (1.f - momentum) * running_var[stat] + momentum * (float)unb;
        }
    }
    for (int mm = threadIdx.x; mm < nmem; mm += 256) {
        const int64_t row = row0 + (int64_t)mm * rstride;
        const int c = (int)(row % C);
        const float g = gamma ? gamma[c] : 1.f;
        const float bt = beta ? beta[c] : 0.f;
        const float a = g * rstd;
        rowcoef[2 * row] = a;
        rowcoef[2 * row + 1] = bt - meanf * a;
    }
}

// per channel {mean, M2} in fp64 over the N rows of the channel (BatchNorm), from the level-1 triples: what a cross-rank
// combine (nn.SyncBatchNorm, parts.py:32-33) gathers -- the same layout dram_bn_stats writes
__global__ __launch_bounds__(256) void bn_parts_moments_kernel(const double* __restrict__ part2, int groups, int N, int C,
                                                               double expect_count, double* __restrict__ mean_m2) {
    __shared__ double sh[12];
    const int c = blockIdx.x;
    const int64_t items = (int64_t)N * groups;
    Moments m = {0.0, 0.0, 0.0};
    for (int64_t it = threadIdx.x; it < items; it += 256) {
        const int64_t n = it / groups;
        const int g = (int)(it % groups);
        const double* o = part2 + ((n * C + c) * groups + g) * 3;
        m = chan_merge(m, Moments{o[0], o[1], o[2]});
    }
    m = block_merge_256(m, sh);
    if (threadIdx.x == 0) {
        const bool bad = m.n != expect_count;          // a partial went missing: poison instead of a silent bias
        mean_m2[2 * c] = bad ? NAN : m.mean;
        mean_m2[2 * c + 1] = bad ? NAN : m.m2;
    }
}

static inline bool vec_ok(const void* p, int64_t S) { return (S % 4 == 0) && (((uintptr_t)p) % 16 == 0); }

static int check_norm(const char* who, int kind, int G, int N, int C, int64_t S) {
    DRAM_REQUIRE(N > 0 && C > 0 && S > 0, "%s: non-positive dimension", who);
    DRAM_REQUIRE(kind == DRAM_NORM_BATCH || kind == DRAM_NORM_GROUP, "%s: unknown norm kind %d", who, kind);
    if (kind == DRAM_NORM_GROUP) DRAM_REQUIRE(G > 0 && C % G == 0, "%s: C=%d not divisible by G=%d", who, C, G);
    DRAM_REQUIRE((int64_t)N * C <= 65535 * 64LL, "%s: too many rows", who);
    return DRAM_OK;
}

static inline int nchunks_of(int64_t S) { return (int)cdiv64(S, CHUNK); }

// rows go on grid.y (max 65535): fold when N*C is larger
static inline dim3 row_grid(int nchunks, int64_t rows) { return dim3((unsigned)nchunks, (unsigned)rows); }

}  // namespace dram

using namespace dram;

extern "C" size_t dram_norm_ws_bytes(int N, int C, int64_t S) {
    if (N <= 0 || C <= 0 || S <= 0) return 0;
    const size_t rows = (size_t)N * C;
    // chunk partials (2 floats) + row sums (2 doubles) + pqr (3 floats)
    return align_up(rows * nchunks_of(S) * 2 * sizeof(float), 256) + align_up(rows * 2 * sizeof(double), 256) +
           align_up(rows * 3 * sizeof(float), 256);
}

extern "C" int dram_norm_fwd_train(const float* x, const float* gamma, const float* beta, float* y, float* save_mean,
                                   float* save_rstd, float* rowcoef, float* running_mean, float* running_var,
                                   float momentum, float eps, int kind, int G, int relu, int N, int C, int64_t S,
                                   void* ws, size_t ws_bytes, void* stream) {
    DRAM_REQUIRE(x && y && save_mean && save_rstd && rowcoef && ws, "norm_fwd_train: null pointer");
    int rc = check_norm("norm_fwd_train", kind, G, N, C, S);
    if (rc) return rc;
    DRAM_REQUIRE((int64_t)N * C <= 65535, "norm_fwd_train: N*C > 65535 rows not supported");
    if (ws_bytes < dram_norm_ws_bytes(N, C, S)) {
        set_error("norm_fwd_train: workspace too small");
        return DRAM_EWS;
    }
    hipStream_t st = (hipStream_t)stream;
    const int nch = nchunks_of(S);
    const int64_t rows = (int64_t)N * C;
    float* part = (float*)ws;
    const bool vec = vec_ok(x, S) && vec_ok(y, S);
    if (vec) hipLaunchKernelGGL(row_moments_kernel<true>, row_grid(nch, rows), dim3(256), 0, st, x, part, S, nch);
    else hipLaunchKernelGGL(row_moments_kernel<false>, row_grid(nch, rows), dim3(256), 0, st, x, part, S, nch);
    const int nstat = kind == DRAM_NORM_BATCH ? C : N * G;
    hipLaunchKernelGGL(norm_finalize_kernel, dim3(nstat), dim3(256), 0, st, part, nch, S, kind, N, C,
                       kind == DRAM_NORM_BATCH ? 1 : G, gamma, beta, eps, save_mean, save_rstd, rowcoef,
                       running_mean, running_var, momentum);
    if (vec) hipLaunchKernelGGL(row_affine_act_kernel<true>, row_grid(nch, rows), dim3(256), 0, st, x, y, rowcoef, S, relu);
    else hipLaunchKernelGGL(row_affine_act_kernel<false>, row_grid(nch, rows), dim3(256), 0, st, x, y, rowcoef, S, relu);
    return check_launch("norm_fwd_train");
}

extern "C" int dram_bn_fwd_eval(const float* x, const float* gamma, const float* beta, const float* running_mean,
                                const float* running_var, float* y, float* save_mean, float* save_rstd,
                                float* rowcoef, float eps, int relu, int N, int C, int64_t S, void* stream) {
    DRAM_REQUIRE(x && y && running_mean && running_var && save_mean && save_rstd && rowcoef, "bn_fwd_eval: null pointer");
    int rc = check_norm("bn_fwd_eval", DRAM_NORM_BATCH, 1, N, C, S);
    if (rc) return rc;
    DRAM_REQUIRE((int64_t)N * C <= 65535, "bn_fwd_eval: N*C > 65535 rows not supported");
    hipStream_t st = (hipStream_t)stream;
    const int nch = nchunks_of(S);
    const int64_t rows = (int64_t)N * C;
    hipLaunchKernelGGL(bn_eval_coef_kernel, dim3((unsigned)cdiv64(rows, 256)), dim3(256), 0, st, gamma, beta,
                       running_mean, running_var, eps, N, C, save_mean, save_rstd, rowcoef);
    if (vec_ok(x, S) && vec_ok(y, S))
        hipLaunchKernelGGL(row_affine_act_kernel<true>, row_grid(nch, rows), dim3(256), 0, st, x, y, rowcoef, S, relu);
    else
        hipLaunchKernelGGL(row_affine_act_kernel<false>, row_grid(nch, rows), dim3(256), 0, st, x, y, rowcoef, S, relu);
    return check_launch("bn_fwd_eval");
}

extern "C" int dram_norm_bwd(const float* dy, const float* x, const float* gamma, const float* save_mean,
                             const float* save_rstd, const float* rowcoef, float* dx, float* dgamma, float* dbeta,
                             int kind, int G, int relu, int batch_stats, int N, int C, int64_t S, void* ws,
                             size_t ws_bytes, void* stream) {
    DRAM_REQUIRE(dy && x && save_mean && save_rstd && rowcoef && dx && ws, "norm_bwd: null pointer");
    int rc = check_norm("norm_bwd", kind, G, N, C, S);
    if (rc) return rc;
    DRAM_REQUIRE((int64_t)N * C <= 65535, "norm_bwd: N*C > 65535 rows not supported");
    DRAM_REQUIRE(kind == DRAM_NORM_BATCH || batch_stats, "norm_bwd: GroupNorm always uses batch statistics");
    if (ws_bytes < dram_norm_ws_bytes(N, C, S)) {
        set_error("norm_bwd: workspace too small");
        return DRAM_EWS;
    }
    hipStream_t st = (hipStream_t)stream;
    const int nch = nchunks_of(S);
    const int64_t rows = (int64_t)N * C;
    char* w = (char*)ws;
    float* part = (float*)w;
    w += align_up((size_t)rows * nch * 2 * sizeof(float), 256);
    double* rowsum = (double*)w;
    w += align_up((size_t)rows * 2 * sizeof(double), 256);
    float* pqr = (float*)w;
    const int Gk = kind == DRAM_NORM_BATCH ? 1 : G;
    const bool vec = vec_ok(x, S) && vec_ok(dy, S) && vec_ok(dx, S);
    if (vec)
        hipLaunchKernelGGL(row_bwd_reduce_kernel<true>, row_grid(nch, rows), dim3(256), 0, st, dy, x, rowcoef,
                           save_mean, save_rstd, part, S, nch, kind, C, Gk, relu);
    else
        hipLaunchKernelGGL(row_bwd_reduce_kernel<false>, row_grid(nch, rows), dim3(256), 0, st, dy, x, rowcoef,
                           save_mean, save_rstd, part, S, nch, kind, C, Gk, relu);
    hipLaunchKernelGGL(row_sum_chunks_kernel, dim3((unsigned)cdiv64(rows, 256)), dim3(256), 0, st, part, rowsum,
                       (int)rows, nch);
    if (kind == DRAM_NORM_BATCH) {
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(C, 64)), dim3(64), 0, st, rowsum, gamma, save_mean,
                           save_rstd, dgamma, dbeta, pqr, N, C, S, batch_stats);
    } else {
        hipLaunchKernelGGL(gn_bwd_finalize_rows_kernel, dim3(cdiv(N * G, 64)), dim3(64), 0, st, rowsum, gamma,
                           save_mean, save_rstd, pqr, N, C, G, S);
        if (dgamma || dbeta)
            hipLaunchKernelGGL(gn_bwd_finalize_params_kernel, dim3(cdiv(C, 64)), dim3(64), 0, st, rowsum, dgamma,
                               dbeta, N, C);
    }
    if (vec)
        hipLaunchKernelGGL(row_bwd_apply_kernel<true>, row_grid(nch, rows), dim3(256), 0, st, dy, x, rowcoef, pqr, dx, S, relu);
    else
        hipLaunchKernelGGL(row_bwd_apply_kernel<false>, row_grid(nch, rows), dim3(256), 0, st, dy, x, rowcoef, pqr, dx, S, relu);
    return check_launch("norm_bwd");
}

extern "C" int dram_bn_stats(const float* x, double* mean_m2, int N, int C, int64_t S, void* ws, size_t ws_bytes,
                             void* stream) {
    DRAM_REQUIRE(x && mean_m2 && ws, "bn_stats: null pointer");
    int rc = check_norm("bn_stats", DRAM_NORM_BATCH, 1, N, C, S);
    if (rc) return rc;
    DRAM_REQUIRE((int64_t)N * C <= 65535, "bn_stats: N*C > 65535 rows not supported");
    if (ws_bytes < dram_norm_ws_bytes(N, C, S)) {
        set_error("bn_stats: workspace too small");
        return DRAM_EWS;
    }
    hipStream_t st = (hipStream_t)stream;
    const int nch = nchunks_of(S);
    const int64_t rows = (int64_t)N * C;
    float* part = (float*)ws;
    if (vec_ok(x, S)) hipLaunchKernelGGL(row_moments_kernel<true>, row_grid(nch, rows), dim3(256), 0, st, x, part, S, nch);
    else hipLaunchKernelGGL(row_moments_kernel<false>, row_grid(nch, rows), dim3(256), 0, st, x, part, S, nch);
    hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(C), dim3(256), 0, st, part, nch, S, N, C, mean_m2);
    return check_launch("bn_stats");
}

extern "C" int dram_bn_bwd_sums(const float* dy, const float* x, const float* save_mean, const float* save_rstd,
                                const float* rowcoef, double* sums, int relu, int N, int C, int64_t S, void* ws,
                                size_t ws_bytes, void* stream) {
    DRAM_REQUIRE(dy && x && save_mean && save_rstd && rowcoef && sums && ws, "bn_bwd_sums: null pointer");
    int rc = check_norm("bn_bwd_sums", DRAM_NORM_BATCH, 1, N, C, S);
    if (rc) return rc;
    DRAM_REQUIRE((int64_t)N * C <= 65535, "bn_bwd_sums: N*C > 65535 rows not supported");
    if (ws_bytes < dram_norm_ws_bytes(N, C, S)) {
        set_error("bn_bwd_sums: workspace too small");
        return DRAM_EWS;
    }
    hipStream_t st = (hipStream_t)stream;
    const int nch = nchunks_of(S);
    const int64_t rows = (int64_t)N * C;
    float* part = (float*)ws;
    double* rowsum = (double*)((char*)ws + align_up((size_t)rows * nch * 2 * sizeof(float), 256));
    if (vec_ok(x, S) && vec_ok(dy, S))
        hipLaunchKernelGGL(row_bwd_reduce_kernel<true>, row_grid(nch, rows), dim3(256), 0, st, dy, x, rowcoef, save_mean,
                           save_rstd, part, S, nch, DRAM_NORM_BATCH, C, 1, relu);
    else
        hipLaunchKernelGGL(row_bwd_reduce_kernel<false>, row_grid(nch, rows), dim3(256), 0, st, dy, x, rowcoef, save_mean,
                           save_rstd, part, S, nch, DRAM_NORM_BATCH, C, 1, relu);
    hipLaunchKernelGGL(row_sum_chunks_kernel, dim3((unsigned)cdiv64(rows, 256)), dim3(256), 0, st, part, rowsum, (int)rows, nch);
    hipLaunchKernelGGL(bn_bwd_sums_kernel, dim3(cdiv(C, 64)), dim3(64), 0, st, rowsum, sums, N, C);
    return check_launch("bn_bwd_sums");
}

extern "C" int dram_bn_bwd_apply_sums(const float* dy, const float* x, const float* gamma, const float* save_mean,
                                      const float* save_rstd, const float* rowcoef, const double* sums, double count,
                                      float* dx, int relu, int N, int C, int64_t S, void* ws, size_t ws_bytes,
                                      void* stream) {
    DRAM_REQUIRE(dy && x && save_mean && save_rstd && rowcoef && sums && dx && ws, "bn_bwd_apply_sums: null pointer");
    int rc = check_norm("bn_bwd_apply_sums", DRAM_NORM_BATCH, 1, N, C, S);
    if (rc) return rc;
    DRAM_REQUIRE((int64_t)N * C <= 65535 && count > 0.0, "bn_bwd_apply_sums: bad dimensions");
    if (ws_bytes < dram_norm_ws_bytes(N, C, S)) {
        set_error("bn_bwd_apply_sums: workspace too small");
        return DRAM_EWS;
    }
    hipStream_t st = (hipStream_t)stream;
    const int nch = nchunks_of(S);
    const int64_t rows = (int64_t)N * C;
    float* pqr = (float*)((char*)ws + align_up((size_t)rows * nch * 2 * sizeof(float), 256) +
                          align_up((size_t)rows * 2 * sizeof(double), 256));
    hipLaunchKernelGGL(bn_pqr_from_sums_kernel, dim3(cdiv(C, 64)), dim3(64), 0, st, sums, count, gamma, save_mean, save_rstd,
                       pqr, N, C);
    if (vec_ok(x, S) && vec_ok(dy, S) && vec_ok(dx, S))
        hipLaunchKernelGGL(row_bwd_apply_kernel<true>, row_grid(nch, rows), dim3(256), 0, st, dy, x, rowcoef, pqr, dx, S, relu);
    else
        hipLaunchKernelGGL(row_bwd_apply_kernel<false>, row_grid(nch, rows), dim3(256), 0, st, dy, x, rowcoef, pqr, dx, S, relu);
    return check_launch("bn_bwd_apply_sums");
}


// ---- statistics that arrive from the conv epilogue ---------------------------------------------------------------
static inline int parts_groups(int nparts) { return cdiv(nparts, PARTS_PER_GROUP); }

extern "C" size_t dram_norm_parts_ws_bytes(int N, int C, int nparts) {
    if (N <= 0 || C <= 0 || nparts <= 0) return 0;
    return (size_t)N * C * parts_groups(nparts) * 3 * sizeof(double);
}

// Training-mode statistics of BatchNorm / GroupNorm from the {mean, M2, count} partials that the fused convolution
// left per (row, part) (dram_conv3d_k3_fwd_fused): save_mean / save_rstd per statistic, the per-row {a, b} with
// y = a*x + b, running statistics updated as dram_norm_fwd_train does.  Nothing reads or writes the tensor itself.
extern "C" int dram_norm_finalize_parts(const float* parts, int nparts, const float* gamma, const float* beta,
                                        float* save_mean, float* save_rstd, float* rowcoef, float* running_mean,
                                        float* running_var, float momentum, float eps, int kind, int G, int N, int C,
                                        int64_t S, void* ws, size_t ws_bytes, void* stream) {
    DRAM_REQUIRE(parts && save_mean && save_rstd && rowcoef && ws && nparts > 0, "norm_finalize_parts: null pointer");
    int rc = check_norm("norm_finalize_parts", kind, G, N, C, S);
    if (rc) return rc;
    DRAM_REQUIRE((int64_t)N * C <= 65535, "norm_finalize_parts: N*C > 65535 rows not supported");
    if (ws_bytes < dram_norm_parts_ws_bytes(N, C, nparts)) {
        set_error("norm_finalize_parts: workspace too small");
        return DRAM_EWS;
    }
    hipStream_t st = (hipStream_t)stream;
    const int groups = parts_groups(nparts);
    double* part2 = (double*)ws;
    hipLaunchKernelGGL(parts_reduce_kernel, dim3(groups, N * C), dim3(256), 0, st, parts, part2, nparts, groups);
    const int Gk = kind == DRAM_NORM_BATCH ? 1 : G;
    const int nstat = kind == DRAM_NORM_BATCH ? C : N * G;
    const double expect = (double)(kind == DRAM_NORM_BATCH ? N : C / G) * (double)S;
    hipLaunchKernelGGL(norm_finalize_parts_kernel, dim3(nstat), dim3(256), 0, st, part2, groups, kind, N, C, Gk, gamma, beta,
                       eps, save_mean, save_rstd, rowcoef, running_mean, running_var, momentum, expect);
    return check_launch("norm_finalize_parts");
}

// This rank's per-channel {mean, M2} (fp64, interleaved like dram_bn_stats) from the conv epilogue's partials: the input of
// the cross-rank Chan combine of "sbn" on the fused engine.  (Inverting save_rstd -- var = rstd^-2 - eps -- cancels for
// channels whose variance is far below eps.)
extern "C" int dram_bn_parts_stats(const float* parts, int nparts, double* mean_m2, int N, int C, int64_t S, void* ws,
                                   size_t ws_bytes, void* stream) {
    DRAM_REQUIRE(parts && mean_m2 && ws && nparts > 0, "bn_parts_stats: null pointer");
    int rc = check_norm("bn_parts_stats", DRAM_NORM_BATCH, 1, N, C, S);
    if (rc) return rc;
    DRAM_REQUIRE((int64_t)N * C <= 65535, "bn_parts_stats: N*C > 65535 rows not supported");
    if (ws_bytes < dram_norm_parts_ws_bytes(N, C, nparts)) {
        set_error("bn_parts_stats: workspace too small");
        return DRAM_EWS;
    }
    hipStream_t st = (hipStream_t)stream;
    const int groups = parts_groups(nparts);
    double* part2 = (double*)ws;
    hipLaunchKernelGGL(parts_reduce_kernel, dim3(groups, N * C), dim3(256), 0, st, parts, part2, nparts, groups);
    hipLaunchKernelGGL(bn_parts_moments_kernel, dim3(C), dim3(256), 0, st, part2, groups, N, C, (double)N * (double)S, mean_m2);
    return check_launch("bn_parts_stats");
}

// y = act(rowcoef[row][0] * x + rowcoef[row][1]): materialises a lazily normalised tensor (the consumers without an
// on-load path, and the fallback of every fused kernel)
extern "C" int dram_row_affine_act(const float* x, const float* rowcoef, float* y, int relu, int64_t rows, int64_t S,
                                   void* stream) {
    DRAM_REQUIRE(x && rowcoef && y && rows > 0 && S > 0, "row_affine_act: bad arguments");
    DRAM_REQUIRE(rows <= 65535, "row_affine_act: more than 65535 rows not supported");
    hipStream_t st = (hipStream_t)stream;
    const int nch = nchunks_of(S);
    if (vec_ok(x, S) && vec_ok(y, S))
        hipLaunchKernelGGL(row_affine_act_kernel<true>, row_grid(nch, rows), dim3(256), 0, st, x, y, rowcoef, S, relu);
    else
        hipLaunchKernelGGL(row_affine_act_kernel<false>, row_grid(nch, rows), dim3(256), 0, st, x, y, rowcoef, S, relu);
    return check_launch("row_affine_act");
}

// eval-mode BatchNorm coefficients only (running statistics -> per-row {a, b}; no pass over the tensor)
extern "C" int dram_bn_eval_coef(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                                 float* save_mean, float* save_rstd, float* rowcoef, float eps, int N, int C, void* stream) {
    DRAM_REQUIRE(running_mean && running_var && save_mean && save_rstd && rowcoef && N > 0 && C > 0, "bn_eval_coef: bad arguments");
    const int64_t rows = (int64_t)N * C;
    hipLaunchKernelGGL(bn_eval_coef_kernel, dim3((unsigned)cdiv64(rows, 256)), dim3(256), 0, (hipStream_t)stream, gamma, beta,
                       running_mean, running_var, eps, N, C, save_mean, save_rstd, rowcoef);
    return check_launch("bn_eval_coef");
}

extern "C" int dram_relu_fwd(const float* x, float* y, int64_t n, void* stream) {
    DRAM_REQUIRE(x && y && n >= 0, "relu_fwd: bad arguments");
    if (n == 0) return DRAM_OK;
    const unsigned grid = (unsigned)(cdiv64(n, 256) < 8192 ? cdiv64(n, 256) : 8192);
    hipLaunchKernelGGL(relu_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, y, n);
    return check_launch("relu_fwd");
}

extern "C" int dram_relu_bwd(const float* dy, const float* y, float* dx, int64_t n, void* stream) {
    DRAM_REQUIRE(dy && y && dx && n >= 0, "relu_bwd: bad arguments");
    if (n == 0) return DRAM_OK;
    const unsigned grid = (unsigned)(cdiv64(n, 256) < 8192 ? cdiv64(n, 256) : 8192);
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dy, y, dx, n);
    return check_launch("relu_bwd");
}
