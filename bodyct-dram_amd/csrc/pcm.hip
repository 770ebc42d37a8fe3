// PCM local attention on a regular voxel grid, gfx950.
//
// Replaces the reference's graph formulation, dram/models.py: PCM.init_graph (221-258: a networkx
// DiGraph with one edge per (neighbour -> node) pair, ~4.7 M edges for 64^3 x 18, turned into a
// dgl.DGLGraph), PCM.forward's g.update_all (333-363), message_func / reduce_func (399-411) and
// compute_cross_x + merge_func (259-331, 365-397).  On a grid the graph is a stencil: node i receives
// from i + o_e for the E offsets of the structuring element (in-grid ones only), so
//
//   a[b,e,i]   = softmax_e( s(deg_i) * n_i( act( sum_f theta[b,f,i] * phi[b,f,i+o_e] ) ) )      (attention)
//   out[b,c,i] = sum_e a[b,e,i] * v[b,c,i+o_e]                                                   (aggregate)
//
// with act = identity | ReLU, n_i = identity | L2 normalisation over the node's edges, s = 1 |
// 1/sqrt(#edges of the node) | 1/0.01 -- the dot-product family of merge_func.  The geo variants of merge_func
// (scaled_dot_product_geo(_relu), att_is_all: an appearance term plus a positional-encoding term) are the same
// kernels on concatenated / summed feature planes, with the ReLU confined to the first F_relu feature planes:
// logit = act(sum_{f < F_relu} theta_f phi_f) + sum_{f >= F_relu} theta_f phi_f.  theta, phi, v and the
// result are NCDHW planes; the attention weights are kept as E planes [B,E,S] (saved for the backward
// and re-used by every non_local_iter).  HBM-bound stencil work, one thread per node along x;
// every reduction is a gather (no atomics), so results are deterministic.
#include "common.h"

namespace dram {

constexpr int PCM_MAXE = 128;
constexpr float PCM_L2EPS = 1e-12f;     // F.normalize's eps

struct PcmOffsets {
    int n;
    signed char dz[PCM_MAXE], dy[PCM_MAXE], dx[PCM_MAXE];
};

struct PcmGrid {
    int D, H, W;
    int64_t S;
};

enum { PCM_RELU = 1, PCM_L2NORM = 2 };
enum { PCM_SCALE_NONE = 0, PCM_SCALE_RSQRT_DEG = 1, PCM_SCALE_100 = 2 };

__device__ __forceinline__ bool pcm_neighbour(const PcmGrid& g, const PcmOffsets& o, int e, int z, int y, int x, int64_t i,
                                              int64_t& j) {
    const int zz = z + o.dz[e], yy = y + o.dy[e], xx = x + o.dx[e];
    j = i + ((int64_t)o.dz[e] * g.H + o.dy[e]) * g.W + o.dx[e];
    return (unsigned)zz < (unsigned)g.D && (unsigned)yy < (unsigned)g.H && (unsigned)xx < (unsigned)g.W;
}

__device__ __forceinline__ float pcm_scale(int scale_mode, float deg) {
    return scale_mode == PCM_SCALE_RSQRT_DEG ? 1.f / sqrtf(deg) : (scale_mode == PCM_SCALE_100 ? 100.f : 1.f);
}

// ---------------------------------------------------------------- attention weights
__global__ __launch_bounds__(256) void pcm_attn_fwd_kernel(const float* __restrict__ theta, const float* __restrict__ phi,
                                                           float* __restrict__ attn, PcmOffsets o, PcmGrid g, int F,
                                                           int F_relu, int flags, int scale_mode) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= g.S) return;
    const int b = blockIdx.y;
    const int x = (int)(i % g.W), y = (int)((i / g.W) % g.H), z = (int)(i / ((int64_t)g.W * g.H));
    const float* th = theta + (int64_t)b * F * g.S + i;
    const float* ph = phi + (int64_t)b * F * g.S;
    float* a = attn + (int64_t)b * o.n * g.S + i;
    // pass A: activated dot products u_e (parked in the output planes), degree, max, sum of squares
    float deg = 0.f, umax = -INFINITY, ss = 0.f;
    for (int e = 0; e < o.n; ++e) {
        int64_t j;
        float u = 0.f;
        if (pcm_neighbour(g, o, e, z, y, x, i, j)) {
            for (int f = 0; f < F_relu; ++f) u = fmaf(th[(int64_t)f * g.S], ph[(int64_t)f * g.S + j], u);
            if (flags & PCM_RELU) u = fmaxf(u, 0.f);
            float u2 = 0.f;                                     // (the feature planes outside the activation)
            for (int f = F_relu; f < F; ++f) u2 = fmaf(th[(int64_t)f * g.S], ph[(int64_t)f * g.S + j], u2);
            u += u2;
            deg += 1.f;
            umax = fmaxf(umax, u);
            ss = fmaf(u, u, ss);
            a[(int64_t)e * g.S] = u;
        } else {
            a[(int64_t)e * g.S] = -INFINITY;        // not an edge
        }
    }
    if (deg == 0.f) {                                // isolated node (1x1x1 grid without self loop): no messages
        for (int e = 0; e < o.n; ++e) a[(int64_t)e * g.S] = 0.f;
        return;
    }
    float c = pcm_scale(scale_mode, deg);
    if (flags & PCM_L2NORM) c /= fmaxf(sqrtf(ss), PCM_L2EPS);
    // pass B: exponentials (c > 0, so the largest logit is c * umax)
    float sum = 0.f;
    for (int e = 0; e < o.n; ++e) {
        const float u = a[(int64_t)e * g.S];
        const float w = u == -INFINITY ? 0.f : expf(c * (u - umax));
        sum += w;
        a[(int64_t)e * g.S] = w;
    }
    const float inv = 1.f / sum;
    for (int e = 0; e < o.n; ++e) a[(int64_t)e * g.S] *= inv;
}

// dlogit (gradient w.r.t. the raw dot products) from dattn; dtheta from it.  ds has the layout of attn.
__global__ __launch_bounds__(256) void pcm_attn_bwd_kernel(const float* __restrict__ theta, const float* __restrict__ phi,
                                                           const float* __restrict__ attn, const float* __restrict__ dattn,
                                                           float* __restrict__ ds, float* __restrict__ ds2,
                                                           float* __restrict__ dtheta, PcmOffsets o, PcmGrid g, int F, int F_relu,
                                                           int flags, int scale_mode) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= g.S) return;
    const int b = blockIdx.y;
    const int x = (int)(i % g.W), y = (int)((i / g.W) % g.H), z = (int)(i / ((int64_t)g.W * g.H));
    const float* th = theta + (int64_t)b * F * g.S + i;
    const float* ph = phi + (int64_t)b * F * g.S;
    const float* a = attn + (int64_t)b * o.n * g.S + i;
    const float* da = dattn + (int64_t)b * o.n * g.S + i;
    float* s = ds + (int64_t)b * o.n * g.S + i;
    // ds2 (split activation only, F_relu < F; host: no L2 normalisation then): the gradient w.r.t. the un-activated part
    float* s2 = ds2 ? ds2 + (int64_t)b * o.n * g.S + i : nullptr;
    // pass A: recompute the raw dots of the activated part (parked in ds), sum_k a_k da_k, degree, sum of squares of the
    // activated dots
    float deg = 0.f, ada = 0.f, ss = 0.f;
    for (int e = 0; e < o.n; ++e) {
        int64_t j;
        float dot = 0.f;
        if (pcm_neighbour(g, o, e, z, y, x, i, j)) {
            for (int f = 0; f < F_relu; ++f) dot = fmaf(th[(int64_t)f * g.S], ph[(int64_t)f * g.S + j], dot);
            deg += 1.f;
            const float u = (flags & PCM_RELU) ? fmaxf(dot, 0.f) : dot;
            ss = fmaf(u, u, ss);
            ada = fmaf(a[(int64_t)e * g.S], da[(int64_t)e * g.S], ada);
        }
        s[(int64_t)e * g.S] = dot;
    }
    const float scale = deg > 0.f ? pcm_scale(scale_mode, deg) : 0.f;
    const float nrm = fmaxf(sqrtf(ss), PCM_L2EPS);
    const bool l2 = flags & PCM_L2NORM, l2live = sqrtf(ss) > PCM_L2EPS;
    // pass B (L2 only): sum_k v_k dv_k with v = u / nrm, dv = scale * dlogit
    float vdv = 0.f;
    if (l2 && l2live) {
        for (int e = 0; e < o.n; ++e) {
            int64_t j;
            if (!pcm_neighbour(g, o, e, z, y, x, i, j)) continue;
            const float dot = s[(int64_t)e * g.S];
            const float u = (flags & PCM_RELU) ? fmaxf(dot, 0.f) : dot;
            const float ae = a[(int64_t)e * g.S];
            vdv = fmaf(u / nrm, scale * ae * (da[(int64_t)e * g.S] - ada), vdv);
        }
    }
    // pass C: gradient w.r.t. the raw dot of every edge
    for (int e = 0; e < o.n; ++e) {
        int64_t j;
        float d = 0.f;
        if (pcm_neighbour(g, o, e, z, y, x, i, j)) {
            const float dot = s[(int64_t)e * g.S];
            const float u = (flags & PCM_RELU) ? fmaxf(dot, 0.f) : dot;
            const float ae = a[(int64_t)e * g.S];
            const float dl = scale * ae * (da[(int64_t)e * g.S] - ada);      // w.r.t. the normalised, unscaled logit
            d = l2 ? (l2live ? (dl - (u / nrm) * vdv) / nrm : dl / nrm) : dl;
            if (s2) s2[(int64_t)e * g.S] = d;                   // the un-activated part sees the logit's gradient as it is
            if ((flags & PCM_RELU) && !(dot > 0.f)) d = 0.f;
        } else if (s2) {
            s2[(int64_t)e * g.S] = 0.f;
        }
        s[(int64_t)e * g.S] = d;
    }
    // dtheta[b,f,i] = sum_e ds_e * phi[b,f,i+o_e]   (ds2 for the feature planes outside the activation)
    float* dth = dtheta + (int64_t)b * F * g.S + i;
    for (int f = 0; f < F; ++f) {
        const float* sf = (f < F_relu || !s2) ? s : s2;
        float acc = 0.f;
        for (int e = 0; e < o.n; ++e) {
            int64_t j;
            if (pcm_neighbour(g, o, e, z, y, x, i, j)) acc = fmaf(sf[(int64_t)e * g.S], ph[(int64_t)f * g.S + j], acc);
        }
        dth[(int64_t)f * g.S] = acc;
    }
}

// ---------------------------------------------------------------- sum-normalised merges (cosine, heu1, heu2)
// merge_func's variants without a softmax (models.py:300-302, 307-320): a per-edge similarity v_e >= ... of (theta_i, phi_j),
//     cosine: v = sum_f (theta_f / max(|theta|, 1e-8)) (phi_f / max(|phi|, 1e-8))          (F.cosine_similarity)
//     heu1:   u = theta.phi / (1 + sum_f |theta_f - phi_f|),  v = u if u >= 0.03 else 0     (the mask carries no gradient)
//     heu2:   v = relu(u)
// normalised by the sum over the node's edges, a_e = v_e / (eps + sum_k v_k), eps = 0 (cosine) / 1e-7 (heu).
enum { PCM_SUM_COSINE = 0, PCM_SUM_HEU1 = 1, PCM_SUM_HEU2 = 2 };
constexpr int PCM_SUM_MAXF = 64;
constexpr float PCM_COS_EPS = 1e-8f;

// similarity of one (node, neighbour) pair and what its derivative needs: aux0 = |theta| clamp (cosine) / 1 + L1 (heu),
// aux1 = |phi_j| clamp (cosine) / u (heu); gate = dv/du of heu (1 / 0)
__device__ __forceinline__ float pcm_sum_value(int mode, const float* th, const float* phj, int64_t S, int F, float& aux0,
                                               float& aux1, float& gate) {
    float dot = 0.f;
    if (mode == PCM_SUM_COSINE) {
        float tt = 0.f, pp = 0.f;
        for (int f = 0; f < F; ++f) {
            const float t = th[(int64_t)f * S], q = phj[(int64_t)f * S];
            dot = fmaf(t, q, dot); tt = fmaf(t, t, tt); pp = fmaf(q, q, pp);
        }
        aux0 = fmaxf(sqrtf(tt), PCM_COS_EPS);
        aux1 = fmaxf(sqrtf(pp), PCM_COS_EPS);
        gate = 1.f;
        return dot / (aux0 * aux1);
    }
    float l1 = 0.f;
    for (int f = 0; f < F; ++f) {
        const float t = th[(int64_t)f * S], q = phj[(int64_t)f * S];
        dot = fmaf(t, q, dot); l1 += fabsf(t - q);
    }
    aux0 = 1.f + l1;
    const float u = dot / aux0;
    aux1 = u;
    gate = mode == PCM_SUM_HEU1 ? (u < 0.03f ? 0.f : 1.f) : (u > 0.f ? 1.f : 0.f);
    return gate * u;
}

__global__ __launch_bounds__(256) void pcm_attn_sum_fwd_kernel(const float* __restrict__ theta, const float* __restrict__ phi,
                                                               float* __restrict__ attn, PcmOffsets o, PcmGrid g, int F, int mode) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= g.S) return;
    const int b = blockIdx.y;
    const int x = (int)(i % g.W), y = (int)((i / g.W) % g.H), z = (int)(i / ((int64_t)g.W * g.H));
    const float* th = theta + (int64_t)b * F * g.S + i;
    const float* ph = phi + (int64_t)b * F * g.S;
    float* a = attn + (int64_t)b * o.n * g.S + i;
    float sum = 0.f;
    for (int e = 0; e < o.n; ++e) {
        int64_t j;
        float v = 0.f, a0, a1, gt;
        if (pcm_neighbour(g, o, e, z, y, x, i, j)) v = pcm_sum_value(mode, th, ph + j, g.S, F, a0, a1, gt);
        sum += v;
        a[(int64_t)e * g.S] = v;
    }
    const float inv = 1.f / (sum + (mode == PCM_SUM_COSINE ? 0.f : 1e-7f));
    for (int e = 0; e < o.n; ++e) a[(int64_t)e * g.S] *= inv;
}

// ds[b,e,i] = d loss / d v_e (through the normalisation); dtheta from it
__global__ __launch_bounds__(256) void pcm_attn_sum_bwd_kernel(const float* __restrict__ theta, const float* __restrict__ phi,
                                                               const float* __restrict__ attn, const float* __restrict__ dattn,
                                                               float* __restrict__ ds, float* __restrict__ dtheta, PcmOffsets o,
                                                               PcmGrid g, int F, int mode) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= g.S) return;
    const int b = blockIdx.y;
    const int x = (int)(i % g.W), y = (int)((i / g.W) % g.H), z = (int)(i / ((int64_t)g.W * g.H));
    const float* th = theta + (int64_t)b * F * g.S + i;
    const float* ph = phi + (int64_t)b * F * g.S;
    const float* a = attn + (int64_t)b * o.n * g.S + i;
    const float* da = dattn + (int64_t)b * o.n * g.S + i;
    float* s = ds + (int64_t)b * o.n * g.S + i;
    float acc[PCM_SUM_MAXF];
    for (int f = 0; f < F; ++f) acc[f] = 0.f;
    float sum = 0.f, ada = 0.f;
    for (int e = 0; e < o.n; ++e) {
        int64_t j;
        float a0, a1, gt;
        if (pcm_neighbour(g, o, e, z, y, x, i, j)) {
            sum += pcm_sum_value(mode, th, ph + j, g.S, F, a0, a1, gt);
            ada = fmaf(a[(int64_t)e * g.S], da[(int64_t)e * g.S], ada);
        }
    }
    const float invT = 1.f / (sum + (mode == PCM_SUM_COSINE ? 0.f : 1e-7f));
    for (int e = 0; e < o.n; ++e) {
        int64_t j;
        float dv = 0.f;
        if (pcm_neighbour(g, o, e, z, y, x, i, j)) {
            float a0, a1, gt;
            const float v = pcm_sum_value(mode, th, ph + j, g.S, F, a0, a1, gt);
            dv = (da[(int64_t)e * g.S] - ada) * invT;
            const float* phj = ph + j;
            if (mode == PCM_SUM_COSINE) {       // dv/dtheta_f = (phi^_f - v theta^_f) / c_theta  (theta^ = theta / c_theta; no v term under the clamp)
                float tt = 0.f;
                for (int f = 0; f < F; ++f) tt = fmaf(th[(int64_t)f * g.S], th[(int64_t)f * g.S], tt);
                const float live = sqrtf(tt) > PCM_COS_EPS ? 1.f : 0.f;
                for (int f = 0; f < F; ++f)
                    acc[f] = fmaf(dv, (phj[(int64_t)f * g.S] / a1 - live * v * th[(int64_t)f * g.S] / a0) / a0, acc[f]);
            } else {                            // du/dtheta_f = (phi_f - u sgn(theta_f - phi_f)) / (1 + L1)
                const float du = dv * gt / a0;
                for (int f = 0; f < F; ++f) {
                    const float t = th[(int64_t)f * g.S], q = phj[(int64_t)f * g.S];
                    const float sg = t > q ? 1.f : (t < q ? -1.f : 0.f);
                    acc[f] = fmaf(du, q - a1 * sg, acc[f]);
                }
            }
        }
        s[(int64_t)e * g.S] = dv;
    }
    float* dth = dtheta + (int64_t)b * F * g.S + i;
    for (int f = 0; f < F; ++f) dth[(int64_t)f * g.S] = acc[f];
}

// dphi[b,f,j] = sum over the edges (i -> j = i + o_e) of ds[b,e,i] * d v_e(i) / d phi_f(j): gather at j, pair values recomputed
__global__ __launch_bounds__(256) void pcm_attn_sum_dphi_kernel(const float* __restrict__ theta, const float* __restrict__ phi,
                                                                const float* __restrict__ ds, float* __restrict__ dphi, PcmOffsets o,
                                                                PcmGrid g, int F, int mode) {
    const int64_t jn = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (jn >= g.S) return;
    const int b = blockIdx.y;
    const int x = (int)(jn % g.W), y = (int)((jn / g.W) % g.H), z = (int)(jn / ((int64_t)g.W * g.H));
    const float* thb = theta + (int64_t)b * F * g.S;
    const float* phj = phi + (int64_t)b * F * g.S + jn;
    const float* sb = ds + (int64_t)b * o.n * g.S;
    float acc[PCM_SUM_MAXF];
    for (int f = 0; f < F; ++f) acc[f] = 0.f;
    for (int e = 0; e < o.n; ++e) {
        const int zz = z - o.dz[e], yy = y - o.dy[e], xx = x - o.dx[e];
        if (!((unsigned)zz < (unsigned)g.D && (unsigned)yy < (unsigned)g.H && (unsigned)xx < (unsigned)g.W)) continue;
        const int64_t i = jn - (((int64_t)o.dz[e] * g.H + o.dy[e]) * g.W + o.dx[e]);
        const float dv = sb[(int64_t)e * g.S + i];
        const float* th = thb + i;
        float a0, a1, gt;
        const float v = pcm_sum_value(mode, th, phj, g.S, F, a0, a1, gt);
        if (mode == PCM_SUM_COSINE) {           // dv/dphi_f = (theta^_f - v phi^_f) / c_phi
            float pp = 0.f;
            for (int f = 0; f < F; ++f) pp = fmaf(phj[(int64_t)f * g.S], phj[(int64_t)f * g.S], pp);
            const float live = sqrtf(pp) > PCM_COS_EPS ? 1.f : 0.f;
            for (int f = 0; f < F; ++f)
                acc[f] = fmaf(dv, (th[(int64_t)f * g.S] / a0 - live * v * phj[(int64_t)f * g.S] / a1) / a1, acc[f]);
        } else {                                // du/dphi_f = (theta_f + u sgn(theta_f - phi_f)) / (1 + L1)
            const float du = dv * gt / a0;
            for (int f = 0; f < F; ++f) {
                const float t = th[(int64_t)f * g.S], q = phj[(int64_t)f * g.S];
                const float sg = t > q ? 1.f : (t < q ? -1.f : 0.f);
                acc[f] = fmaf(du, t + a1 * sg, acc[f]);
            }
        }
    }
    float* o_ = dphi + (int64_t)b * F * g.S + jn;
    for (int f = 0; f < F; ++f) o_[(int64_t)f * g.S] = acc[f];
}

// Adjoint gather shared by dphi and dv:  out[b,c,j] = sum_e w[b,e,j-o_e] * src[b,c,j-o_e]  (j - o_e inside the grid)
__global__ __launch_bounds__(256) void pcm_scatter_adjoint_kernel(const float* __restrict__ w, const float* __restrict__ src,
                                                                  float* __restrict__ out, PcmOffsets o, PcmGrid g, int C,
                                                                  int c_lo) {
    const int64_t jn = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (jn >= g.S) return;
    const int c = c_lo + blockIdx.y, b = blockIdx.z;      // channels [c_lo, c_lo + gridDim.y) of C
    const int x = (int)(jn % g.W), y = (int)((jn / g.W) % g.H), z = (int)(jn / ((int64_t)g.W * g.H));
    const float* wb = w + (int64_t)b * o.n * g.S;
    const float* sb = src + ((int64_t)b * C + c) * g.S;
    float acc = 0.f;
    for (int e = 0; e < o.n; ++e) {
        const int zz = z - o.dz[e], yy = y - o.dy[e], xx = x - o.dx[e];
        if ((unsigned)zz < (unsigned)g.D && (unsigned)yy < (unsigned)g.H && (unsigned)xx < (unsigned)g.W) {
            const int64_t i = jn - (((int64_t)o.dz[e] * g.H + o.dy[e]) * g.W + o.dx[e]);
            acc = fmaf(wb[(int64_t)e * g.S + i], sb[i], acc);
        }
    }
    out[((int64_t)b * C + c) * g.S + jn] = acc;
}

// ---------------------------------------------------------------- aggregation
__global__ __launch_bounds__(256) void pcm_aggregate_fwd_kernel(const float* __restrict__ attn, const float* __restrict__ v,
                                                                float* __restrict__ out, PcmOffsets o, PcmGrid g, int C) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= g.S) return;
    const int c = blockIdx.y, b = blockIdx.z;
    const int x = (int)(i % g.W), y = (int)((i / g.W) % g.H), z = (int)(i / ((int64_t)g.W * g.H));
    const float* a = attn + (int64_t)b * o.n * g.S + i;
    const float* vb = v + ((int64_t)b * C + c) * g.S;
    float acc = 0.f;
    for (int e = 0; e < o.n; ++e) {
        int64_t j;
        if (pcm_neighbour(g, o, e, z, y, x, i, j)) acc = fmaf(a[(int64_t)e * g.S], vb[j], acc);
    }
    out[((int64_t)b * C + c) * g.S + i] = acc;
}

// dattn[b,e,i] = sum_c dout[b,c,i] * v[b,c,i+o_e]
__global__ __launch_bounds__(256) void pcm_aggregate_dattn_kernel(const float* __restrict__ dout, const float* __restrict__ v,
                                                                  float* __restrict__ dattn, PcmOffsets o, PcmGrid g, int C) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= g.S) return;
    const int b = blockIdx.y;
    const int x = (int)(i % g.W), y = (int)((i / g.W) % g.H), z = (int)(i / ((int64_t)g.W * g.H));
    const float* d = dout + (int64_t)b * C * g.S + i;
    const float* vb = v + (int64_t)b * C * g.S;
    float* da = dattn + (int64_t)b * o.n * g.S + i;
    for (int e = 0; e < o.n; ++e) {
        int64_t j;
        float acc = 0.f;
        if (pcm_neighbour(g, o, e, z, y, x, i, j))
            for (int c = 0; c < C; ++c) acc = fmaf(d[(int64_t)c * g.S], vb[(int64_t)c * g.S + j], acc);
        da[(int64_t)e * g.S] = acc;
    }
}

static int pcm_setup(const int* offsets, int E, int D, int H, int W, PcmOffsets& o, PcmGrid& g, const char* who) {
    if (!offsets || E <= 0 || E > PCM_MAXE) {
        set_error("%s: need 1..128 neighbour offsets", who);
        return DRAM_EINVAL;
    }
    if (D <= 0 || H <= 0 || W <= 0 || (int64_t)D * H * W >= ((int64_t)1 << 31)) {
        set_error("%s: bad grid", who);
        return DRAM_EINVAL;
    }
    o.n = E;
    for (int e = 0; e < E; ++e) {
        for (int k = 0; k < 3; ++k)
            if (offsets[3 * e + k] < -127 || offsets[3 * e + k] > 127) {
                set_error("%s: offset out of range", who);
                return DRAM_EINVAL;
            }
        o.dz[e] = (signed char)offsets[3 * e];
        o.dy[e] = (signed char)offsets[3 * e + 1];
        o.dx[e] = (signed char)offsets[3 * e + 2];
    }
    g.D = D; g.H = H; g.W = W; g.S = (int64_t)D * H * W;
    return DRAM_OK;
}

}  // namespace dram

using namespace dram;

extern "C" int dram_pcm_attention_split_fwd(const float* theta, const float* phi, const int* offsets, int E, int flags,
                                            int scale_mode, int F_relu, float* attn, int B, int F, int D, int H, int W,
                                            void* stream) {
    DRAM_REQUIRE(theta && phi && attn, "pcm_attention_fwd: null pointer");
    DRAM_REQUIRE(B > 0 && B <= 65535 && F > 0, "pcm_attention_fwd: bad dimensions");
    DRAM_REQUIRE((flags & ~3) == 0 && scale_mode >= 0 && scale_mode <= 2, "pcm_attention_fwd: unknown mode");
    DRAM_REQUIRE(F_relu >= 0 && F_relu <= F && (F_relu == F || !(flags & PCM_L2NORM)),
                 "pcm_attention_fwd: the activated part is F_relu = %d of %d feature planes (no L2 normalisation with a split)", F_relu, F);
    PcmOffsets o;
    PcmGrid g;
    if (int rc = pcm_setup(offsets, E, D, H, W, o, g, "pcm_attention_fwd")) return rc;
    hipLaunchKernelGGL(pcm_attn_fwd_kernel, dim3((unsigned)cdiv64(g.S, 256), B), dim3(256), 0, (hipStream_t)stream, theta, phi,
                       attn, o, g, F, F_relu, flags, scale_mode);
    return check_launch("pcm_attention_fwd");
}

extern "C" int dram_pcm_attention_fwd(const float* theta, const float* phi, const int* offsets, int E, int flags,
                                      int scale_mode, float* attn, int B, int F, int D, int H, int W, void* stream) {
    return dram_pcm_attention_split_fwd(theta, phi, offsets, E, flags, scale_mode, F, attn, B, F, D, H, W, stream);
}

extern "C" int dram_pcm_attention_split_bwd(const float* theta, const float* phi, const float* attn, const float* dattn,
                                            const int* offsets, int E, int flags, int scale_mode, int F_relu, float* ds,
                                            float* ds2, float* dtheta, float* dphi, int B, int F, int D, int H, int W,
                                            void* stream) {
    DRAM_REQUIRE(theta && phi && attn && dattn && ds && dtheta && dphi, "pcm_attention_bwd: null pointer");
    DRAM_REQUIRE(B > 0 && B <= 65535 && F > 0 && F <= 65535, "pcm_attention_bwd: bad dimensions");
    DRAM_REQUIRE((flags & ~3) == 0 && scale_mode >= 0 && scale_mode <= 2, "pcm_attention_bwd: unknown mode");
    DRAM_REQUIRE(F_relu >= 0 && F_relu <= F && (F_relu == F || !(flags & PCM_L2NORM)),
                 "pcm_attention_bwd: the activated part is F_relu = %d of %d feature planes (no L2 normalisation with a split)", F_relu, F);
    const bool split = F_relu < F && (flags & PCM_RELU);      // (without an activation the two parts share one gradient)
    DRAM_REQUIRE(!split || ds2, "pcm_attention_bwd: a split activation needs the second gradient buffer ds2");
    PcmOffsets o;
    PcmGrid g;
    if (int rc = pcm_setup(offsets, E, D, H, W, o, g, "pcm_attention_bwd")) return rc;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(pcm_attn_bwd_kernel, dim3((unsigned)cdiv64(g.S, 256), B), dim3(256), 0, st, theta, phi, attn, dattn, ds,
                       split ? ds2 : (float*)nullptr, dtheta, o, g, F, split ? F_relu : F, flags, scale_mode);
    // dphi[b,f,j] = sum_e ds[b,e,j-o_e] * theta[b,f,j-o_e]
    const int F1 = split ? F_relu : F;
    if (F1 > 0)
        hipLaunchKernelGGL(pcm_scatter_adjoint_kernel, dim3((unsigned)cdiv64(g.S, 256), F1, B), dim3(256), 0, st, ds, theta, dphi, o, g, F, 0);
    if (F1 < F)
        hipLaunchKernelGGL(pcm_scatter_adjoint_kernel, dim3((unsigned)cdiv64(g.S, 256), F - F1, B), dim3(256), 0, st, ds2, theta, dphi, o,
                           g, F, F1);
    return check_launch("pcm_attention_bwd");
}

extern "C" int dram_pcm_attention_bwd(const float* theta, const float* phi, const float* attn, const float* dattn,
                                      const int* offsets, int E, int flags, int scale_mode, float* ds, float* dtheta,
                                      float* dphi, int B, int F, int D, int H, int W, void* stream) {
    return dram_pcm_attention_split_bwd(theta, phi, attn, dattn, offsets, E, flags, scale_mode, F, ds, nullptr, dtheta, dphi, B, F,
                                        D, H, W, stream);
}

extern "C" int dram_pcm_attention_sum_fwd(const float* theta, const float* phi, const int* offsets, int E, int mode,
                                          float* attn, int B, int F, int D, int H, int W, void* stream) {
    DRAM_REQUIRE(theta && phi && attn, "pcm_attention_sum_fwd: null pointer");
    DRAM_REQUIRE(B > 0 && B <= 65535 && F > 0 && F <= PCM_SUM_MAXF, "pcm_attention_sum_fwd: bad dimensions (1..%d feature planes)", PCM_SUM_MAXF);
    DRAM_REQUIRE(mode >= 0 && mode <= 2, "pcm_attention_sum_fwd: unknown mode");
    PcmOffsets o;
    PcmGrid g;
    if (int rc = pcm_setup(offsets, E, D, H, W, o, g, "pcm_attention_sum_fwd")) return rc;
    hipLaunchKernelGGL(pcm_attn_sum_fwd_kernel, dim3((unsigned)cdiv64(g.S, 256), B), dim3(256), 0, (hipStream_t)stream, theta, phi,
                       attn, o, g, F, mode);
    return check_launch("pcm_attention_sum_fwd");
}

extern "C" int dram_pcm_attention_sum_bwd(const float* theta, const float* phi, const float* attn, const float* dattn,
                                          const int* offsets, int E, int mode, float* ds, float* dtheta, float* dphi, int B,
                                          int F, int D, int H, int W, void* stream) {
    DRAM_REQUIRE(theta && phi && attn && dattn && ds && dtheta && dphi, "pcm_attention_sum_bwd: null pointer");
    DRAM_REQUIRE(B > 0 && B <= 65535 && F > 0 && F <= PCM_SUM_MAXF, "pcm_attention_sum_bwd: bad dimensions (1..%d feature planes)", PCM_SUM_MAXF);
    DRAM_REQUIRE(mode >= 0 && mode <= 2, "pcm_attention_sum_bwd: unknown mode");
    PcmOffsets o;
    PcmGrid g;
    if (int rc = pcm_setup(offsets, E, D, H, W, o, g, "pcm_attention_sum_bwd")) return rc;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(pcm_attn_sum_bwd_kernel, dim3((unsigned)cdiv64(g.S, 256), B), dim3(256), 0, st, theta, phi, attn, dattn, ds,
                       dtheta, o, g, F, mode);
    hipLaunchKernelGGL(pcm_attn_sum_dphi_kernel, dim3((unsigned)cdiv64(g.S, 256), B), dim3(256), 0, st, theta, phi, ds, dphi, o, g, F, mode);
    return check_launch("pcm_attention_sum_bwd");
}

extern "C" int dram_pcm_aggregate_fwd(const float* attn, const float* v, const int* offsets, int E, float* out, int B, int C,
                                      int D, int H, int W, void* stream) {
    DRAM_REQUIRE(attn && v && out, "pcm_aggregate_fwd: null pointer");
    DRAM_REQUIRE(B > 0 && B <= 65535 && C > 0 && C <= 65535, "pcm_aggregate_fwd: bad dimensions");
    PcmOffsets o;
    PcmGrid g;
    if (int rc = pcm_setup(offsets, E, D, H, W, o, g, "pcm_aggregate_fwd")) return rc;
    hipLaunchKernelGGL(pcm_aggregate_fwd_kernel, dim3((unsigned)cdiv64(g.S, 256), C, B), dim3(256), 0, (hipStream_t)stream, attn, v,
                       out, o, g, C);
    return check_launch("pcm_aggregate_fwd");
}

extern "C" int dram_pcm_aggregate_bwd(const float* attn, const float* v, const float* dout, const int* offsets, int E,
                                      float* dattn, float* dv, int B, int C, int D, int H, int W, void* stream) {
    DRAM_REQUIRE(attn && v && dout && dattn && dv, "pcm_aggregate_bwd: null pointer");
    DRAM_REQUIRE(B > 0 && B <= 65535 && C > 0 && C <= 65535, "pcm_aggregate_bwd: bad dimensions");
    PcmOffsets o;
    PcmGrid g;
    if (int rc = pcm_setup(offsets, E, D, H, W, o, g, "pcm_aggregate_bwd")) return rc;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(pcm_aggregate_dattn_kernel, dim3((unsigned)cdiv64(g.S, 256), B), dim3(256), 0, st, dout, v, dattn, o, g, C);
    // dv[b,c,j] = sum_e attn[b,e,j-o_e] * dout[b,c,j-o_e]
    hipLaunchKernelGGL(pcm_scatter_adjoint_kernel, dim3((unsigned)cdiv64(g.S, 256), C, B), dim3(256), 0, st, attn, dout, dv, o, g, C, 0);
    return check_launch("pcm_aggregate_bwd");
}
