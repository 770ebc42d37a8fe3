// 2x2x2 max-pool, trilinear (align_corners=True) resize and crop-concat for fp32 NCDHW, gfx950.
//
// Replaces the ATen max_pool3d / upsample_trilinear3d / cat dispatches (+ backward) of
// reference dram/parts.py:191 (MaxPool3d(2,2,0)), parts.py:149 and models.py:146
// (nn.Upsample trilinear, align_corners=True) and parts.py:37-46 (crop_concat_5d).
//
// All HBM-bound streaming kernels: one thread per output element along x (coalesced
// 256-byte wave stores), reads are row segments that the wave covers contiguously.
#include "common.h"

namespace dram {

// ---------------------------------------------------------------- max pool
// thread per pooled output; idx = dz*4 + dy*2 + dx of the FIRST maximum in (z,y,x) scan
// order (ATen: `if (val > maxval || isnan(val))`), which is where the gradient goes.
// `coef` (optional): x holds the RAW conv output and the pooled tensor is act(a*x + b) per (n,c) row, ReLU if `relu`
// ("normalise + ReLU on load": the activated tensor is never written; same fmaf / fmaxf as the materialising kernel)
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                           uint8_t* __restrict__ idx, int D, int H, int W, int Do,
                                                           int Ho, int Wo, const float* __restrict__ coef, int relu) {
    const int64_t plane = blockIdx.y;  // n*C + c
    const int So = Do * Ho * Wo;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= So) return;
    const int xo = e % Wo, yo = (e / Wo) % Ho, zo = e / (Wo * Ho);
    const float* p = x + plane * ((int64_t)D * H * W) + ((int64_t)(2 * zo) * H + 2 * yo) * W + 2 * xo;
    const float ca = coef ? coef[2 * plane] : 1.f, cb = coef ? coef[2 * plane + 1] : 0.f;
    const float lo = (coef && relu) ? 0.f : -INFINITY;
    auto at = [&](int k) {
        const int dz = k >> 2, dy = (k >> 1) & 1, dx = k & 1;
        const float v = p[((int64_t)dz * H + dy) * W + dx];
        return coef ? fmaxf(fmaf(ca, v, cb), lo) : v;
    };
    float best = at(0);
    int bi = 0;
#pragma unroll
    for (int k = 1; k < 8; ++k) {
        const float v = at(k);
        if (v > best || v != v) { best = v; bi = k; }
    }
    out[plane * So + e] = best;
    idx[plane * So + e] = (uint8_t)bi;
}

// thread per INPUT element (coalesced dx stores); positions beyond the floor-cropped extent get 0.
// `accumulate`: dx += (the gradient is added to one that is already there: the skip connection's)
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const float* __restrict__ dout,
                                                           const uint8_t* __restrict__ idx, float* dx,
                                                           int D, int H, int W, int Do, int Ho, int Wo, int accumulate) {
    const int64_t plane = blockIdx.y;
    const int S = D * H * W;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= S) return;
    const int xi = e % W, yi = (e / W) % H, zi = e / (W * H);
    const int xo = xi >> 1, yo = yi >> 1, zo = zi >> 1;
    float g = 0.f;
    if (xo < Wo && yo < Ho && zo < Do) {
        const int64_t o = plane * ((int64_t)Do * Ho * Wo) + ((int64_t)zo * Ho + yo) * Wo + xo;
        const int local = ((zi & 1) << 2) | ((yi & 1) << 1) | (xi & 1);
        if (idx[o] == local) g = dout[o];
    }
    if (accumulate) g += dx[plane * (int64_t)S + e];
    dx[plane * (int64_t)S + e] = g;
}

// Same, four consecutive x per thread (W % 4 == 0, so Wo = W/2 and the two pooled cells of a quad are adjacent):
// one 16-byte store, one 2-byte index load and one 8-byte gradient load instead of 4 + 4 + 4 narrow ones.
__global__ __launch_bounds__(256) void maxpool2_bwd_vec_kernel(const float* __restrict__ dout,
                                                               const uint8_t* __restrict__ idx, float* dx,
                                                               int D, int H, int W, int Do, int Ho, int Wo, int accumulate) {
    const int64_t plane = blockIdx.y;
    const int W4 = W >> 2;
    const int S4 = D * H * W4;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= S4) return;
    const int x4 = e % W4, yi = (e / W4) % H, zi = e / (W4 * H);
    const int yo = yi >> 1, zo = zi >> 1;
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    if (yo < Ho && zo < Do) {
        const int64_t o = plane * ((int64_t)Do * Ho * Wo) + ((int64_t)zo * Ho + yo) * Wo + 2 * x4;   // even: 2-/8-byte aligned
        const unsigned short two = *reinterpret_cast<const unsigned short*>(idx + o);
        const float2 d = *reinterpret_cast<const float2*>(dout + o);
        const int base = ((zi & 1) << 2) | ((yi & 1) << 1);
        const int i0 = two & 0xff, i1 = two >> 8;
        g.x = i0 == base ? d.x : 0.f;
        g.y = i0 == (base | 1) ? d.x : 0.f;
        g.z = i1 == base ? d.y : 0.f;
        g.w = i1 == (base | 1) ? d.y : 0.f;
    }
    float4* q = reinterpret_cast<float4*>(dx + plane * ((int64_t)D * H * W) + 4 * (int64_t)e);
    if (accumulate) {
        const float4 old = *q;
        g.x += old.x; g.y += old.y; g.z += old.z; g.w += old.w;
    }
    *q = g;
}

// ---------------------------------------------------------------- trilinear, align_corners=True
// ATen (UpSample.h area_pixel_compute_scale / compute_source_index, align_corners branch):
//   scale = out > 1 ? (in-1)/(out-1) : 0 (fp32);  src = scale*dst;  i0 = (int)src;
//   i1 = i0 + (i0 < in-1);  l1 = src - i0;  l0 = 1 - l1.
struct Axis {
    int in, out;
    float scale;
    int half;     // 0: align_corners=True (src = scale*dst); 1: align_corners=False (src = max(scale*(dst+0.5)-0.5, 0))
};

__device__ __forceinline__ void src_index(const Axis& a, int o, int& i0, int& i1, float& l0, float& l1) {
    const float src = a.half ? fmaxf(a.scale * ((float)o + 0.5f) - 0.5f, 0.f) : a.scale * (float)o;
    i0 = (int)src;
    if (i0 > a.in - 1) i0 = a.in - 1;
    i1 = i0 + (i0 < a.in - 1 ? 1 : 0);
    l1 = src - (float)i0;
    l0 = 1.f - l1;
}

// one thread per output voxel; the source indices / weights are computed once and reused for CPT
// consecutive (n,c) planes (the index arithmetic, not the 8 loads, dominates a plane-per-thread version)
constexpr int TRI_CPT = 8;
__global__ __launch_bounds__(256) void trilinear_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                            Axis az, Axis ay, Axis ax, int planes,
                                                            const float* __restrict__ coef, int relu) {
    const int So = az.out * ay.out * ax.out;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= So) return;
    const int xo = e % ax.out, yo = (e / ax.out) % ay.out, zo = e / (ax.out * ay.out);
    int z0, z1, y0, y1, x0, x1;
    float a0, a1, b0, b1, c0, c1;
    src_index(az, zo, z0, z1, a0, a1);
    src_index(ay, yo, y0, y1, b0, b1);
    src_index(ax, xo, x0, x1, c0, c1);
    const int H = ay.in, W = ax.in;
    const int Si = az.in * H * W;
    const int o000 = (z0 * H + y0) * W + x0, o001 = (z0 * H + y0) * W + x1;
    const int o010 = (z0 * H + y1) * W + x0, o011 = (z0 * H + y1) * W + x1;
    const int o100 = (z1 * H + y0) * W + x0, o101 = (z1 * H + y0) * W + x1;
    const int o110 = (z1 * H + y1) * W + x0, o111 = (z1 * H + y1) * W + x1;
    const int p0 = blockIdx.y * TRI_CPT;
#pragma unroll
    for (int u = 0; u < TRI_CPT; ++u) {
        const int plane = p0 + u;
        if (plane >= planes) break;
        const float* p = x + (int64_t)plane * Si;
        float t[8] = {p[o000], p[o001], p[o010], p[o011], p[o100], p[o101], p[o110], p[o111]};
        if (coef) {   // the source is a RAW conv output: interpolate act(a*x + b) (normalise + ReLU on load)
            const float ca = coef[2 * plane], cb = coef[2 * plane + 1], lo = relu ? 0.f : -INFINITY;
#pragma unroll
            for (int k = 0; k < 8; ++k) t[k] = fmaxf(fmaf(ca, t[k], cb), lo);
        }
        const float v = a0 * (b0 * (c0 * t[0] + c1 * t[1]) + b1 * (c0 * t[2] + c1 * t[3])) +
                        a1 * (b0 * (c0 * t[4] + c1 * t[5]) + b1 * (c0 * t[6] + c1 * t[7]));
        y[(int64_t)plane * So + e] = v;
    }
}

// Four consecutive x outputs per thread, for magnifications of 1.5x and more along x (scale <= 0.6: the four outputs
// read at most four consecutive source columns): per (z,y) tap row ONE 16-byte load of the source columns
// [xs, xs+3] instead of eight 4-byte loads, one 16-byte store instead of four 4-byte ones.  The z and y blends are
// applied to the four source columns first and the x blend last (ATen blends x first: same terms, different rounding
// order -- both are fp32 evaluations of the same trilinear form).
__device__ __forceinline__ float sel4(const float (&t)[4], int i) {
    return i == 0 ? t[0] : (i == 1 ? t[1] : (i == 2 ? t[2] : t[3]));
}
__global__ __launch_bounds__(256) void trilinear_fwd_x4_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                               Axis az, Axis ay, Axis ax, int planes,
                                                               const float* __restrict__ coef, int relu) {
    const int Wo4 = ax.out >> 2;
    const int So4 = az.out * ay.out * Wo4;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= So4) return;
    const int xq = e % Wo4, yo = (e / Wo4) % ay.out, zo = e / (Wo4 * ay.out);
    int z0, z1, y0, y1;
    float a0, a1, b0, b1;
    src_index(az, zo, z0, z1, a0, a1);
    src_index(ay, yo, y0, y1, b0, b1);
    int r0[4], r1[4];
    float c0[4], c1[4];
    int xs = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int i0, i1;
        src_index(ax, 4 * xq + k, i0, i1, c0[k], c1[k]);
        if (k == 0) xs = i0 < ax.in - 4 ? i0 : ax.in - 4;     // keep the 16-byte load inside the row
        r0[k] = i0 - xs;
        r1[k] = i1 - xs;
    }
    const int H = ay.in, W = ax.in;
    const int64_t Si = (int64_t)az.in * H * W, So = (int64_t)az.out * ay.out * ax.out;
    const int o00 = (z0 * H + y0) * W + xs, o01 = (z0 * H + y1) * W + xs;
    const int o10 = (z1 * H + y0) * W + xs, o11 = (z1 * H + y1) * W + xs;
    const int p0 = blockIdx.y * TRI_CPT;
    const float lo = relu ? 0.f : -INFINITY;
#pragma unroll
    for (int u = 0; u < TRI_CPT; ++u) {
        const int plane = p0 + u;
        if (plane >= planes) break;
        const float* p = x + (int64_t)plane * Si;
        float v00[4], v01[4], v10[4], v11[4];
        __builtin_memcpy(v00, p + o00, 16);     // 4-byte aligned 16-byte loads (global_load_dwordx4)
        __builtin_memcpy(v01, p + o01, 16);
        __builtin_memcpy(v10, p + o10, 16);
        __builtin_memcpy(v11, p + o11, 16);
        if (coef) {   // the source is a RAW conv output: interpolate act(a*x + b) (normalise + ReLU on load)
            const float ca = coef[2 * plane], cb = coef[2 * plane + 1];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v00[j] = fmaxf(fmaf(ca, v00[j], cb), lo); v01[j] = fmaxf(fmaf(ca, v01[j], cb), lo);
                v10[j] = fmaxf(fmaf(ca, v10[j], cb), lo); v11[j] = fmaxf(fmaf(ca, v11[j], cb), lo);
            }
        }
        float t[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = a0 * (b0 * v00[j] + b1 * v01[j]) + a1 * (b0 * v10[j] + b1 * v11[j]);
        float4 o;
        o.x = c0[0] * sel4(t, r0[0]) + c1[0] * sel4(t, r1[0]);
        o.y = c0[1] * sel4(t, r0[1]) + c1[1] * sel4(t, r1[1]);
        o.z = c0[2] * sel4(t, r0[2]) + c1[2] * sel4(t, r1[2]);
        o.w = c0[3] * sel4(t, r0[3]) + c1[3] * sel4(t, r1[3]);
        *reinterpret_cast<float4*>(y + (int64_t)plane * So + 4 * (int64_t)e) = o;
    }
}

// LDS-tiled form of the same kernel for magnifications of 2x and more along z and y (the DC3D decoder's x2 resizes):
// a block produces a TT_Z x TT_Y x Wo tile of TT_CPT planes from the <= TT_RZ x TT_RY source rows the tile touches,
// fetched ONCE as aligned 16-byte row segments (activated once per source element when the source is lazy) instead of
// four unaligned 16-byte loads per 16-byte store through the vector L1 (the x4 kernel: 3.4 TB/s), and with the z/y
// weights multiplied out and the x blend as a small matrix (fewer vector instructions; the same trilinear form, the
// products rounded in a different order than in trilinear_fwd_x4_kernel).
constexpr int TT_Z = 4, TT_Y = 8, TT_RZ = 4, TT_RY = 6, TT_CPT = 4, TT_MAXW = 128;
__global__ __launch_bounds__(256) void trilinear_fwd_tile_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                 Axis az, Axis ay, Axis ax, int planes, int nty,
                                                                 const float* __restrict__ coef, int relu) {
    extern __shared__ __attribute__((aligned(16))) float tt_tile[];     // [TT_CPT][TT_RZ * TT_RY][ax.in]
    const int tid = threadIdx.x;
    const int Win = ax.in, H = ay.in;
    const int zo0 = (blockIdx.x / nty) * TT_Z, yo0 = (blockIdx.x % nty) * TT_Y;
    const int zo1 = min(zo0 + TT_Z, az.out) - 1, yo1 = min(yo0 + TT_Y, ay.out) - 1;
    int zlo, zhi, ylo, yhi, dmy;
    float f0, f1;
    src_index(az, zo0, zlo, dmy, f0, f1);
    src_index(az, zo1, dmy, zhi, f0, f1);
    src_index(ay, yo0, ylo, dmy, f0, f1);
    src_index(ay, yo1, dmy, yhi, f0, f1);
    const int ny = yhi - ylo + 1, nrow = (zhi - zlo + 1) * ny;          // <= TT_RY, TT_RZ * TT_RY (host: scales <= 0.5)
    const int p0 = blockIdx.y * TT_CPT;
    const int np = min(TT_CPT, planes - p0);
    const int64_t Si = (int64_t)az.in * H * Win, So = (int64_t)az.out * ay.out * ax.out;
    const int pstride = TT_RZ * TT_RY * Win;
    const float lo = relu ? 0.f : -INFINITY;
    {   // stage the source rows: 16-byte segments (Win % 4 == 0, x 16-byte aligned: host)
        const int W4 = Win >> 2, per_plane = nrow * W4;
        for (int e = tid; e < np * per_plane; e += 256) {
            const int u = e / per_plane, r = e - u * per_plane;
            const int row = r / W4, c4 = r - row * W4;
            const int gz = zlo + row / ny, gy = ylo + row % ny;
            float4 v = *reinterpret_cast<const float4*>(x + (p0 + u) * Si + ((int64_t)gz * H + gy) * Win + 4 * c4);
            if (coef) {   // the source is a RAW conv output: interpolate act(a*x + b) (normalise + ReLU on load)
                const float ca = coef[2 * (p0 + u)], cb = coef[2 * (p0 + u) + 1];
                v.x = fmaxf(fmaf(ca, v.x, cb), lo); v.y = fmaxf(fmaf(ca, v.y, cb), lo);
                v.z = fmaxf(fmaf(ca, v.z, cb), lo); v.w = fmaxf(fmaf(ca, v.w, cb), lo);
            }
            *reinterpret_cast<float4*>(tt_tile + u * pstride + row * Win + 4 * c4) = v;
        }
    }
    __syncthreads();
    const int Wo4 = ax.out >> 2;
    const int nout = TT_Z * TT_Y * Wo4;
    const bool fixed_x = (256 % Wo4) == 0;       // then a thread keeps its x group over the tile's rows
    // x blend as a 4x4 matrix over the four source columns [xs, xs + 3] (row k: c0 at column i0(k) - xs, c1 at i1(k) - xs,
    // zeros elsewhere): 16 multiply-adds per four outputs instead of 8 plus 24 selects -- this kernel is bound by its
    // vector instructions, not by memory (a non-finite source value therefore spreads over its group of four outputs)
    int xs = 0, xq_have = -1;
    float m[4][4];
    for (int o = tid; o < nout; o += 256) {
        const int xq = o % Wo4, rr = o / Wo4;
        const int yo = yo0 + rr % TT_Y, zo = zo0 + rr / TT_Y;
        if (zo > zo1 || yo > yo1) continue;
        if (!fixed_x || xq_have != xq) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int i0, i1;
                float c0, c1;
                src_index(ax, 4 * xq + k, i0, i1, c0, c1);
                if (k == 0) xs = i0 < ax.in - 4 ? i0 : ax.in - 4;     // the four source columns [xs, xs + 3]
#pragma unroll
                for (int j = 0; j < 4; ++j) m[k][j] = (i0 - xs == j ? c0 : 0.f) + (i1 - xs == j ? c1 : 0.f);
            }
            xq_have = xq;
        }
        int z0, z1, y0, y1;
        float a0, a1, b0, b1;
        src_index(az, zo, z0, z1, a0, a1);
        src_index(ay, yo, y0, y1, b0, b1);
        const int l00 = ((z0 - zlo) * ny + (y0 - ylo)) * Win + xs, l01 = ((z0 - zlo) * ny + (y1 - ylo)) * Win + xs;
        const int l10 = ((z1 - zlo) * ny + (y0 - ylo)) * Win + xs, l11 = ((z1 - zlo) * ny + (y1 - ylo)) * Win + xs;
        float* yp = y + p0 * So + ((int64_t)zo * ay.out + yo) * ax.out + 4 * xq;
        const float w00 = a0 * b0, w01 = a0 * b1, w10 = a1 * b0, w11 = a1 * b1;
        for (int u = 0; u < np; ++u) {
            const float* t0 = tt_tile + u * pstride;
            float t[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                t[j] = fmaf(w11, t0[l11 + j], fmaf(w10, t0[l10 + j], fmaf(w01, t0[l01 + j], w00 * t0[l00 + j])));
            float4 ov;
            ov.x = fmaf(m[0][3], t[3], fmaf(m[0][2], t[2], fmaf(m[0][1], t[1], m[0][0] * t[0])));
            ov.y = fmaf(m[1][3], t[3], fmaf(m[1][2], t[2], fmaf(m[1][1], t[1], m[1][0] * t[0])));
            ov.z = fmaf(m[2][3], t[3], fmaf(m[2][2], t[2], fmaf(m[2][1], t[1], m[2][0] * t[0])));
            ov.w = fmaf(m[3][3], t[3], fmaf(m[3][2], t[2], fmaf(m[3][1], t[1], m[3][0] * t[0])));
            {   // written once, read much later by the next conv: a streaming store (-4 % at 64^3 -> 128^3, -20 % at 16^3 -> 32^3)
                typedef float tt_f32x4 __attribute__((ext_vector_type(4)));
                const tt_f32x4 w = {ov.x, ov.y, ov.z, ov.w};
                __builtin_nontemporal_store(w, reinterpret_cast<tt_f32x4*>(yp + u * So));
            }
        }
    }
}

// Adjoint in gather form.  For input index i the outputs that touch it form the contiguous range
// [lo, hi] = {o : i0(o) in {i-1, i}}; each contributes l0(o) if i0(o)==i plus l1(o) if i1(o)==i.
__device__ __forceinline__ void touch_range(const Axis& a, int i, int& lo, int& hi) {
    if (a.out <= 1 || a.scale <= 0.f) { lo = 0; hi = a.out - 1; return; }
    // conservative bounds from the inverse map, then clamp
    const float inv = 1.f / a.scale;
    const float sh = a.half ? 0.5f : 0.f;
    lo = (int)floorf(((float)(i - 1) + sh) * inv - sh) - 1;
    hi = (int)ceilf(((float)(i + 1) + sh) * inv - sh) + 1;
    if (lo < 0) lo = 0;
    if (hi > a.out - 1) hi = a.out - 1;
}

__device__ __forceinline__ float touch_weight(const Axis& a, int o, int i) {
    int i0, i1;
    float l0, l1;
    src_index(a, o, i0, i1, l0, l1);
    float w = 0.f;
    if (i0 == i) w += l0;
    if (i1 == i) w += l1;   // when i1 == i0 (last row) both weights land on the same input, as in ATen's backward
    return w;
}

// per-axis taps of one input index: outputs lo..lo+MAXT-1 with their weights (0 where they do not touch it)
constexpr int MAXT = 6;
struct Taps {
    int lo, n;
    float w[MAXT];
};
__device__ __forceinline__ Taps axis_taps(const Axis& a, int i) {
    Taps t;
    int lo, hi;
    touch_range(a, i, lo, hi);
    // shrink to the outputs that really touch input i
    while (lo <= hi && touch_weight(a, lo, i) == 0.f) ++lo;
    while (hi >= lo && touch_weight(a, hi, i) == 0.f) --hi;
    t.lo = lo;
    t.n = hi - lo + 1;
#pragma unroll
    for (int k = 0; k < MAXT; ++k) t.w[k] = (k < t.n) ? touch_weight(a, lo + k, i) : 0.f;
    return t;
}

// host-side bound on the taps per input index, so that MAXT is never exceeded silently
static int max_taps_host_scale(double s) { return s > 0.0 ? (int)(2.0 / s) + 3 : 1 << 20; }
static int max_taps_host(int in, int out) {
    if (out <= 1 || in <= 1) return out;
    // outputs o with floor(o*s) in {i-1, i}: at most ceil(2/s)+1
    const double s = (double)(in - 1) / (double)(out - 1);
    return (int)(2.0 / s) + 2;
}

__global__ __launch_bounds__(256) void trilinear_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx,
                                                            Axis az, Axis ay, Axis ax, int planes) {
    const int S = az.in * ay.in * ax.in;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= S) return;
    const int xi = e % ax.in, yi = (e / ax.in) % ay.in, zi = e / (ax.in * ay.in);
    const Taps tz = axis_taps(az, zi), ty = axis_taps(ay, yi), tx = axis_taps(ax, xi);
    const int So = az.out * ay.out * ax.out;
    const int p0 = blockIdx.y * TRI_CPT;
    for (int u = 0; u < TRI_CPT; ++u) {
        const int plane = p0 + u;
        if (plane >= planes) break;
        const float* p = dy + (int64_t)plane * So;
        float acc = 0.f;
        for (int kz = 0; kz < tz.n; ++kz) {
            float zacc = 0.f;
            for (int ky = 0; ky < ty.n; ++ky) {
                const float* row = p + ((int64_t)(tz.lo + kz) * ay.out + ty.lo + ky) * ax.out + tx.lo;
                float racc = 0.f;
#pragma unroll
                for (int kx = 0; kx < MAXT; ++kx)
                    if (kx < tx.n) racc += tx.w[kx] * row[kx];
                zacc += ty.w[ky] * racc;
            }
            acc += tz.w[kz] * zacc;
        }
        dx[(int64_t)plane * S + e] = acc;
    }
}

// Two-stage adjoint, stage 1: reduce along z only.  tmp[plane][zi][yo][xo] = sum_kz wz * dy[plane][tz.lo+kz][yo][xo],
// four x at a time (rows are contiguous in xo: coalesced 16-byte loads, no index arithmetic per tap).  Stage 2 is
// trilinear_bwd_kernel itself on tmp with an identity z axis.  A coarse output then costs ~4 float4 + 16 scalar
// loads instead of 64 scalar ones, and the strided (every second x) gathers run over a tensor half the size.
__global__ __launch_bounds__(256) void trilinear_bwd_z_kernel(const float* __restrict__ dy, float* __restrict__ tmp,
                                                              Axis az, int Ho, int Wo, int planes) {
    const int W4 = Wo >> 2;
    const int per_plane4 = az.in * Ho * W4;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= per_plane4) return;
    const int zi = e / (Ho * W4);
    const int rem = e - zi * (Ho * W4);           // (yo, xo4) linear = offset/4 inside a z slab
    const Taps tz = axis_taps(az, zi);
    const int64_t slab = (int64_t)Ho * Wo;         // floats per z slab of dy / tmp
    const int64_t So = (int64_t)az.out * slab, St = (int64_t)az.in * slab;
    const int p0 = blockIdx.y * TRI_CPT;
    for (int u = 0; u < TRI_CPT; ++u) {
        const int plane = p0 + u;
        if (plane >= planes) break;
        const float* p = dy + (int64_t)plane * So + (int64_t)tz.lo * slab + 4 * (int64_t)rem;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int kz = 0; kz < MAXT; ++kz)
            if (kz < tz.n) {
                const float4 v = *reinterpret_cast<const float4*>(p + (int64_t)kz * slab);
                const float w = tz.w[kz];
                acc.x += w * v.x; acc.y += w * v.y; acc.z += w * v.z; acc.w += w * v.w;
            }
        *reinterpret_cast<float4*>(tmp + (int64_t)plane * St + (int64_t)zi * slab + 4 * (int64_t)rem) = acc;
    }
}

// Stage 2 of the two-stage adjoint with 16-byte rows: a thread owns FOUR consecutive inputs xi0..xi0+3 of one (z, y)
// row.  For magnifications between 1x and 2.2x along x (0.4545 <= scale <= 1) every output that touches them lies in
// the 16 outputs [base, base + 15], base = (first touching output - 1) rounded down to a multiple of 4: per y tap ONE
// row segment of four aligned 16-byte loads feeds all four inputs through a dense 4 x 16 weight table (computed once
// per thread with the same touch_weight as the gather kernel; zeros where an output does not touch an input) -- 4
// loads per input and y tap instead of ~3 scalar ones per (input, x tap).
__global__ __launch_bounds__(256) void trilinear_bwd_yx4_kernel(const float* __restrict__ tmp, float* __restrict__ dx, int D,
                                                                Axis ay, Axis ax, int planes) {
    const int W4 = ax.in >> 2;
    const int per_plane4 = D * ay.in * W4;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= per_plane4) return;
    const int x4 = e % W4, yi = (e / W4) % ay.in, zi = e / (W4 * ay.in);
    const int xi0 = 4 * x4;
    int base = 0;
    if (xi0 >= 1) {
        base = (int)ceilf((float)(xi0 - 1) / ax.scale) - 2;      // one below the first output with floor(scale*o) >= xi0-1, and one for rounding
        base = base < 0 ? 0 : base & ~3;
    }
    float wx[4][16];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int m = 0; m < 16; ++m) wx[j][m] = (base + m) < ax.out ? touch_weight(ax, base + m, xi0 + j) : 0.f;
    const Taps ty = axis_taps(ay, yi);
    const int Ho = ay.out, Wo = ax.out;
    const int64_t St = (int64_t)D * Ho * Wo, S = (int64_t)D * ay.in * ax.in;
    const int p0 = blockIdx.y * TRI_CPT;
    for (int u = 0; u < TRI_CPT; ++u) {
        const int plane = p0 + u;
        if (plane >= planes) break;
        const float* p = tmp + (int64_t)plane * St + ((int64_t)zi * Ho + ty.lo) * Wo + base;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int ky = 0; ky < ty.n; ++ky) {
            const float* row = p + (int64_t)ky * Wo;
            float v[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {       // (a 16-byte piece beyond the row end is not read: its weights are 0)
                const float4 t = (base + 4 * q) < Wo ? *reinterpret_cast<const float4*>(row + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
                v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
            }
            const float wy = ty.w[ky];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float r = 0.f;
#pragma unroll
                for (int m = 0; m < 16; ++m) r = fmaf(wx[j][m], v[m], r);
                acc[j] = fmaf(wy, r, acc[j]);
            }
        }
        *reinterpret_cast<float4*>(dx + (int64_t)plane * S + 4 * (int64_t)e) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
}

// general fallback (any magnification): plane per blockIdx.y, weights recomputed inside the loops
__global__ __launch_bounds__(256) void trilinear_bwd_general_kernel(const float* __restrict__ dy,
                                                                    float* __restrict__ dx, Axis az, Axis ay, Axis ax) {
    const int64_t plane = blockIdx.y;
    const int S = az.in * ay.in * ax.in;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= S) return;
    const int xi = e % ax.in, yi = (e / ax.in) % ay.in, zi = e / (ax.in * ay.in);
    int zl, zh, yl, yh, xl, xh;
    touch_range(az, zi, zl, zh);
    touch_range(ay, yi, yl, yh);
    touch_range(ax, xi, xl, xh);
    const float* p = dy + plane * ((int64_t)az.out * ay.out * ax.out);
    float acc = 0.f;
    for (int zo = zl; zo <= zh; ++zo) {
        const float wz = touch_weight(az, zo, zi);
        if (wz == 0.f) continue;
        for (int yo = yl; yo <= yh; ++yo) {
            const float wy = touch_weight(ay, yo, yi);
            if (wy == 0.f) continue;
            const float* row = p + ((int64_t)zo * ay.out + yo) * ax.out;
            float racc = 0.f;
            for (int xo = xl; xo <= xh; ++xo) racc += touch_weight(ax, xo, xi) * row[xo];
            acc += wz * wy * racc;
        }
    }
    dx[plane * (int64_t)S + e] = acc;
}

// ---------------------------------------------------------------- crop + concat
__global__ __launch_bounds__(256) void crop_concat_fwd_kernel(const float* __restrict__ t1, const float* __restrict__ t2,
                                                              float* __restrict__ out, int C1, int C2, int D, int H,
                                                              int W, int D2, int H2, int W2, int oz, int oy, int ox) {
    const int Ct = C1 + C2;
    const int64_t plane = blockIdx.y;  // n*Ct + c
    const int n = (int)(plane / Ct), c = (int)(plane % Ct);
    const int S = D * H * W;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= S) return;
    float v;
    if (c < C1) {
        v = t1[((int64_t)n * C1 + c) * S + e];
    } else {
        const int xi = e % W, yi = (e / W) % H, zi = e / (W * H);
        v = t2[((int64_t)n * C2 + (c - C1)) * ((int64_t)D2 * H2 * W2) + ((int64_t)(zi + oz) * H2 + yi + oy) * W2 + xi + ox];
    }
    out[plane * (int64_t)S + e] = v;
}

__global__ __launch_bounds__(256) void crop_concat_bwd1_kernel(const float* __restrict__ dout, float* __restrict__ dt1,
                                                               int C1, int C2, int S) {
    const int64_t plane = blockIdx.y;  // n*C1 + c
    const int n = (int)(plane / C1), c = (int)(plane % C1);
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= S) return;
    dt1[plane * (int64_t)S + e] = dout[((int64_t)n * (C1 + C2) + c) * S + e];
}

__global__ __launch_bounds__(256) void crop_concat_bwd2_kernel(const float* __restrict__ dout, float* __restrict__ dt2,
                                                               int C1, int C2, int D, int H, int W, int D2, int H2,
                                                               int W2, int oz, int oy, int ox) {
    const int64_t plane = blockIdx.y;  // n*C2 + c
    const int n = (int)(plane / C2), c = (int)(plane % C2);
    const int S2 = D2 * H2 * W2;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= S2) return;
    const int xi = e % W2 - ox, yi = (e / W2) % H2 - oy, zi = e / (W2 * H2) - oz;
    float v = 0.f;
    if (xi >= 0 && xi < W && yi >= 0 && yi < H && zi >= 0 && zi < D)
        v = dout[((int64_t)n * (C1 + C2) + C1 + c) * ((int64_t)D * H * W) + ((int64_t)zi * H + yi) * W + xi];
    dt2[plane * (int64_t)S2 + e] = v;
}


// ---------------------------------------------------------------- Rotate3DXOneShot (data_transforms.py:1186-1208)
// F.grid_sample(x, F.affine_grid(theta, x.size())) with the defaults the reference relies on (bilinear = trilinear for
// 5-D, zeros padding, align_corners=False), theta [3][4] the same for every sample: output voxel (d,h,w) has the
// normalised coordinates xn = (2w+1)/W - 1, yn, zn; the source point is theta @ (xn, yn, zn, 1) = (gx, gy, gz) (x <-> W,
// y <-> H, z <-> D), un-normalised as ix = ((gx + 1) * W - 1) / 2; eight taps, those outside the volume contribute 0.
struct Affine34 {
    float m[12];
};
__device__ __forceinline__ void affine_source(const Affine34& t, int D, int H, int W, int d, int h, int w, float& iz, float& iy,
                                              float& ix) {
    const float xn = (2.f * (float)w + 1.f) / (float)W - 1.f;
    const float yn = (2.f * (float)h + 1.f) / (float)H - 1.f;
    const float zn = (2.f * (float)d + 1.f) / (float)D - 1.f;
    const float gx = t.m[0] * xn + t.m[1] * yn + t.m[2] * zn + t.m[3];
    const float gy = t.m[4] * xn + t.m[5] * yn + t.m[6] * zn + t.m[7];
    const float gz = t.m[8] * xn + t.m[9] * yn + t.m[10] * zn + t.m[11];
    ix = ((gx + 1.f) * (float)W - 1.f) * 0.5f;
    iy = ((gy + 1.f) * (float)H - 1.f) * 0.5f;
    iz = ((gz + 1.f) * (float)D - 1.f) * 0.5f;
}
// BWD = false: y[plane][e] = sum over taps; BWD = true: dx[plane][tap] += w * dy[plane][e] (float atomics: the scatter
// form of the adjoint; the order of the additions, and with it the last bits, is not fixed -- as in ATen's backward)
template <bool BWD>
__global__ __launch_bounds__(256) void affine_sample_kernel(const float* __restrict__ src, float* dst, Affine34 t, int D, int H,
                                                            int W, int planes) {
    const int S = D * H * W;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= S) return;
    const int w = e % W, h = (e / W) % H, d = e / (W * H);
    float iz, iy, ix;
    affine_source(t, D, H, W, d, h, w, iz, iy, ix);
    const float fz = floorf(iz), fy = floorf(iy), fx = floorf(ix);
    const int z0 = (int)fz, y0 = (int)fy, x0 = (int)fx;
    const float lz = iz - fz, ly = iy - fy, lx = ix - fx;
    int off[8];
    float wt[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int zz = z0 + (k >> 2), yy = y0 + ((k >> 1) & 1), xx = x0 + (k & 1);
        const bool in = zz >= 0 && zz < D && yy >= 0 && yy < H && xx >= 0 && xx < W;
        off[k] = in ? (zz * H + yy) * W + xx : -1;
        wt[k] = ((k >> 2) ? lz : 1.f - lz) * (((k >> 1) & 1) ? ly : 1.f - ly) * ((k & 1) ? lx : 1.f - lx);
    }
    const int p0 = blockIdx.y * TRI_CPT;
    for (int u = 0; u < TRI_CPT; ++u) {
        const int plane = p0 + u;
        if (plane >= planes) break;
        if (!BWD) {
            const float* p = src + (int64_t)plane * S;
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (off[k] >= 0) acc += wt[k] * p[off[k]];
            dst[(int64_t)plane * S + e] = acc;
        } else {
            const float g = src[(int64_t)plane * S + e];
            float* p = dst + (int64_t)plane * S;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (off[k] >= 0) atomicAdd(p + off[k], wt[k] * g);
        }
    }
}

// ---------------------------------------------------------------- OneShot transforms
// F.interpolate(mode='nearest'): src = min(floor(dst * scale), in-1), scale = in/out (ATen nearest_neighbor_compute_source_index)
__global__ __launch_bounds__(256) void resize_nearest_kernel(const float* __restrict__ x, float* __restrict__ y, int D, int H,
                                                             int W, int Do, int Ho, int Wo, float sz, float sy, float sx) {
    const int64_t plane = blockIdx.y;
    const int So = Do * Ho * Wo;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= So) return;
    const int xo = e % Wo, yo = (e / Wo) % Ho, zo = e / (Wo * Ho);
    const int zi = min((int)floorf((float)zo * sz), D - 1), yi = min((int)floorf((float)yo * sy), H - 1),
              xi = min((int)floorf((float)xo * sx), W - 1);
    y[plane * So + e] = x[plane * ((int64_t)D * H * W) + ((int64_t)zi * H + yi) * W + xi];
}

// torch.flip / torch.rot90 / transposes of the spatial axes: out[o0,o1,o2] = in[i], i[perm[k]] = flip[k] ? n-1-o[k] : o[k]
struct PermFlip {
    int od[3];      // output extents
    int perm[3];    // output axis k reads input axis perm[k]
    int flip[3];
    int istride[3]; // element stride of input axis a
};
__global__ __launch_bounds__(256) void permute_flip_kernel(const float* __restrict__ x, float* __restrict__ y, PermFlip p) {
    const int64_t plane = blockIdx.y;
    const int So = p.od[0] * p.od[1] * p.od[2];
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= So) return;
    const int o2 = e % p.od[2], o1 = (e / p.od[2]) % p.od[1], o0 = e / (p.od[2] * p.od[1]);
    const int o[3] = {o0, o1, o2};
    int64_t off = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) off += (int64_t)(p.flip[k] ? p.od[k] - 1 - o[k] : o[k]) * p.istride[p.perm[k]];
    y[plane * So + e] = x[plane * So + off];
}

static inline Axis make_axis(int in, int out) {
    Axis a;
    a.in = in;
    a.out = out;
    a.scale = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
    a.half = 0;
    return a;
}

// align_corners=False: ATen area_pixel_compute_scale = in/out, or 1/scale_factor when the caller passed one
static inline Axis make_axis_half(int in, int out, float given_scale) {
    Axis a;
    a.in = in;
    a.out = out;
    a.scale = given_scale > 0.f ? given_scale : (float)in / (float)out;
    a.half = 1;
    return a;
}

static int check_planes(const char* who, int64_t planes, int64_t S) {
    DRAM_REQUIRE(planes > 0 && planes <= 65535, "%s: N*C=%lld out of range (1..65535)", who, (long long)planes);
    DRAM_REQUIRE(S > 0 && S < 0x7fffffffLL, "%s: spatial size out of range", who);
    return DRAM_OK;
}

}  // namespace dram

using namespace dram;

extern "C" int dram_maxpool3d_2_fwd(const float* x, float* out, uint8_t* idx, int N, int C, int D, int H, int W,
                                    void* stream) {
    DRAM_REQUIRE(x && out && idx, "maxpool3d_2_fwd: null pointer");
    DRAM_REQUIRE(D >= 2 && H >= 2 && W >= 2, "maxpool3d_2_fwd: spatial size below the 2x2x2 window");
    int rc = check_planes("maxpool3d_2_fwd", (int64_t)N * C, (int64_t)D * H * W);
    if (rc) return rc;
    const int Do = D / 2, Ho = H / 2, Wo = W / 2;
    dim3 grid(cdiv(Do * Ho * Wo, 256), N * C);
    hipLaunchKernelGGL(maxpool2_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, out, idx, D, H, W, Do, Ho, Wo,
                       (const float*)nullptr, 0);
    return check_launch("maxpool3d_2_fwd");
}

// max_pool3d(act(coef * x)) with the norm (+ReLU) applied on load; coef = per-row {a, b} (null: plain input)
extern "C" int dram_maxpool3d_2_fwd_lazy(const float* x, const float* coef, int relu, float* out, uint8_t* idx, int N, int C,
                                         int D, int H, int W, void* stream) {
    DRAM_REQUIRE(x && out && idx, "maxpool3d_2_fwd_lazy: null pointer");
    DRAM_REQUIRE(D >= 2 && H >= 2 && W >= 2, "maxpool3d_2_fwd_lazy: spatial size below the 2x2x2 window");
    int rc = check_planes("maxpool3d_2_fwd_lazy", (int64_t)N * C, (int64_t)D * H * W);
    if (rc) return rc;
    const int Do = D / 2, Ho = H / 2, Wo = W / 2;
    dim3 grid(cdiv(Do * Ho * Wo, 256), N * C);
    hipLaunchKernelGGL(maxpool2_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, out, idx, D, H, W, Do, Ho, Wo, coef,
                       relu);
    return check_launch("maxpool3d_2_fwd_lazy");
}

static int maxpool_bwd_run(const char* who, const float* dout, const uint8_t* idx, float* dx, int N, int C, int D, int H,
                           int W, int accumulate, void* stream) {
    DRAM_REQUIRE(dout && idx && dx, "%s: null pointer", who);
    DRAM_REQUIRE(D >= 2 && H >= 2 && W >= 2, "%s: spatial size below the 2x2x2 window", who);
    int rc = check_planes(who, (int64_t)N * C, (int64_t)D * H * W);
    if (rc) return rc;
    if (W % 4 == 0 && ((((uintptr_t)dout) & 7) | (((uintptr_t)idx) & 1) | (((uintptr_t)dx) & 15)) == 0) {
        dim3 grid(cdiv(D * H * (W / 4), 256), N * C);
        hipLaunchKernelGGL(maxpool2_bwd_vec_kernel, grid, dim3(256), 0, (hipStream_t)stream, dout, idx, dx, D, H, W,
                           D / 2, H / 2, W / 2, accumulate);
    } else {
        dim3 grid(cdiv(D * H * W, 256), N * C);
        hipLaunchKernelGGL(maxpool2_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, dout, idx, dx, D, H, W, D / 2,
                           H / 2, W / 2, accumulate);
    }
    return check_launch(who);
}

extern "C" int dram_maxpool3d_2_bwd(const float* dout, const uint8_t* idx, float* dx, int N, int C, int D, int H,
                                    int W, void* stream) {
    return maxpool_bwd_run("maxpool3d_2_bwd", dout, idx, dx, N, C, D, H, W, 0, stream);
}

// dx += scatter(dout): the pooled branch's gradient added onto the skip branch's, in one pass
extern "C" int dram_maxpool3d_2_bwd_acc(const float* dout, const uint8_t* idx, float* dx, int N, int C, int D, int H,
                                        int W, void* stream) {
    return maxpool_bwd_run("maxpool3d_2_bwd_acc", dout, idx, dx, N, C, D, H, W, 1, stream);
}

// 16-byte path of the forward resize: x magnification >= 1.67, rows of >= 4 source columns, 16-byte aligned output rows
static bool tri_fwd_x4_ok(const float* y, const Axis& ax) {
    return ax.half == 0 && ax.in >= 4 && (ax.out % 4) == 0 && ax.scale <= 0.6f && (((uintptr_t)y) & 15) == 0;
}
static int tri_fwd_launch(const float* x, float* y, const Axis& az, const Axis& ay, const Axis& ax, int planes, const float* coef,
                          int relu, hipStream_t st) {
    const bool tiled = tri_fwd_x4_ok(y, ax) && az.half == 0 && ay.half == 0 && az.scale <= 0.5f && ay.scale <= 0.5f &&
                       (ax.in % 4) == 0 && ax.in <= TT_MAXW && (((uintptr_t)x) & 15) == 0 && cdiv(planes, TT_CPT) <= 65535 &&
                       getenv("DRAM_TRI_NO_TILE") == nullptr;
    if (tiled) {
        const int ntz = cdiv(az.out, TT_Z), nty = cdiv(ay.out, TT_Y);
        const size_t lds = (size_t)TT_CPT * TT_RZ * TT_RY * ax.in * sizeof(float);      // <= 48 KB
        hipLaunchKernelGGL(trilinear_fwd_tile_kernel, dim3(ntz * nty, cdiv(planes, TT_CPT)), dim3(256), lds, st, x, y, az, ay, ax,
                           planes, nty, coef, relu);
    } else if (tri_fwd_x4_ok(y, ax)) {
        dim3 grid(cdiv(az.out * ay.out * (ax.out / 4), 256), cdiv(planes, TRI_CPT));
        hipLaunchKernelGGL(trilinear_fwd_x4_kernel, grid, dim3(256), 0, st, x, y, az, ay, ax, planes, coef, relu);
    } else {
        dim3 grid(cdiv(az.out * ay.out * ax.out, 256), cdiv(planes, TRI_CPT));
        hipLaunchKernelGGL(trilinear_fwd_kernel, grid, dim3(256), 0, st, x, y, az, ay, ax, planes, coef, relu);
    }
    return DRAM_OK;
}

extern "C" int dram_upsample_trilinear_ac_fwd(const float* x, float* y, int N, int C, int D, int H, int W, int Do,
                                              int Ho, int Wo, void* stream) {
    DRAM_REQUIRE(x && y, "upsample_trilinear_ac_fwd: null pointer");
    DRAM_REQUIRE(D > 0 && H > 0 && W > 0 && Do > 0 && Ho > 0 && Wo > 0, "upsample_trilinear_ac_fwd: bad sizes");
    int rc = check_planes("upsample_trilinear_ac_fwd", (int64_t)N * C, (int64_t)Do * Ho * Wo);
    if (rc) return rc;
    tri_fwd_launch(x, y, make_axis(D, Do), make_axis(H, Ho), make_axis(W, Wo), N * C, nullptr, 0, (hipStream_t)stream);
    return check_launch("upsample_trilinear_ac_fwd");
}

// upsample(act(coef * x)) with the norm (+ReLU) applied on load; coef = per-row {a, b} (null: plain input)
extern "C" int dram_upsample_trilinear_ac_fwd_lazy(const float* x, const float* coef, int relu, float* y, int N, int C, int D,
                                                   int H, int W, int Do, int Ho, int Wo, void* stream) {
    DRAM_REQUIRE(x && y, "upsample_trilinear_ac_fwd_lazy: null pointer");
    DRAM_REQUIRE(D > 0 && H > 0 && W > 0 && Do > 0 && Ho > 0 && Wo > 0, "upsample_trilinear_ac_fwd_lazy: bad sizes");
    int rc = check_planes("upsample_trilinear_ac_fwd_lazy", (int64_t)N * C, (int64_t)Do * Ho * Wo);
    if (rc) return rc;
    tri_fwd_launch(x, y, make_axis(D, Do), make_axis(H, Ho), make_axis(W, Wo), N * C, coef, relu, (hipStream_t)stream);
    return check_launch("upsample_trilinear_ac_fwd_lazy");
}

// Two-stage backward (z first, then y/x) through a caller-provided workspace; planes are processed in groups
// that fit the workspace.  Falls back to the single-stage kernel when the shape does not qualify.
static bool tri_two_stage_ok(const float* dy, int D, int H, int W, int Do, int Ho, int Wo) {
    return Do > D && (Wo % 4) == 0 && (((uintptr_t)dy) & 15) == 0 && max_taps_host(D, Do) <= MAXT &&
           max_taps_host(H, Ho) <= MAXT && max_taps_host(W, Wo) <= MAXT && (int64_t)D * Ho * Wo < 0x7fffffffLL;
}

extern "C" size_t dram_upsample_trilinear_ac_bwd_ws_bytes(int N, int C, int D, int H, int W, int Do, int Ho, int Wo) {
    if (N <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0 || Do <= D || (Wo % 4) != 0) return 0;
    return (size_t)N * C * D * Ho * Wo * sizeof(float);
}

extern "C" int dram_upsample_trilinear_ac_bwd_ws(const float* dy, float* dx, void* ws, size_t ws_bytes, int N, int C,
                                                 int D, int H, int W, int Do, int Ho, int Wo, void* stream) {
    DRAM_REQUIRE(dy && dx, "upsample_trilinear_ac_bwd: null pointer");
    DRAM_REQUIRE(D > 0 && H > 0 && W > 0 && Do > 0 && Ho > 0 && Wo > 0, "upsample_trilinear_ac_bwd: bad sizes");
    const size_t per_plane = (size_t)D * Ho * Wo * sizeof(float);
    int64_t group = ws ? (int64_t)(ws_bytes / per_plane) : 0;
    group -= group % TRI_CPT;
    if (group < TRI_CPT || (((uintptr_t)ws) & 15) != 0 || !tri_two_stage_ok(dy, D, H, W, Do, Ho, Wo))
        return dram_upsample_trilinear_ac_bwd(dy, dx, N, C, D, H, W, Do, Ho, Wo, stream);
    int rc = check_planes("upsample_trilinear_ac_bwd", (int64_t)N * C, (int64_t)D * H * W);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int64_t planes = (int64_t)N * C;
    const Axis az = make_axis(D, Do), ay = make_axis(H, Ho), ax = make_axis(W, Wo), idz = make_axis(D, D);
    const int64_t So = (int64_t)Do * Ho * Wo, S = (int64_t)D * H * W;
    for (int64_t p0 = 0; p0 < planes; p0 += group) {
        const int g = (int)((planes - p0) < group ? (planes - p0) : group);
        hipLaunchKernelGGL(trilinear_bwd_z_kernel, dim3(cdiv(D * Ho * (Wo / 4), 256), cdiv(g, TRI_CPT)), dim3(256), 0, st,
                           dy + p0 * So, (float*)ws, az, Ho, Wo, g);
        if (ax.scale >= 0.4546f && ax.scale <= 1.f && (W % 4) == 0 && ((((uintptr_t)(dx + p0 * S)) & 15) == 0))
            hipLaunchKernelGGL(trilinear_bwd_yx4_kernel, dim3(cdiv(D * H * (W / 4), 256), cdiv(g, TRI_CPT)), dim3(256), 0, st,
                               (const float*)ws, dx + p0 * S, D, ay, ax, g);
        else
            hipLaunchKernelGGL(trilinear_bwd_kernel, dim3(cdiv(D * H * W, 256), cdiv(g, TRI_CPT)), dim3(256), 0, st,
                               (const float*)ws, dx + p0 * S, idz, ay, ax, g);
    }
    return check_launch("upsample_trilinear_ac_bwd(two-stage)");
}

extern "C" int dram_upsample_trilinear_ac_bwd(const float* dy, float* dx, int N, int C, int D, int H, int W, int Do,
                                              int Ho, int Wo, void* stream) {
    DRAM_REQUIRE(dy && dx, "upsample_trilinear_ac_bwd: null pointer");
    DRAM_REQUIRE(D > 0 && H > 0 && W > 0 && Do > 0 && Ho > 0 && Wo > 0, "upsample_trilinear_ac_bwd: bad sizes");
    int rc = check_planes("upsample_trilinear_ac_bwd", (int64_t)N * C, (int64_t)D * H * W);
    if (rc) return rc;
    if (max_taps_host(D, Do) <= MAXT && max_taps_host(H, Ho) <= MAXT && max_taps_host(W, Wo) <= MAXT) {
        dim3 grid(cdiv(D * H * W, 256), cdiv(N * C, TRI_CPT));
        hipLaunchKernelGGL(trilinear_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, dy, dx, make_axis(D, Do),
                           make_axis(H, Ho), make_axis(W, Wo), N * C);
    } else {   // magnification above ~2.5x on some axis
        dim3 grid(cdiv(D * H * W, 256), N * C);
        hipLaunchKernelGGL(trilinear_bwd_general_kernel, grid, dim3(256), 0, (hipStream_t)stream, dy, dx,
                           make_axis(D, Do), make_axis(H, Ho), make_axis(W, Wo));
    }
    return check_launch("upsample_trilinear_ac_bwd");
}

static int check_crop(const char* who, int D, int H, int W, int D2, int H2, int W2, int oz, int oy, int ox) {
    DRAM_REQUIRE(oz >= 0 && oy >= 0 && ox >= 0 && oz + D <= D2 && oy + H <= H2 && ox + W <= W2,
                 "%s: crop window outside the second tensor", who);
    return DRAM_OK;
}

extern "C" int dram_crop_concat_fwd(const float* t1, const float* t2, float* out, int N, int C1, int C2, int D, int H,
                                    int W, int D2, int H2, int W2, int oz, int oy, int ox, void* stream) {
    DRAM_REQUIRE(t1 && t2 && out, "crop_concat_fwd: null pointer");
    DRAM_REQUIRE(C1 > 0 && C2 > 0, "crop_concat_fwd: bad channel counts");
    int rc = check_planes("crop_concat_fwd", (int64_t)N * (C1 + C2), (int64_t)D * H * W);
    if (rc) return rc;
    if ((rc = check_crop("crop_concat_fwd", D, H, W, D2, H2, W2, oz, oy, ox))) return rc;
    dim3 grid(cdiv(D * H * W, 256), N * (C1 + C2));
    hipLaunchKernelGGL(crop_concat_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, t1, t2, out, C1, C2, D, H, W,
                       D2, H2, W2, oz, oy, ox);
    return check_launch("crop_concat_fwd");
}

extern "C" int dram_crop_concat_bwd(const float* dout, float* dt1, float* dt2, int N, int C1, int C2, int D, int H,
                                    int W, int D2, int H2, int W2, int oz, int oy, int ox, void* stream) {
    DRAM_REQUIRE(dout, "crop_concat_bwd: null pointer");
    DRAM_REQUIRE(C1 > 0 && C2 > 0, "crop_concat_bwd: bad channel counts");
    int rc = check_planes("crop_concat_bwd", (int64_t)N * (C1 + C2), (int64_t)D2 * H2 * W2);
    if (rc) return rc;
    if ((rc = check_crop("crop_concat_bwd", D, H, W, D2, H2, W2, oz, oy, ox))) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (dt1) {
        dim3 grid(cdiv(D * H * W, 256), N * C1);
        hipLaunchKernelGGL(crop_concat_bwd1_kernel, grid, dim3(256), 0, st, dout, dt1, C1, C2, D * H * W);
    }
    if (dt2) {
        dim3 grid(cdiv(D2 * H2 * W2, 256), N * C2);
        hipLaunchKernelGGL(crop_concat_bwd2_kernel, grid, dim3(256), 0, st, dout, dt2, C1, C2, D, H, W, D2, H2, W2,
                           oz, oy, ox);
    }
    return check_launch("crop_concat_bwd");
}

// ---- F.interpolate(mode='trilinear', align_corners=False) (Rescale3DOneShot on "#image" tensors,
//      dram/data_transforms.py:1202-1239); scale_* > 0: the 1/scale_factor ATen uses when the caller gave scale factors
extern "C" int dram_resize_trilinear_fwd(const float* x, float* y, int N, int C, int D, int H, int W, int Do, int Ho,
                                         int Wo, float scale_z, float scale_y, float scale_x, void* stream) {
    DRAM_REQUIRE(x && y, "resize_trilinear_fwd: null pointer");
    DRAM_REQUIRE(D > 0 && H > 0 && W > 0 && Do > 0 && Ho > 0 && Wo > 0, "resize_trilinear_fwd: bad sizes");
    int rc = check_planes("resize_trilinear_fwd", (int64_t)N * C, (int64_t)Do * Ho * Wo);
    if (rc) return rc;
    dim3 grid(cdiv(Do * Ho * Wo, 256), cdiv(N * C, TRI_CPT));
    hipLaunchKernelGGL(trilinear_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, y, make_axis_half(D, Do, scale_z),
                       make_axis_half(H, Ho, scale_y), make_axis_half(W, Wo, scale_x), N * C, (const float*)nullptr, 0);
    return check_launch("resize_trilinear_fwd");
}

extern "C" int dram_resize_trilinear_bwd(const float* dy, float* dx, int N, int C, int D, int H, int W, int Do, int Ho,
                                         int Wo, float scale_z, float scale_y, float scale_x, void* stream) {
    DRAM_REQUIRE(dy && dx, "resize_trilinear_bwd: null pointer");
    DRAM_REQUIRE(D > 0 && H > 0 && W > 0 && Do > 0 && Ho > 0 && Wo > 0, "resize_trilinear_bwd: bad sizes");
    int rc = check_planes("resize_trilinear_bwd", (int64_t)N * C, (int64_t)D * H * W);
    if (rc) return rc;
    const Axis az = make_axis_half(D, Do, scale_z), ay = make_axis_half(H, Ho, scale_y), ax = make_axis_half(W, Wo, scale_x);
    if (max_taps_host_scale(az.scale) <= MAXT && max_taps_host_scale(ay.scale) <= MAXT && max_taps_host_scale(ax.scale) <= MAXT) {
        dim3 grid(cdiv(D * H * W, 256), cdiv(N * C, TRI_CPT));
        hipLaunchKernelGGL(trilinear_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, dy, dx, az, ay, ax, N * C);
    } else {
        dim3 grid(cdiv(D * H * W, 256), N * C);
        hipLaunchKernelGGL(trilinear_bwd_general_kernel, grid, dim3(256), 0, (hipStream_t)stream, dy, dx, az, ay, ax);
    }
    return check_launch("resize_trilinear_bwd");
}

// ---- F.interpolate(mode='nearest') (Rescale3DOneShot on "#reference" tensors) ----
extern "C" int dram_resize_nearest(const float* x, float* y, int N, int C, int D, int H, int W, int Do, int Ho, int Wo,
                                   float scale_z, float scale_y, float scale_x, void* stream) {
    DRAM_REQUIRE(x && y, "resize_nearest: null pointer");
    DRAM_REQUIRE(D > 0 && H > 0 && W > 0 && Do > 0 && Ho > 0 && Wo > 0 && (int64_t)N * C <= 65535, "resize_nearest: bad sizes");
    int rc = check_planes("resize_nearest", (int64_t)N * C, (int64_t)Do * Ho * Wo);
    if (rc) return rc;
    const float sz = scale_z > 0.f ? scale_z : (float)D / (float)Do, sy = scale_y > 0.f ? scale_y : (float)H / (float)Ho,
                sx = scale_x > 0.f ? scale_x : (float)W / (float)Wo;
    hipLaunchKernelGGL(resize_nearest_kernel, dim3(cdiv(Do * Ho * Wo, 256), N * C), dim3(256), 0, (hipStream_t)stream, x, y, D,
                       H, W, Do, Ho, Wo, sz, sy, sx);
    return check_launch("resize_nearest");
}

// ---- torch.flip / torch.rot90 over the spatial axes (Flip3DOneShot, Rotate903DOneShot, data_transforms.py:1140-1181):
//      out[o] = in[i], i[perm[k]] = flip[k] ? n-1-o[k] : o[k]; (D,H,W) are the INPUT extents ----
extern "C" int dram_spatial_permute_flip(const float* x, float* y, int N, int C, int D, int H, int W, const int* perm,
                                         const int* flip, void* stream) {
    DRAM_REQUIRE(x && y && perm && flip, "spatial_permute_flip: null pointer");
    DRAM_REQUIRE(D > 0 && H > 0 && W > 0 && (int64_t)N * C <= 65535, "spatial_permute_flip: bad sizes");
    int seen = 0;
    for (int k = 0; k < 3; ++k) {
        DRAM_REQUIRE(perm[k] >= 0 && perm[k] < 3, "spatial_permute_flip: perm must be a permutation of 0,1,2");
        seen |= 1 << perm[k];
    }
    DRAM_REQUIRE(seen == 7, "spatial_permute_flip: perm must be a permutation of 0,1,2");
    int rc = check_planes("spatial_permute_flip", (int64_t)N * C, (int64_t)D * H * W);
    if (rc) return rc;
    const int id[3] = {D, H, W};
    PermFlip p;
    p.istride[0] = H * W; p.istride[1] = W; p.istride[2] = 1;
    for (int k = 0; k < 3; ++k) { p.od[k] = id[perm[k]]; p.perm[k] = perm[k]; p.flip[k] = flip[k] ? 1 : 0; }
    hipLaunchKernelGGL(permute_flip_kernel, dim3(cdiv(D * H * W, 256), N * C), dim3(256), 0, (hipStream_t)stream, x, y, p);
    return check_launch("spatial_permute_flip");
}

extern "C" int dram_affine_sample_fwd(const float* x, float* y, const float* theta12, int N, int C, int D, int H, int W,
                                      void* stream) {
    DRAM_REQUIRE(x && y && theta12, "affine_sample_fwd: null pointer");
    DRAM_REQUIRE(D > 0 && H > 0 && W > 0, "affine_sample_fwd: bad sizes");
    int rc = check_planes("affine_sample_fwd", (int64_t)N * C, (int64_t)D * H * W);
    if (rc) return rc;
    Affine34 t;
    for (int i = 0; i < 12; ++i) t.m[i] = theta12[i];
    hipLaunchKernelGGL(affine_sample_kernel<false>, dim3(cdiv(D * H * W, 256), cdiv(N * C, TRI_CPT)), dim3(256), 0,
                       (hipStream_t)stream, x, y, t, D, H, W, N * C);
    return check_launch("affine_sample_fwd");
}

extern "C" int dram_affine_sample_bwd(const float* dy, float* dx, const float* theta12, int N, int C, int D, int H, int W,
                                      void* stream) {
    DRAM_REQUIRE(dy && dx && theta12, "affine_sample_bwd: null pointer");
    DRAM_REQUIRE(D > 0 && H > 0 && W > 0, "affine_sample_bwd: bad sizes");
    int rc = check_planes("affine_sample_bwd", (int64_t)N * C, (int64_t)D * H * W);
    if (rc) return rc;
    Affine34 t;
    for (int i = 0; i < 12; ++i) t.m[i] = theta12[i];
    hipStream_t st = (hipStream_t)stream;
    (void)hipMemsetAsync(dx, 0, (size_t)N * C * D * H * W * sizeof(float), st);
    hipLaunchKernelGGL(affine_sample_kernel<true>, dim3(cdiv(D * H * W, 256), cdiv(N * C, TRI_CPT)), dim3(256), 0, st, dy, dx, t,
                       D, H, W, N * C);
    return check_launch("affine_sample_bwd");
}
