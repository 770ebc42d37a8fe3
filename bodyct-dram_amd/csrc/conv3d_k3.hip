// 3x3x3 / stride 1 / pad 1 convolution for fp32 NCDHW tensors on gfx950.
//
// Replaces the ATen conv3d dispatch (and its two backward kernels) issued by the
// nn.Conv3d modules of reference dram/parts.py:95,105,133,142,177,185.
//
// Design (MI355X): every 3x3x3 conv of DC3D with Cin >= 32 is FP32-FLOP bound
// (200-2500 FLOP/B), so all kernels are implicit GEMMs on the
// exact-fp32 matrix cores (v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32,
// 157 TFLOP/s peak = the fp32 vector peak but one VGPR per operand and the VALU
// left free).  Operand tiles are staged in LDS straight from NCDHW memory
// (rows along x are contiguous -> coalesced), zero padding is materialised in
// the LDS halo, and the taps become compile-time LDS offsets.
//
// Two families (the library picks per layer, conv_fwd_dispatch / wgrad_plan):
//   * Winograd F(2,3) along z -- conv3d_k3_fwd_wz_kernel (forward / backward-data) and
//     conv3d_k3_wgrad_wz_kernel (backward-weights, the transposed algorithm): a pair of
//     output planes costs 4 instead of 6 products per (ci, ky, kx) column, i.e. 2/3 of
//     the MFMAs, with exact fp32 arithmetic.  Default for Cin >= 8 / W % 4 == 0.
//   * direct 27-tap kernels -- conv3d_k3_fwd_kernel, conv3d_k3_wgrad(_vec)_kernel,
//     conv3d_k3_wgrad_c1_kernel: the first layer (Cin = 1), widths the Winograd wgrad
//     does not cover, and the A/B baseline (DRAM_CONV_DIRECT=1).
//
//   forward / backward-data  D[co][voxel] += W[co][ci] * X[ci][voxel+tap]
//       block = 256 voxels x 32*COT output channels, 4 waves of 32x32 accumulator
//       tiles, K loop over channel chunks of 4 with all taps unrolled.
//       The voxel index sits on the MFMA column (lane) axis so that every
//       accumulator register is a 128-byte run along x of one output channel.
//   backward-weights         dW[co][ci][tap] += dY[co][voxel] * X[ci][voxel+tap]
//       one 512-thread block per CU, each wave a 16x16 (co,ci) tile for all taps
//       (27 / 36 independent accumulators), K loop over boxes of voxels, partial
//       slabs + ordered reduce (deterministic, no float atomics).
//
// The input of forward and the output of backward-data may be a *virtual*
// channel concatenation of two tensors (crop_concat_5d fused away).
#include "common.h"
#include "lane_reduce.h"
#include <type_traits>
#include <stdlib.h>
#include <atomic>

namespace dram {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// channels [0,C1) live in p1 (spatial D,H,W); channels [C1,C1+C2) in p2 (spatial
// D2,H2,W2) seen through a crop window starting at (oz,oy,ox).
struct CatView {
    float* p1;
    float* p2;
    int C1, C2;
    int D2, H2, W2;
    int oz, oy, ox;
};

struct ConvArgs {
    CatView src;
    CatView dst;
    const float* wt;    // [27][Cin][Cout]
    const float* bias;  // [Cout] or null
    int N, Cin, Cout, D, H, W;
    int nbx, nby, nbz, co_tiles;
    // "normalise + ReLU on load": source tensor k is the RAW output y of the producing conv and the operand of
    // this conv is act_k(coefk[row][0] * y + coefk[row][1]) per (n, c) row, ReLU if reluk (coefk null: the tensor
    // is used as it is).  The activated tensor of the norm -> ReLU between two convs is then never written.
    const float* coef1;
    const float* coef2;
    int relu1, relu2;
    // BatchNorm / GroupNorm statistics of the OUTPUT in the epilogue: per (row, box, wave) {mean, M2, count} of the
    // wave's 64 outputs of that channel -> stats[(row * nparts + part) * 3]; null: not wanted.
    float* stats;
    int nparts;
    // division by co_tiles / nbx / nby / nbz as a multiply + shift (the persistent (z,y) kernel decodes three item cursors per
    // item: 17 runtime integer divisions, each a v_rcp_iflag sequence with a VALU -> SALU round trip, ~2,000 cycles per item).
    // One 48-byte record {divisor, multiplier, shift} x 4, so that a decode reads it with three wide scalar loads.
    alignas(16) unsigned dv_d[4];
    alignas(16) unsigned dv_m[4];
    alignas(16) unsigned dv_s[4];
};

// x / d for x < 2^31 by a host-prepared multiply + shift (Granlund-Montgomery, branch-free): l = ceil(log2 d),
// m = floor(2^32 (2^l - d) / d) + 1, x / d = (mulhi(x, m) + x) >> l (d == 1: l = 0, m = 1: mulhi = 0).
static inline void fast_div_prepare(unsigned d, unsigned& m, unsigned& sh) {
    if (d == 0) d = 1;
    unsigned l = 0;
    while ((1ull << l) < d) ++l;
    m = (unsigned)(((((unsigned long long)1 << l) - d) << 32) / d + 1);
    sh = l;
}
__device__ __forceinline__ unsigned fast_div(unsigned x, unsigned m, unsigned sh) {
    return (__umulhi(x, m) + x) >> sh;
}

constexpr int KC = 4;  // input channels per LDS stage

template <int BX, int BY, int BZ, int COT>
struct FwdGeom {
    static constexpr int HX = BX + 2, HY = BY + 2, HZ = BZ + 2;
    static constexpr int HV = HX * HY * HZ;
    static constexpr int PS = HV;
    static constexpr int COB = 32 * COT;
    static constexpr int NQ = (HV + 255) / 256;
    static constexpr int WROWS = 27 * KC;
    static constexpr int WSLOTS = WROWS * COB / 4;             // 16-byte slots of the weight tile
    static constexpr int WPASS = (WSLOTS + 255) / 256;          // dwordx4 loads per thread and chunk
    static constexpr int STAGE = KC * PS + WROWS * COB;  // floats per LDS stage
    static constexpr size_t LDS_BYTES = 2 * (size_t)STAGE * sizeof(float);
};

// Buffer (SRD) loads: 32-bit per-lane byte offset against a wave-uniform descriptor.  Offsets at or
// beyond num_records return 0, so zero padding (volume border, channel tails) needs neither a
// branch nor a select -- and a branch around a load would make hipcc wait vmcnt(0) per element.
constexpr unsigned OOB = 0x80000000u;   // > any plane size in bytes (check_conv_shape)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float buf_load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
    return __builtin_bit_cast(f32x4, v);
}
__device__ __forceinline__ const float* uniform_ptr(const float* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const float*)(((unsigned long long)hi << 32) | lo);
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// Workgroups are dealt round-robin over the 8 XCDs (each with a private 4 MiB L2): hardware block b
// runs on XCD b % 8.  Map it to a logical work item so that every XCD walks a contiguous range of
// items: neighbouring boxes (shared halos) and the tiles that share a box then hit the same L2.
// Bijective for any n; placement is a speed matter only.
__device__ __forceinline__ int xcd_remap(int b, int n) {
    const int q = n / 8, r = n % 8;
    const int xcd = b % 8, idx = b / 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// Statistics of a conv output tile in the epilogue (BatchNorm / GroupNorm moments without a pass over the tensor).
// A lane holds, for each of its NREG = 16*COT accumulator registers i (channel ch(i) below) two outputs Y(0,i),
// Y(1,i) (valid if ok0 / ok1).  Per wave half and channel: two-pass {mean, M2} over the <= 64 valid values
// (mean first, then squared deviations from it: no E[x^2] - E[x]^2 cancellation), written with the count to
// part[(row * nparts + pidx) * 3].  The partials are combined with Chan's formula in fp64 by norm_finalize_parts.
template <int COT, typename YF>
__device__ __forceinline__ void stats_epilogue(YF Y, bool ok0, bool ok1, int lane, float* __restrict__ stats,
                                               int64_t row0, int co0, int Cout, int nparts, int pidx) {
    constexpr int NREG = 16 * COT;
    const int kh = lane >> 5;
    const unsigned long long half = kh ? 0xffffffff00000000ull : 0x00000000ffffffffull;
    const float nvalid = (float)(__popcll(__ballot(ok0) & half) + __popcll(__ballot(ok1) & half));
    // interior boxes (every lane's two outputs inside the volume: wave-uniform): no selects, packed fp32 arithmetic
    const bool all_valid = NREG == 16 && __ballot(ok0 && ok1) == ~0ull;     // (32 registers: the second code path costs the z-only
                                                                            //  kernels 27 spilled registers and ~1 %)
    float s[NREG];
    if (all_valid) {
#pragma unroll
        for (int i = 0; i < NREG; i += 2) {
            const f32x2 y0 = {Y(0, i), Y(0, i + 1)}, y1 = {Y(1, i), Y(1, i + 1)};
            const f32x2 t = y0 + y1;
            s[i] = t[0];
            s[i + 1] = t[1];
        }
    } else {
#pragma unroll
        for (int i = 0; i < NREG; ++i) s[i] = (ok0 ? Y(0, i) : 0.f) + (ok1 ? Y(1, i) : 0.f);
    }
    lane_transpose_reduce<NREG>(s, lane);
    const float mean_j = nvalid > 0.f ? s[0] / nvalid : 0.f;
    __builtin_amdgcn_sched_barrier(0);     // keep the two phases apart: the sums are dead before the deviations are born
    float q[NREG];
    q[0] = mean_j;
    lane_transpose_broadcast<NREG>(q, lane);                      // q[i] = mean of register i's channel
    if (all_valid) {
#pragma unroll
        for (int i = 0; i < NREG; i += 2) {
            const f32x2 m = {q[i], q[i + 1]};
            const f32x2 y0 = {Y(0, i), Y(0, i + 1)}, y1 = {Y(1, i), Y(1, i + 1)};
            const f32x2 d0 = y0 - m, d1 = y1 - m;
            const f32x2 t = d0 * d0 + d1 * d1;
            q[i] = t[0];
            q[i + 1] = t[1];
        }
    } else {
#pragma unroll
        for (int i = 0; i < NREG; ++i) {
            const float d0 = Y(0, i) - q[i], d1 = Y(1, i) - q[i];
            q[i] = (ok0 ? d0 * d0 : 0.f) + (ok1 ? d1 * d1 : 0.f);
        }
    }
    lane_transpose_reduce<NREG>(q, lane);
    const int i = lane_register_index<NREG>(lane);
    const int co = co0 + 32 * (i >> 4) + (i & 3) + 8 * ((i & 15) >> 2) + 4 * kh;   // channel of register i (MFMA 32x32 layout)
    if (lane_writes_total<NREG>(lane) && co < Cout) {
        float* o = stats + ((row0 + co) * (int64_t)nparts + pidx) * 3;
        o[0] = mean_j;
        o[1] = q[0];
        o[2] = nvalid;
    }
}

// The same for NT 32-voxel tiles of ONE 32-channel tile per lane (16 registers each; tile t valid if OK(t)): per wave half and
// channel the two-pass {mean, M2} over the <= 32 * NT valid values.
template <int NT, typename YF, typename OKF>
__device__ __forceinline__ void stats_epilogue_tiles(YF Y, OKF OK, int lane, float* __restrict__ stats, int64_t row0, int co0, int Cout,
                                                     int nparts, int pidx) {
    const int kh = lane >> 5;
    const unsigned long long half = kh ? 0xffffffff00000000ull : 0x00000000ffffffffull;
    int nv = 0;
#pragma unroll
    for (int t = 0; t < NT; ++t) nv += __popcll(__ballot(OK(t)) & half);
    const float nvalid = (float)nv;
    float s[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        s[i] = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) s[i] += OK(t) ? Y(t, i) : 0.f;
    }
    lane_transpose_reduce<16>(s, lane);
    const float mean_j = nvalid > 0.f ? s[0] / nvalid : 0.f;
    __builtin_amdgcn_sched_barrier(0);
    float q[16];
    q[0] = mean_j;
    lane_transpose_broadcast<16>(q, lane);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float m = q[i];
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float d = Y(t, i) - m;
            acc += OK(t) ? d * d : 0.f;
        }
        q[i] = acc;
    }
    lane_transpose_reduce<16>(q, lane);
    const int i = lane_register_index<16>(lane);
    const int co = co0 + (i & 3) + 8 * (i >> 2) + 4 * kh;
    if (lane_writes_total<16>(lane) && co < Cout) {
        float* o = stats + ((row0 + co) * (int64_t)nparts + pidx) * 3;
        o[0] = mean_j;
        o[1] = q[0];
        o[2] = nvalid;
    }
}

template <typename YF>
__device__ __forceinline__ void stats_epilogue_tiles4(YF Y, bool ok, int lane, float* __restrict__ stats, int64_t row0, int co0, int Cout,
                                                      int nparts, int pidx) {
    stats_epilogue_tiles<4>(Y, [&](int) { return ok; }, lane, stats, row0, co0, Cout, nparts, pidx);
}

// Operand transform of a lazily normalised source (ConvArgs::coef1/2): per K-chunk channel the wave-uniform
// {a, b, lo}: v -> max(a*v + b, lo), lo = 0 with ReLU and -inf without; identity {1, 0, -inf} for a plain source,
// {0, 0, 0} for the channel tail beyond Cin.
struct LazyCoef {
    float a, b, lo;
};
__device__ __forceinline__ LazyCoef lazy_coef(const ConvArgs& a, int n, int ci) {
    LazyCoef c;
    if (ci >= a.Cin) { c.a = 0.f; c.b = 0.f; c.lo = 0.f; return c; }
    const bool first = ci < a.src.C1;
    const float* cf = first ? a.coef1 : a.coef2;
    const int relu = first ? a.relu1 : a.relu2;
    if (cf == nullptr) { c.a = 1.f; c.b = 0.f; c.lo = -INFINITY; return c; }
    const int64_t row = first ? (int64_t)n * a.src.C1 + ci : (int64_t)n * a.src.C2 + (ci - a.src.C1);
    c.a = cf[2 * row];
    c.b = cf[2 * row + 1];
    c.lo = relu ? 0.f : -INFINITY;
    return c;
}

// Software pipeline (one barrier per K chunk): while the MFMAs of chunk c run out of LDS stage
// c&1, the global loads of chunk c+1 are in flight into registers; they are written to the other
// stage after the MFMAs and become visible at the barrier.  Inside a chunk the LDS operand reads
// of k-step s+1 are issued ahead of the MFMAs of k-step s (two operand register sets).  Two blocks
// (2 x 80 KB of LDS, 2 waves per SIMD) share a CU, so one block's write/barrier phase is covered
// by the other's MFMAs.
template <int BX, int BY, int BZ, int COT, bool FUSED = false>
__global__ __launch_bounds__(256, 2) void conv3d_k3_fwd_kernel(ConvArgs a) {
    static_assert(BX * BY * BZ == 256, "block covers 256 voxels");
    using G = FwdGeom<BX, BY, BZ, COT>;
    constexpr int HX = G::HX, HY = G::HY, HV = G::HV, PS = G::PS, COB = G::COB, NQ = G::NQ;
    constexpr int WSLOTS = G::WSLOTS, WPASS = G::WPASS, STAGE = G::STAGE;
    static_assert(NQ <= 4, "halo elements per thread");

    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int b = xcd_remap(blockIdx.x, gridDim.x);       // logical item = (box, co tile), co tile fastest
    const int co0 = (b % a.co_tiles) * COB; b /= a.co_tiles;
    const int bx = b % a.nbx; b /= a.nbx;
    const int by = b % a.nby; b /= a.nby;
    const int bz = b % a.nbz;
    const int n = b / a.nbz;
    const int x0 = bx * BX, y0 = by * BY, z0 = bz * BZ;
    const int D = a.D, H = a.H, W = a.W;
    const int S = D * H * W;
    const int S2 = a.src.D2 * a.src.H2 * a.src.W2;

    // per-thread halo elements of the input stage: byte offsets inside one channel plane
    u32x4 off1, off2;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = tid + 256 * q;
        const int hx = e % HX, hy = (e / HX) % HY, hz = e / (HX * HY);
        const int gx = x0 - 1 + hx, gy = y0 - 1 + hy, gz = z0 - 1 + hz;
        const bool ok = (q < NQ) && (e < HV) && gx >= 0 && gx < W && gy >= 0 && gy < H && gz >= 0 && gz < D;
        off1[q] = ok ? 4u * (unsigned)((gz * H + gy) * W + gx) : OOB;
        off2[q] = ok ? 4u * (unsigned)(((gz + a.src.oz) * a.src.H2 + gy + a.src.oy) * a.src.W2 + gx + a.src.ox) : OOB;
    }
    const float* s1 = a.src.p1 + (size_t)n * a.src.C1 * S;
    const float* s2 = a.src.p2 ? a.src.p2 + (size_t)n * a.src.C2 * S2 : s1;

    f32x16 acc[COT][2];
#pragma unroll
    for (int c = 0; c < COT; ++c)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][t][r] = 0.f;

    const int j = lane & 31, kh = lane >> 5;
    int bbase[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int v = 32 * (2 * wave + t) + j;
        const int vx = v % BX, vy = (v / BX) % BY, vz = v / (BX * BY);
        bbase[t] = kh * PS + (vz * HY + vy) * HX + vx;
    }
    const int abase = kh * COB + j;

    // weights: the [27*KC rows][COB] tile of a chunk is staged with 16-byte loads (a row = COB contiguous
    // floats of wt[tap][ci][co0..]): slot f = p*256 + tid -> row f/(COB/4), columns 4*(f%(COB/4))..+3.
    // Columns of channels >= Cout hold whatever follows in memory (finite weights, or 0 beyond the
    // buffer): they only feed accumulator rows that the epilogue never stores.
    const __amdgpu_buffer_rsrc_t wsrd = make_rsrc(a.wt, 27u * (unsigned)a.Cin * (unsigned)a.Cout * 4u);
    const unsigned tap_stride = 4u * (unsigned)a.Cin * (unsigned)a.Cout;
    unsigned wvoff[WPASS];     // byte offset of this thread's slot p for chunk 0 (OOB for slots past the tile)
    int wkc[WPASS];
#pragma unroll
    for (int p = 0; p < WPASS; ++p) {
        const int f = p * 256 + tid;
        const int row = f / (COB / 4), c4 = f % (COB / 4);
        const int tap = row / KC;
        wkc[p] = row % KC;
        wvoff[p] = f < WSLOTS ? (unsigned)tap * tap_stride + 4u * (unsigned)(wkc[p] * a.Cout + co0 + 4 * c4) : OOB;
    }

    float rin[KC][NQ];   // prefetched input halo elements of the next chunk
    f32x4 rw[WPASS];     // prefetched weights of the next chunk
    const bool lazy = FUSED && (a.coef1 != nullptr || a.coef2 != nullptr);   // normalise + ReLU on load (block-uniform)
    LazyCoef lc[KC];     // the chunk's operand transforms (wave-uniform: scalar registers)

    auto load_chunk = [&](int c0) {
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            const int ci = c0 + kc;                      // wave-uniform
            const bool first = ci < a.src.C1;
            const float* plane = first ? s1 + (size_t)ci * S : s2 + (size_t)(ci - a.src.C1) * S2;
            const unsigned bytes = ci < a.Cin ? 4u * (unsigned)(first ? S : S2) : 0u;   // 0 records: all zeros
            const __amdgpu_buffer_rsrc_t srd = make_rsrc(uniform_ptr(plane), bytes);
            const u32x4 off = first ? off1 : off2;
#pragma unroll
            for (int q = 0; q < NQ; ++q) rin[kc][q] = buf_load(srd, off[q], 0);
            if (lazy) lc[kc] = lazy_coef(a, n, ci);
        }
        const unsigned cbase = 4u * (unsigned)c0 * (unsigned)a.Cout;
        if (c0 + KC <= a.Cin) {
#pragma unroll
            for (int p = 0; p < WPASS; ++p) rw[p] = buf_load4(wsrd, wvoff[p] + cbase, 0);
        } else {   // last, partial chunk: rows of channels >= Cin must read as 0
#pragma unroll
            for (int p = 0; p < WPASS; ++p) rw[p] = buf_load4(wsrd, (c0 + wkc[p]) < a.Cin ? wvoff[p] + cbase : OOB, 0);
        }
    };
    auto store_chunk = [&](float* stage) {
        float* lin = stage;
        float* lw = stage + KC * PS;
        if (lazy) {   // the zero padding belongs to the ACTIVATED tensor: out-of-volume halo elements stay 0
#pragma unroll
            for (int kc = 0; kc < KC; ++kc)
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    rin[kc][q] = off1[q] != OOB ? fmaxf(fmaf(lc[kc].a, rin[kc][q], lc[kc].b), lc[kc].lo) : 0.f;
        }
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                if (tid + 256 * q < HV) lin[kc * PS + tid + 256 * q] = rin[kc][q];
#pragma unroll
        for (int p = 0; p < WPASS; ++p) {
            const int f = p * 256 + tid;
            if (WPASS * 256 == WSLOTS || f < WSLOTS) *reinterpret_cast<f32x4*>(lw + 4 * f) = rw[p];
        }
    };
    auto compute = [&](const float* stage) {
        const float* lin = stage;
        const float* lw = stage + KC * PS;
        constexpr int NS = 27 * (KC / 2);   // k-steps of this chunk
        float av[2][COT], bv[2][2];
#pragma unroll
        for (int s = 0; s <= NS; ++s) {
            if (s < NS) {      // operands of k-step s -> register set s&1
                const int tap = s / (KC / 2), kk = s % (KC / 2);
                const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
                const int toff = (dz * HY + dy) * HX + dx;
#pragma unroll
                for (int c = 0; c < COT; ++c) av[s & 1][c] = lw[abase + (tap * KC + 2 * kk) * COB + 32 * c];
#pragma unroll
                for (int t = 0; t < 2; ++t) bv[s & 1][t] = lin[bbase[t] + 2 * kk * PS + toff];
            }
            if (s > 0) {       // MFMAs of k-step s-1
#pragma unroll
                for (int c = 0; c < COT; ++c)
#pragma unroll
                    for (int t = 0; t < 2; ++t)
                        acc[c][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[(s - 1) & 1][c], bv[(s - 1) & 1][t],
                                                                          acc[c][t], 0, 0, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, COT + 2, 0);   // DS reads of step s ...
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * COT, 0);   // ... then MFMAs of step s-1
        }
    };

    load_chunk(0);
    store_chunk(lds);
    __syncthreads();
    int cur = 0;
    for (int c0 = 0; c0 < a.Cin; c0 += KC) {
        const bool has_next = (c0 + KC) < a.Cin;
        if (has_next) load_chunk(c0 + KC);          // global loads in flight during the MFMAs
        __builtin_amdgcn_sched_barrier(0);
        compute(lds + cur * STAGE);
        __builtin_amdgcn_sched_barrier(0);
        if (has_next) store_chunk(lds + (cur ^ 1) * STAGE);
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: accumulator register r of lane (j,kh) = channel (r&3)+8(r>>2)+4kh, voxel j ----
    const int dS2 = a.dst.D2 * a.dst.H2 * a.dst.W2;
    bool okt[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int v = 32 * (2 * wave + t) + j;
        const int vx = v % BX, vy = (v / BX) % BY, vz = v / (BX * BY);
        const int gx = x0 + vx, gy = y0 + vy, gz = z0 + vz;
        okt[t] = gx < W && gy < H && gz < D;
        const int sp1 = (gz * H + gy) * W + gx;
        const int sp2 = ((gz + a.dst.oz) * a.dst.H2 + gy + a.dst.oy) * a.dst.W2 + gx + a.dst.ox;
#pragma unroll
        for (int c = 0; c < COT; ++c) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + 32 * c + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (co < a.Cout) {
                    float val = acc[c][t][r];
                    if (a.bias) val += a.bias[co];          // (never together with statistics: checked by the host)
                    if (okt[t]) {
                        if (co < a.dst.C1)
                            a.dst.p1[((size_t)n * a.dst.C1 + co) * S + sp1] = val;
                        else
                            a.dst.p2[((size_t)n * a.dst.C2 + (co - a.dst.C1)) * dS2 + sp2] = val;
                    }
                }
            }
        }
    }
    if (FUSED && a.stats) {
        const int pidx = (((bz * a.nby) + by) * a.nbx + bx) * 4 + wave;
        stats_epilogue<COT>([&](int t, int i) { return acc[i >> 4][t][i & 15]; }, okt[0], okt[1], lane, a.stats,
                            (int64_t)n * a.Cout, co0, a.Cout, a.nparts, pidx);
    }
}

// ---------------------------------------------------------------------------------------------
// forward of the FIRST layer (Cin == 1, e.g. DC3D ds_modules.0 conv 1->32, reference parts.py:177 with in_ch_list[0] = 1):
// the one 3x3x3 conv of the network that is HBM-bound (13 FLOP/B, SURVEY F5: 4 B read + 4*Cout B written per voxel).  The
// generic direct kernel pads the single input channel to a 4-channel K chunk and is then bound by matrix issue on zeros
// (0.20 of the HBM roof).  Here the 27 taps are the GEMM's K dimension (as in conv3d_k3_wgrad_c1_kernel):
//     Y[co][v] = sum_tap W[co][tap] * X[v + tap]        M = co (32 per tile), N = voxels, K = 27 (+1 zero)
// on v_mfma_f32_32x32x2_f32: 14 MFMAs per 32 voxels x 32 channels (1.5 ms of matrix time for 64 x 128^3, under the ~3.5 ms
// the bytes need).  A = the filter, 14 registers per lane for the whole block; B = one LDS read per MFMA whose address is a
// loop-invariant per-lane register (halo position of the lane's voxel + its k-index's tap offset) plus an immediate per row.
// Block = 32 x 8 x 4 voxels (8 KB halo in LDS), wave = one z plane = 8 rows of 32 voxels; a row's 16 accumulator registers
// are stored as they complete (128-byte runs along x per channel: buffer stores, out-of-volume lanes fall out of the
// descriptor's range check) and stay in registers for the statistics epilogue of their group of four rows (two-pass moments of
// <= 128 outputs per channel -> one partial per (row, box, wave, group), like the other forward kernels write them).
struct FwdC1Geom {
    static constexpr int BX = 32, BY = 8, BZ = 4;
    static constexpr int HX = BX + 2, HY = BY + 2, HZ = BZ + 2;
    static constexpr int HV = HX * HY * HZ;     // 2040
    static constexpr int NQ = (HV + 255) / 256;
};

__global__ __launch_bounds__(256, 3) void conv3d_k3_fwd_c1_kernel(ConvArgs a) {
    using G = FwdC1Geom;
    constexpr int BY = G::BY, HX = G::HX, HY = G::HY, HV = G::HV, NQ = G::NQ;
    __shared__ float lx[HV + 8];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, kh = lane >> 5;
    int b = xcd_remap(blockIdx.x, gridDim.x);       // logical item = (box, co tile), co tile fastest
    const int co0 = (b % a.co_tiles) * 32; b /= a.co_tiles;
    const int bx = b % a.nbx; b /= a.nbx;
    const int by = b % a.nby; b /= a.nby;
    const int bz = b % a.nbz;
    const int n = b / a.nbz;
    const int x0 = bx * G::BX, y0 = by * G::BY, z0 = bz * G::BZ;
    const int D = a.D, H = a.H, W = a.W, Cout = a.Cout;
    const int S = D * H * W;

    // input halo -> LDS (zero padding by the descriptor's range check)
    {
        const __amdgpu_buffer_rsrc_t srx = make_rsrc(uniform_ptr(a.src.p1 + (size_t)n * S), 4u * (unsigned)S);
        float v[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int e = tid + 256 * q;
            const int hx = e % HX, hy = (e / HX) % HY, hz = e / (HX * HY);
            const int gx = x0 - 1 + hx, gy = y0 - 1 + hy, gz = z0 - 1 + hz;
            const int ok = (int)(e < HV) & (int)((unsigned)gx < (unsigned)W) & (int)((unsigned)gy < (unsigned)H) & (int)((unsigned)gz < (unsigned)D);
            v[q] = buf_load(srx, ok ? 4u * (unsigned)((gz * H + gy) * W + gx) : OOB, 0);
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            if (tid + 256 * q < HV) lx[tid + 256 * q] = v[q];
    }
    // A operand: W[co0 + j][tap = 2 s + kh] (tap 27: the zero that pads K to 28); B offsets of the lane's k index
    float wa[14];
    int boff[14];
    {
        const __amdgpu_buffer_rsrc_t wsrd = make_rsrc(a.wt, 27u * 4u * (unsigned)Cout);
#pragma unroll
        for (int s = 0; s < 14; ++s) {
            const int tap = 2 * s + kh;
            const int ok = (int)(tap < 27) & (int)(co0 + j < Cout);
            wa[s] = buf_load(wsrd, ok ? 4u * (unsigned)(tap * Cout + co0 + j) : OOB, 0);
            const int tp = tap < 27 ? tap : 0;
            boff[s] = ((tp / 9 + wave) * HY + (tp / 3) % 3) * HX + tp % 3 + j;
        }
    }
    __syncthreads();

    // destination: one descriptor over sample n's [Cout][S] block; lane offset = its voxel + its 4 kh channels
    const unsigned S4 = 4u * (unsigned)S;
    const __amdgpu_buffer_rsrc_t dsrd = make_rsrc(uniform_ptr(a.dst.p1 + (size_t)n * Cout * S), (unsigned)Cout * S4);
    const int gx = x0 + j, gz = z0 + wave;
    const bool okxz = gx < W && gz < D;
    const unsigned vbase = 4u * (unsigned)((gz * H + y0) * W + gx) + (unsigned)(co0 + 4 * kh) * S4;
    const bool has_bias = a.bias != nullptr;
    // a channel tile that reaches past Cout: the register's channel is part of the scalar offset, which the descriptor's
    // range check does not see -- those lanes get the out-of-range vector offset instead
    const bool full_tile = co0 + 32 <= Cout;
    unsigned cmask = 0xffffu;
    if (!full_tile) {
        cmask = 0u;
#pragma unroll
        for (int r = 0; r < 16; ++r) cmask |= (co0 + 4 * kh + (r & 3) + 8 * (r >> 2) < Cout ? 1u : 0u) << r;
    }

    // two groups of four rows: 64 accumulator registers live (three blocks per CU), one statistics partial per group
    constexpr int NT = BY / 2;
#pragma unroll
    for (int grp = 0; grp < 2; ++grp) {
        f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int row = grp * NT + t;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
            for (int s = 0; s < 14; ++s)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[s], lx[boff[s] + row * HX], acc[t], 0, 0, 0);
            if (has_bias) {
                const __amdgpu_buffer_rsrc_t bsrd = make_rsrc(a.bias, 4u * (unsigned)Cout);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] += buf_load(bsrd, 4u * (unsigned)(co0 + 4 * kh + (r & 3) + 8 * (r >> 2)), 0);
            }
            const unsigned voff = (okxz && (y0 + row) < H) ? vbase + 4u * (unsigned)(row * W) : OOB;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float val = acc[t][r];    // (a copy: __builtin_bit_cast applied to the vector ELEMENT reads element 0 under hipcc 7.2)
                const unsigned vo = full_tile ? voff : (((cmask >> r) & 1u) ? voff : OOB);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), dsrd, (int)vo,
                                                      (int)((unsigned)((r & 3) + 8 * (r >> 2)) * S4), 0);
            }
        }
        if (a.stats) {
            const int pidx = ((((bz * a.nby) + by) * a.nbx + bx) * 4 + wave) * 2 + grp;
            stats_epilogue_tiles<NT>([&](int t, int i) { return acc[t][i]; }, [&](int t) { return okxz && (y0 + grp * NT + t) < H; }, lane,
                                     a.stats, (int64_t)n * Cout, co0, Cout, a.nparts, pidx);
        }
    }
}

// The same for wide volumes (W % 4 == 0, W >= 96): a lane owns FOUR consecutive x of a row -- tile tt of a row holds x = 4 j + tt
// -- so that a channel's 128-wide row leaves as 32 lanes x 16 bytes = 512 contiguous bytes per store instruction (four times
// fewer store instructions; the 32-wide form writes 128-byte pieces).  Block = 128 x 4 x 4 voxels, wave = one z plane, a row =
// four tiles = 64 accumulator registers with its own statistics partial.  B operands are LDS reads at stride 4 (2-way bank
// conflicts: 14 reads against 14 x 64 cycles of MFMA per tile).
struct FwdC1WGeom {
    static constexpr int BX = 128, BY = 4, BZ = 4;
    static constexpr int HX = BX + 2, HY = BY + 2, HZ = BZ + 2;
    static constexpr int HV = HX * HY * HZ;     // 4680
    static constexpr int NQ = (HV + 255) / 256;
};

__global__ __launch_bounds__(256, 3) void conv3d_k3_fwd_c1w_kernel(ConvArgs a) {
    using G = FwdC1WGeom;
    constexpr int BY = G::BY, HX = G::HX, HY = G::HY, HV = G::HV, NQ = G::NQ;
    __shared__ float lx[HV + 8];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, kh = lane >> 5;
    int b = xcd_remap(blockIdx.x, gridDim.x);       // logical item = (box, co tile), co tile fastest
    const int co0 = (b % a.co_tiles) * 32; b /= a.co_tiles;
    const int bx = b % a.nbx; b /= a.nbx;
    const int by = b % a.nby; b /= a.nby;
    const int bz = b % a.nbz;
    const int n = b / a.nbz;
    const int x0 = bx * G::BX, y0 = by * G::BY, z0 = bz * G::BZ;
    const int D = a.D, H = a.H, W = a.W, Cout = a.Cout;
    const int S = D * H * W;

    {   // input halo -> LDS (zero padding by the descriptor's range check)
        const __amdgpu_buffer_rsrc_t srx = make_rsrc(uniform_ptr(a.src.p1 + (size_t)n * S), 4u * (unsigned)S);
#pragma unroll
        for (int q0 = 0; q0 < NQ; q0 += 5) {
            float v[5];
#pragma unroll
            for (int q = q0; q < q0 + 5 && q < NQ; ++q) {
                const int e = tid + 256 * q;
                const int hx = e % HX, hy = (e / HX) % HY, hz = e / (HX * HY);
                const int gx = x0 - 1 + hx, gy = y0 - 1 + hy, gz = z0 - 1 + hz;
                const int ok = (int)(e < HV) & (int)((unsigned)gx < (unsigned)W) & (int)((unsigned)gy < (unsigned)H) & (int)((unsigned)gz < (unsigned)D);
                v[q - q0] = buf_load(srx, ok ? 4u * (unsigned)((gz * H + gy) * W + gx) : OOB, 0);
            }
#pragma unroll
            for (int q = q0; q < q0 + 5 && q < NQ; ++q)
                if (tid + 256 * q < HV) lx[tid + 256 * q] = v[q - q0];
        }
    }
    float wa[14];
    int boff[14];
    {
        const __amdgpu_buffer_rsrc_t wsrd = make_rsrc(a.wt, 27u * 4u * (unsigned)Cout);
#pragma unroll
        for (int s = 0; s < 14; ++s) {
            const int tap = 2 * s + kh;
            const int ok = (int)(tap < 27) & (int)(co0 + j < Cout);
            wa[s] = buf_load(wsrd, ok ? 4u * (unsigned)(tap * Cout + co0 + j) : OOB, 0);
            const int tp = tap < 27 ? tap : 0;
            boff[s] = ((tp / 9 + wave) * HY + (tp / 3) % 3) * HX + tp % 3 + 4 * j;
        }
    }
    __syncthreads();

    const unsigned S4 = 4u * (unsigned)S;
    const __amdgpu_buffer_rsrc_t dsrd = make_rsrc(uniform_ptr(a.dst.p1 + (size_t)n * Cout * S), (unsigned)Cout * S4);
    const int gx = x0 + 4 * j, gz = z0 + wave;
    const bool okxz = gx < W && gz < D;             // (W % 4 == 0: a lane's four x are inside together)
    const unsigned vbase = 4u * (unsigned)((gz * H + y0) * W + gx) + (unsigned)(co0 + 4 * kh) * S4;
    const bool has_bias = a.bias != nullptr;
    const bool full_tile = co0 + 32 <= Cout;
    unsigned cmask = 0xffffu;
    if (!full_tile) {
        cmask = 0u;
#pragma unroll
        for (int r = 0; r < 16; ++r) cmask |= (co0 + 4 * kh + (r & 3) + 8 * (r >> 2) < Cout ? 1u : 0u) << r;
    }
#pragma unroll
    for (int row = 0; row < BY; ++row) {
        f32x16 acc[4];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tt][r] = 0.f;
#pragma unroll
            for (int s = 0; s < 14; ++s)
                acc[tt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[s], lx[boff[s] + row * HX + tt], acc[tt], 0, 0, 0);
        }
        if (has_bias) {
            const __amdgpu_buffer_rsrc_t bsrd = make_rsrc(a.bias, 4u * (unsigned)Cout);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float bb = buf_load(bsrd, 4u * (unsigned)(co0 + 4 * kh + (r & 3) + 8 * (r >> 2)), 0);
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) acc[tt][r] += bb;
            }
        }
        const unsigned voff = (okxz && (y0 + row) < H) ? vbase + 4u * (unsigned)(row * W) : OOB;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
            const float v0 = acc[0][r], v1 = acc[1][r], v2 = acc[2][r], v3 = acc[3][r];
            const u32x4s pk = {__builtin_bit_cast(unsigned, v0), __builtin_bit_cast(unsigned, v1), __builtin_bit_cast(unsigned, v2),
                               __builtin_bit_cast(unsigned, v3)};
            const unsigned vo = full_tile ? voff : (((cmask >> r) & 1u) ? voff : OOB);
            __builtin_amdgcn_raw_buffer_store_b128(pk, dsrd, (int)vo, (int)((unsigned)((r & 3) + 8 * (r >> 2)) * S4), 0);
            // A > 64-bit buffer store still reads its data registers in the cycles after issue.  hipcc 7.2 only separates a
            // following write of those registers when soffset is an immediate; with an SGPR soffset it emitted
            // "buffer_store_dwordx4 v[104:107] ... s58 ; v_mov_b32 v104, ..." back to back and, measured on gfx950, ~0.7 % of
            // the first dwords at 128^3 then carried the NEXT channel's value (timing dependent).  Two wait states, pinned.
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 1");
            __builtin_amdgcn_sched_barrier(0);
        }
        if (a.stats) {
            const int pidx = ((((bz * a.nby) + by) * a.nbx + bx) * 4 + wave) * BY + row;
            stats_epilogue_tiles4([&](int t, int i) { return acc[t][i]; }, okxz && (y0 + row) < H, lane, a.stats, (int64_t)n * Cout, co0,
                                  Cout, a.nparts, pidx);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// forward / backward-data with Winograd F(2,3) along z (exact fp32 arithmetic, 2/3 of the MFMAs).
//
// For the two output planes z0, z0+1 of a box and its four input planes d0..d3 = x[z0-1..z0+2]
//     y[z0]   = g0 d0 + g1 d1 + g2 d2
//     y[z0+1] = g0 d1 + g1 d2 + g2 d3                        (g = the three z taps of one (ky,kx) column)
// are computed as  m0 = (d0-d2) g0,  m1 = (d1+d2)(g0+g1+g2)/2,  m2 = (d2-d1)(g0-g1+g2)/2,  m3 = (d1-d3) g2,
//     y[z0] = m0+m1+m2,  y[z0+1] = m1-m2-m3:
// four products per (ci, ky, kx) instead of six.  In GEMM terms: four independent implicit GEMMs (one per
// transformed plane xi) with K = Cin x 9 in-plane taps, N = the (y,x) positions of the plane pair, whose
// accumulators are combined in registers in the epilogue.  Why z and not x: the transformed input tile
// (4 planes) is exactly as large as the raw halo (BZ + 2 = 4 planes) and the transform is element-wise
// between planes -- a thread loads its (y,x) position from the 4 planes, combines, stores; no shuffles, no
// change to the coalesced row loads, no change to the lane <-> voxel mapping of operands and stores.
// The transformed filters (36 = 9 x 4 matrices instead of 27) are prepared by pack_weights_wz_kernel.
// LDS: input tile double-buffered (2 x 13 KB), filter tile (36 x KC x 64 floats = 36 KB) single-buffered
// (two barriers per chunk) so that two blocks still share a CU.
template <int BX, int BY, int COT>
struct FwdWzGeom {
    static constexpr int HX = BX + 2, HY = BY + 2;
    static constexpr int HP = HX * HY;                  // halo positions of one plane
    static constexpr int PK = 4 * HP;                   // floats per input channel: 4 transformed planes
    static constexpr int COB = 32 * COT;
    static constexpr int WROWS = 36 * KC;
    static constexpr int WSLOTS = WROWS * COB / 4;
    static constexpr int WPASS = (WSLOTS + 255) / 256;
    static constexpr int IN_STAGE = KC * PK;
    static constexpr size_t LDS_BYTES = (size_t)(2 * IN_STAGE + WROWS * COB) * sizeof(float);
};

template <int BX, int BY, int COT, bool FUSED = false>
__global__ __launch_bounds__(256, 2) void conv3d_k3_fwd_wz_kernel(ConvArgs a) {
    static_assert(BX * BY <= 128 && BX * BY > 96, "block covers (up to) 128 (y,x) positions of two z planes; lanes past BX*BY idle");
    using G = FwdWzGeom<BX, BY, COT>;
    constexpr int HX = G::HX, HP = G::HP, PK = G::PK, COB = G::COB;
    constexpr int WSLOTS = G::WSLOTS, WPASS = G::WPASS, IN_STAGE = G::IN_STAGE;
    static_assert(HP <= 256, "one halo position per thread");

    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* lwt = lds + 2 * IN_STAGE;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int b = xcd_remap(blockIdx.x, gridDim.x);       // logical item = (box, co tile), co tile fastest
    const int co0 = (b % a.co_tiles) * COB; b /= a.co_tiles;
    const int bx = b % a.nbx; b /= a.nbx;
    const int by = b % a.nby; b /= a.nby;
    const int bz = b % a.nbz;
    const int n = b / a.nbz;
    const int x0 = bx * BX, y0 = by * BY, z0 = bz * 2;
    const int D = a.D, H = a.H, W = a.W;
    const int S = D * H * W;
    const int S2 = a.src.D2 * a.src.H2 * a.src.W2;

    // staging role: thread = one (y,x) halo position, the 4 input planes z0-1 .. z0+2 of it
    u32x4 off1, off2;
    {
        const int hx = tid % HX, hy = tid / HX;
        const int gx = x0 - 1 + hx, gy = y0 - 1 + hy;
        const bool okp = tid < HP && gx >= 0 && gx < W && gy >= 0 && gy < H;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int gz = z0 - 1 + q;
            const bool ok = okp && gz >= 0 && gz < D;
            off1[q] = ok ? 4u * (unsigned)((gz * H + gy) * W + gx) : OOB;
            off2[q] = ok ? 4u * (unsigned)(((gz + a.src.oz) * a.src.H2 + gy + a.src.oy) * a.src.W2 + gx + a.src.ox) : OOB;
        }
    }
    const float* s1 = a.src.p1 + (size_t)n * a.src.C1 * S;
    const float* s2 = a.src.p2 ? a.src.p2 + (size_t)n * a.src.C2 * S2 : s1;

    f32x16 acc[COT][4];
#pragma unroll
    for (int c = 0; c < COT; ++c)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][t][r] = 0.f;

    const int j = lane & 31, kh = lane >> 5;
    const int pos = 32 * wave + j;                   // this lane's (y,x) position inside the box
    const int vx = pos % BX, vy = pos / BX;
    const int bbase = kh * PK + vy * HX + vx;
    const int abase = kh * COB + j;

    // transformed filters wz[(ky*3+kx)*4 + xi][Cin][Cout]: a chunk's [36*KC rows][COB] tile, 16-byte slots
    const __amdgpu_buffer_rsrc_t wsrd = make_rsrc(a.wt, 36u * (unsigned)a.Cin * (unsigned)a.Cout * 4u);
    const unsigned tap_stride = 4u * (unsigned)a.Cin * (unsigned)a.Cout;
    unsigned wvoff[WPASS];
    int wkc[WPASS];
#pragma unroll
    for (int p = 0; p < WPASS; ++p) {
        const int f = p * 256 + tid;
        const int row = f / (COB / 4), c4 = f % (COB / 4);
        const int tap = row / KC;
        wkc[p] = row % KC;
        wvoff[p] = f < WSLOTS ? (unsigned)tap * tap_stride + 4u * (unsigned)(wkc[p] * a.Cout + co0 + 4 * c4) : OOB;
    }

    float rin[KC][4];
    f32x4 rw[WPASS];
    const bool lazy = FUSED && (a.coef1 != nullptr || a.coef2 != nullptr);   // normalise + ReLU on load (block-uniform)
    LazyCoef lc[KC];     // the chunk's operand transforms (wave-uniform: scalar registers)

    auto load_chunk = [&](int c0) {
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            const int ci = c0 + kc;                      // wave-uniform
            const bool first = ci < a.src.C1;
            const float* plane = first ? s1 + (size_t)ci * S : s2 + (size_t)(ci - a.src.C1) * S2;
            const unsigned bytes = ci < a.Cin ? 4u * (unsigned)(first ? S : S2) : 0u;   // 0 records: all zeros
            const __amdgpu_buffer_rsrc_t srd = make_rsrc(uniform_ptr(plane), bytes);
            const u32x4 off = first ? off1 : off2;
#pragma unroll
            for (int q = 0; q < 4; ++q) rin[kc][q] = buf_load(srd, off[q], 0);
            if (lazy) lc[kc] = lazy_coef(a, n, ci);
        }
        const unsigned cbase = 4u * (unsigned)c0 * (unsigned)a.Cout;
        if (c0 + KC <= a.Cin) {
#pragma unroll
            for (int p = 0; p < WPASS; ++p) rw[p] = buf_load4(wsrd, wvoff[p] + cbase, 0);
        } else {
#pragma unroll
            for (int p = 0; p < WPASS; ++p) rw[p] = buf_load4(wsrd, (c0 + wkc[p]) < a.Cin ? wvoff[p] + cbase : OOB, 0);
        }
    };
    auto store_in = [&](float* lin) {     // B^T d of the four planes
        if (tid < HP) {
            if (lazy) {   // the zero padding belongs to the ACTIVATED tensor: out-of-volume halo elements stay 0
#pragma unroll
                for (int kc = 0; kc < KC; ++kc)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        rin[kc][q] = off1[q] != OOB ? fmaxf(fmaf(lc[kc].a, rin[kc][q], lc[kc].b), lc[kc].lo) : 0.f;
            }
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) {
                const float d0 = rin[kc][0], d1 = rin[kc][1], d2 = rin[kc][2], d3 = rin[kc][3];
                float* o = lin + kc * PK + tid;
                o[0] = d0 - d2;
                o[HP] = d1 + d2;
                o[2 * HP] = d2 - d1;
                o[3 * HP] = d1 - d3;
            }
        }
    };
    auto store_w = [&]() {
#pragma unroll
        for (int p = 0; p < WPASS; ++p) {
            const int f = p * 256 + tid;
            if (WPASS * 256 == WSLOTS || f < WSLOTS) *reinterpret_cast<f32x4*>(lwt + 4 * f) = rw[p];
        }
    };
    auto compute = [&](const float* lin) {
        constexpr int NS = 36 * (KC / 2);   // k-steps of this chunk: (ky,kx) x xi x channel pairs
        float av[2][COT], bv[2];
#pragma unroll
        for (int s = 0; s <= NS; ++s) {
            if (s < NS) {
                const int t36 = s / (KC / 2), kk = s % (KC / 2);
                const int tap9 = t36 / 4, xi = t36 % 4;
                const int toff = xi * HP + (tap9 / 3) * HX + tap9 % 3;
#pragma unroll
                for (int c = 0; c < COT; ++c) av[s & 1][c] = lwt[abase + (t36 * KC + 2 * kk) * COB + 32 * c];
                bv[s & 1] = lin[bbase + 2 * kk * PK + toff];
            }
            if (s > 0) {
                const int xi = ((s - 1) / (KC / 2)) % 4;
#pragma unroll
                for (int c = 0; c < COT; ++c)
                    acc[c][xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[(s - 1) & 1][c], bv[(s - 1) & 1], acc[c][xi], 0, 0, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, COT + 1, 0);   // DS reads of step s ...
            __builtin_amdgcn_sched_group_barrier(0x008, COT, 0);       // ... then MFMAs of step s-1
        }
    };

    load_chunk(0);
    store_in(lds);
    store_w();
    __syncthreads();
    int cur = 0;
    for (int c0 = 0; c0 < a.Cin; c0 += KC) {
        const bool has_next = (c0 + KC) < a.Cin;
        if (has_next) load_chunk(c0 + KC);
        __builtin_amdgcn_sched_barrier(0);
        compute(lds + cur * IN_STAGE);
        __builtin_amdgcn_sched_barrier(0);
        if (has_next) {
            store_in(lds + (cur ^ 1) * IN_STAGE);
            __syncthreads();            // every wave is done reading the filter tile of this chunk
            store_w();
            __syncthreads();
        }
        cur ^= 1;
    }

    // ---- epilogue: A^T m per accumulator element, then the same stores as the direct kernel ----
    const int dS2 = a.dst.D2 * a.dst.H2 * a.dst.W2;
    const int gx = x0 + vx, gy = y0 + vy;
    const bool lane_ok = pos < BX * BY && gx < W && gy < H;
    if (!FUSED || a.stats == nullptr) {
        if (lane_ok) {
#pragma unroll
            for (int zz = 0; zz < 2; ++zz) {
                const int gz = z0 + zz;
                if (gz >= D) continue;
                const int sp1 = (gz * H + gy) * W + gx;
                const int sp2 = ((gz + a.dst.oz) * a.dst.H2 + gy + a.dst.oy) * a.dst.W2 + gx + a.dst.ox;
#pragma unroll
                for (int c = 0; c < COT; ++c) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int co = co0 + 32 * c + (r & 3) + 8 * (r >> 2) + 4 * kh;
                        if (co < a.Cout) {
                            float val = zz == 0 ? (acc[c][0][r] + acc[c][1][r]) + acc[c][2][r]
                                                : (acc[c][1][r] - acc[c][2][r]) - acc[c][3][r];
                            if (a.bias) val += a.bias[co];
                            if (co < a.dst.C1)
                                a.dst.p1[((size_t)n * a.dst.C1 + co) * S + sp1] = val;
                            else
                                a.dst.p2[((size_t)n * a.dst.C2 + (co - a.dst.C1)) * dS2 + sp2] = val;
                        }
                    }
                }
            }
        }
        return;
    }
    // with statistics: the two output planes go to scalar registers first (same arithmetic, same stores; the four
    // accumulator tuples are dead from here on), feed the moments, and are stored last
    float yv[2][16 * COT];
#pragma unroll
    for (int c = 0; c < COT; ++c) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            yv[0][16 * c + r] = (acc[c][0][r] + acc[c][1][r]) + acc[c][2][r];      // (no bias on this path: a conv
            yv[1][16 * c + r] = (acc[c][1][r] - acc[c][2][r]) - acc[c][3][r];      //  followed by a norm has none)
        }
    }
    const bool ok0 = lane_ok && z0 < D, ok1 = lane_ok && (z0 + 1) < D;
    __builtin_amdgcn_sched_barrier(0);
    {
        const int pidx = (((bz * a.nby) + by) * a.nbx + bx) * 4 + wave;
        stats_epilogue<COT>([&](int t, int i) { return yv[t][i]; }, ok0, ok1, lane, a.stats, (int64_t)n * a.Cout, co0,
                            a.Cout, a.nparts, pidx);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int zz = 0; zz < 2; ++zz) {
        if (!(zz == 0 ? ok0 : ok1)) continue;
        const int gz = z0 + zz;
        const int sp1 = (gz * H + gy) * W + gx;
        const int sp2 = ((gz + a.dst.oz) * a.dst.H2 + gy + a.dst.oy) * a.dst.W2 + gx + a.dst.ox;
#pragma unroll
        for (int c = 0; c < COT; ++c) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + 32 * c + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (co < a.Cout) {
                    if (co < a.dst.C1)
                        a.dst.p1[((size_t)n * a.dst.C1 + co) * S + sp1] = yv[zz][16 * c + r];
                    else
                        a.dst.p2[((size_t)n * a.dst.C2 + (co - a.dst.C1)) * dS2 + sp2] = yv[zz][16 * c + r];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// forward / backward-data with Winograd F(2x2, 3x3) over (z, y) -- exact fp32 arithmetic, 4/9 of the direct MFMAs
// (2/3 of the z-only kernel's).
//
// A 2x2 (z,y) group of outputs and its 4x4 input patch d (per x and ci; the three x taps stay direct):
//     V = B^T d B,   U_kx = G g[:,:,kx] G^T,   M += U_kx (.) V(x + kx),   Y = A^T M A
// with the 1-D matrices of the z-only kernel applied along both axes (constants 0, +-1, 1/2; 1/4 in U).  In GEMM terms:
// 16 independent implicit GEMMs (one per transformed element xi = 4 xi_z + xi_y) with K = Cin x 3 x-taps, N = the
// positions (x, y pair) of a plane pair; 16 products per (ci, kx) instead of 36 for 4 outputs.
//
// The 16 accumulator tiles of a 32-position x 32-channel tile are 256 registers -- a whole register file half.  Two
// waves share such a tile instead: one holds xi_z = 0,1, the other xi_z = 2,3 (128 accumulator registers each, two
// waves per SIMD as in the other kernels); in the epilogue each applies the y part of A^T to its rows, the two swap
// one intermediate row pair through LDS and each finishes ONE output plane (so a wave again ends with 32 positions x
// 2 rows x 32 channels, the shape the statistics epilogue and the stores of the z-only kernel work on).
//
// Work item = 32 x positions x 2 y pairs x 1 z pair (256 voxels) x 64 output channels; block = 512 threads:
// wave = (y pair, 32-channel tile, xi_z half); ONE block per CU (133 KB of LDS: inputs AND filters double-buffered,
// one barrier per chunk of 4 input channels), PERSISTENT: block b of G walks the items b', b' + G, ...  What a second
// resident block covers in the other kernels is scheduled by hand here, as one software pipeline over the stream of
// (item, chunk) pairs that does not drain between items:
//   * the next chunk's global loads (the next item's first chunk at an item's end; issued right behind the barrier), its
//     input transform and LDS stores ride in small pieces between the MFMAs of a chunk; filters go global -> LDS directly;
//   * LDS operands arrive as ds_read_b64 (two k-steps per read: the channel-pair index kk is the fastest LDS axis) one
//     iteration (4 MFMAs) ahead; the last iteration's MFMAs are issued behind the chunk's barrier, after the first
//     operand reads of the next chunk, so that their LDS latency is covered by queued matrix work;
//   * an item's epilogue runs inside the first iteration of the next item's first chunk.
// Staging: wave w owns channel c0 + (w & 3) of the chunk; in waves 0-3 a lane = (y pair, x < 32) transforms one 4x4
// patch (16 loads, 32 adds, 16 stores); waves 4-7 do the same for the two extra halo columns (4 lanes).
// Filters: wzy[t = 16 kx + xi][chunk][kh][Cout][kk] (channel = 4 chunk + 2 kk + kh), zero-filled to whole chunks by
// pack_weights_wzy_kernel, so that a chunk's [48][2][64][2] tile is a lane-linear copy.
// Serves Cout % 64 == 0, Cin >= 8, volumes that 32x4x2 boxes cover well (fwd_choice); everything else: the z-only kernel.
// Two box shapes (template parameter BX), the same 256 voxels, 8 waves and LDS budget:
//   BX = 32: 32 x positions x 2 y pairs x 1 z pair: a wave's 32 MFMA columns = the 32 x positions of ONE y pair (128-byte rows);
//   BX = 16: 16 x positions x 2 y pairs x 2 z pairs: a wave's 32 columns = 16 x positions of the TWO y pairs of one z pair -- for
//            volumes that 32-wide boxes pad (the reference's own 80^3 chunks: the widths 80 / 40 / 20 of its pyramid are covered
//            by 16-wide boxes to 100 / 83 / 62 %, by 32-wide ones to 83 / 62 / 31 %; the 16^3 level of 128^3 chunks exactly).
//            Transformed planes are laid out [z pair][hx][y pair][kk], so that the 32 columns of a wave are again 64 consecutive
//            dwords (conflict-free ds_read_b64) and an x tap is a constant offset.  (16 x 8 rows x 1 z pair would need 10 raw
//            rows of 4 planes: 2 KB more than the CU's LDS; 6 rows of 6 planes is less than the 32-wide box's raw stage and
//            fetches 2.25 instead of 2.5 halo elements per output.)
template <int BX>
struct FwdWzyGeomT {
    static_assert(BX == 32 || BX == 16, "box widths of the (z,y) kernel");
    static constexpr int NZP = BX == 32 ? 1 : 2;        // z pairs of a box
    static constexpr int BY = 4, BZ = 2 * NZP;          // rows / planes of a box
    static constexpr int HX = BX + 2;
    static constexpr int KXS = BX == 32 ? 2 : 4;        // floats between two x positions of a transformed plane
    static constexpr int XI_STRIDE = 2 * NZP * HX * 2;  // floats per transformed plane: [y pair][hx][kk] / [z pair][hx][y pair][kk]
    static constexpr int KH_STRIDE = 16 * XI_STRIDE;    // per channel parity kh
    static constexpr int IN_STAGE = 2 * KH_STRIDE;      // 4352 floats
    static constexpr int WT_STRIDE = 2 * 64 * 2;        // per filter matrix t: [kh][co][kk]
    static constexpr int W_STAGE = 48 * WT_STRIDE;      // 12288 floats
    static constexpr int STAGE = IN_STAGE + W_STAGE;    // one stage: inputs, then filters
    static constexpr int WPASS = W_STAGE / 4 / 512;     // 16-byte slots per thread and chunk
    // raw input rows as they come from memory (LDS-DMA, 16 bytes per lane): [ci 4][z plane BZ + 2][row 6][BX + 8 floats = x0 - 4 ..
    // x0 + BX + 3] (240 / 216 16-byte pieces per channel: four LDS-DMA instructions, the last one of 48 / 24 lanes),
    // double-buffered (the chunk after next lands while the next one is transformed)
    static constexpr int RAW_PLANES = BZ + 2, RAW_ROW = BX + 8, RAW_PCS = RAW_ROW / 4;
    static constexpr int RAW_PLANE = 6 * RAW_ROW, RAW_CI = RAW_PLANES * RAW_PLANE, RAW_STAGE = 4 * RAW_CI;
    static constexpr int RAW_PIECES = RAW_CI / 4, RAW_PL_PIECES = RAW_PLANE / 4;
    static_assert(RAW_PIECES > 192 && RAW_PIECES <= 256, "four LDS-DMA instructions per channel");
    static constexpr size_t LDS_BYTES = (size_t)(2 * STAGE + 2 * RAW_STAGE) * sizeof(float);
    static_assert(LDS_BYTES <= 160 * 1024, "one block per CU: the whole LDS");
    // float index of position (z pair zp, y pair yp, halo column hx) inside a transformed plane
    __host__ __device__ static constexpr int pos(int zp, int yp, int hx) {
        return BX == 32 ? (yp * HX + hx) * 2 : (zp * HX + hx) * 4 + yp * 2;
    }
};
using FwdWzyGeom = FwdWzyGeomT<32>;

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// v_mov_b32 with a DPP lane pattern (quad_perm 0x00-0xFF, row_ror:n 0x120 + n, row_mirror 0x140), every lane enabled
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}

#ifdef DRAM_WZY_STAMPS      // diagnostics build only (scripts/diag_wzy_stamps.py): s_memtime shares of the chunk loop
__device__ unsigned long long g_wzy_stamps[32];      // [0 .. 15]: waves 0-3, [16 .. 31]: waves 4-7
#endif

// One instantiation serves plain and fused launches (lazy operands / statistics are runtime-uniform options): a separate
// plain instantiation measured 2-4 % SLOWER than this one run without the options (241 vs 251, 258 vs 269 TFLOP/s
// direct-equivalent at 64->64 / 192->64, 128^3), so it was dropped.
template <int BX>
__device__ __forceinline__ void fwd_wzy_body(const ConvArgs& a, int total_items) {
    using G = FwdWzyGeomT<BX>;
    constexpr int XI = G::XI_STRIDE, KHS = G::KH_STRIDE, IN_STAGE = G::IN_STAGE, KXS = G::KXS;
    constexpr int WTS = G::WT_STRIDE, STAGE = G::STAGE;
    static_assert(WTS == 4 * 64 && G::W_STAGE == 48 * WTS, "a filter matrix of a chunk = 1 KB = one LDS-DMA instruction of a wave");
    constexpr int SL0 = 7;         // first of the four iterations that carry the staging slices (8 measured the same)
    // Iterations in which the next chunk's loads are issued: the input patch right behind the barrier (it is wanted first, at
    // SL0), the filter tile from iteration 1 (wanted at the chunk's end; never in iteration 0, where an item's epilogue uses the
    // idle stage as its exchange buffer).  WHO issues them: the raw rows waves 4-7 (four LDS-DMA instructions, iterations 0-1),
    // the filter tile waves 0-3 (twelve, iterations 1-6) -- a memory instruction stalls the wave that issues it for hundreds of
    // cycles (scattered rows more than contiguous tiles), during which only its SIMD partner feeds the matrix pipe, and a
    // chunk ends when the slower wave group reaches the barrier.  Measured (stamps, cycles per chunk): with the filter tile
    // spread over all eight waves (six each, iterations 2-4) waves 0-3 waited 3,900 cycles per chunk at the barrier for waves
    // 4-7; with this split 160 against 900.  A/B of whole builds in one gpurun call: -1.0...-1.9 % ([4,64->64] / [4,192->64],
    // plain and + statistics), lazy -0.7 %; ten pieces + two on waves 4-7: the same; eight + four: +0.5 %; raw pieces moved to
    // waves 0-3 (one / two of the four): +1...+2 % / +3...+4 %; the tile one iteration later: no gain.
    constexpr int LD_IN = 0, LD_W = 1;
    constexpr int FPIECES = 12;    // filter pieces (one 1 KB matrix each) per filter-loading wave and chunk: iterations LD_W .. LD_W + 5
    static_assert(8 * 32 * 64 <= STAGE, "the epilogue exchange fits one (idle) stage");

    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // this block's items: r, r + G, r + 2G, ... (item = (box, 64-channel tile), channel tile fastest; G blocks).  In one
    // round the blocks of an XCD (hardware block b runs on XCD b % 8) work on G/8 consecutive items: neighbouring boxes
    // (shared halos), the tiles that share a box and the same filter chunks meet in that XCD's L2 at about the same time.
    const int item_step = (int)gridDim.x;
    const int item_lo = xcd_remap(blockIdx.x, gridDim.x), item_hi = total_items;
    if (item_lo >= item_hi) return;
    const int D = a.D, H = a.H, W = a.W;
    const int S = D * H * W;
    const int S2 = a.src.D2 * a.src.H2 * a.src.W2;
    const int Cin = a.Cin;
    // Scalar registers are scarce in the pipeline below (the compiler keeps every kernel argument it sees used in the
    // loop live, and spills the excess to VGPR lanes): the once-per-item code reads its arguments afresh from the
    // kernarg segment through a pointer the optimiser cannot see through.
    typedef const __attribute__((address_space(4))) ConvArgs* KArgs;
    auto kargs = [&]() {
        KArgs kp = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kp));
        return kp;
    };

    // ---- staging role: channel c0 + (wave & 3).  Waves 0-3: lane = (y pair, hx < 32), a whole 4x4 patch per lane (16 loads,
    // transforms in registers, 16 stores).  Waves 4-7 (the SIMD partners of 0-3) take the two extra halo columns hx = 32, 33:
    // four patches, spread over the wave as ONE element per lane -- lane = 16 * (2 * y pair + column) + 4 * row r + plane q --
    // with the transforms across lanes (DPP): ~20 instructions per chunk instead of the ~130 of the patch-per-lane form
    // (which they would spend on 4 active lanes), so that on every SIMD the patch stager runs beside a wave that does
    // little else than issue MFMAs. ----
    const int s_ci = wave & 3;
    const bool st_extra = wave >= 4;
    const int e_q = lane & 3, e_r = (lane >> 2) & 3;
    // (z pair, y pair, halo column) of the lane's patch (patch stagers: BX columns x 2 y pairs x NZP z pairs = 64 lanes) / of its
    // FIRST element (column stagers: four patches -- (y pair 0 / 1) x (column BX / BX + 1) -- per pass of 64 lanes; BX = 16 has
    // eight such patches: a second pass for z pair 1, ONE stage row (HX positions) further in LDS)
    const int i_ty = st_extra ? lane >> 5 : (BX == 32 ? lane >> 5 : (lane >> 4) & 1);
    const int i_tz = (st_extra || BX == 32) ? 0 : lane >> 5;
    const int i_hx = st_extra ? BX + ((lane >> 4) & 1) : lane & (BX - 1);
    constexpr int NEX = G::NZP;                    // elements per lane of a column stager
    constexpr int EX_STEP = BX == 32 ? 0 : G::pos(1, 0, 0) - G::pos(0, 0, 0);
    // LDS float index of the lane's xi = 0 element (patch stagers) / of its (first) element xi = (q, r) (column stagers)
    const int st_idx = (s_ci & 1) * KHS + G::pos(i_tz, i_ty, i_hx) + (s_ci >> 1) + (st_extra ? (4 * e_q + e_r) * XI : 0);
    // B^T along one axis, across the four lanes i = 0..3 of that axis: out_i = sa_i v_i + sb_i v_t(i), t = (2, 2, 1, 1)
    const float e_saz = e_q == 3 ? -1.f : 1.f, e_sbz = (e_q & 1) ? 1.f : -1.f;
    const float e_say = e_r == 3 ? -1.f : 1.f, e_sby = (e_r & 1) ? 1.f : -1.f;
    const bool e_mid = e_r == 1 || e_r == 2;
    // per staged item (set_staging_item):
    unsigned rowmask[4];           // all ones / zero: row inside the volume (column stagers: [0], [1] = their one or two elements)
    int sg_z0 = 0, sg_row1 = 0, sg_row2 = 0;
    // ---- fetch role (every wave): the raw rows of channel c0 + (wave >> 1), half (wave & 1) of its 240 16-byte pieces
    // [plane 4][row 6][x piece 10], by two LDS-DMA instructions (64 + 56 lanes).  Per fetched item (set_fetch_item):
    const int d_ci = wave & 3;
    unsigned dv1[4], dv2[4];       // byte offset of the lane's piece inside a channel of source 1 / 2 (OOB: outside the volume)
    const float* dg_base1 = nullptr;   // sample n of source 1 / 2
    const float* dg_base2 = nullptr;
    unsigned wvoff = 0;            // (waves 0-3) filter matrix `wave` of the item's channel tile, chunk 0; piece p is 4 matrices further
    const int nchunk = (Cin + 3) >> 2;
    const unsigned wchunk_bytes = 16u * (unsigned)a.Cout;
    const unsigned wpass_bytes = 4u * (unsigned)nchunk * wchunk_bytes;     // four filter matrices further
    const __amdgpu_buffer_rsrc_t wsrd = make_rsrc(a.wt, 48u * 16u * (unsigned)nchunk * (unsigned)a.Cout);
    const int C1 = a.src.C1;

    auto decode = [&](KArgs k, int item, int& n, int& x0, int& y0, int& z0, int& co0) {
        // (multiply + shift divisions: ConvArgs::dv_*; the record is read as three 16-byte scalar loads, issued together)
        typedef const __attribute__((address_space(4))) u32x4* K4;
        const u32x4 dd = *(K4)&k->dv_d[0], dm = *(K4)&k->dv_m[0], ds = *(K4)&k->dv_s[0];
        unsigned b = (unsigned)item, q;
        q = fast_div(b, dm[0], ds[0]); co0 = (int)(b - q * dd[0]) * 64; b = q;
        q = fast_div(b, dm[1], ds[1]); x0 = (int)(b - q * dd[1]) * BX; b = q;
        q = fast_div(b, dm[2], ds[2]); y0 = (int)(b - q * dd[2]) * 4; b = q;
        q = fast_div(b, dm[3], ds[3]); z0 = (int)(b - q * dd[3]) * G::BZ;
        n = (int)q;
    };
    // The whole pipeline is instantiated twice, for the patch stagers (waves 0-3) and the column stagers (waves 4-7), behind
    // ONE wave-uniform branch: with the role tested inside the loop the prefetch registers become phi nodes over the two
    // roles' paths, which hipcc resolves with register copies behind `s_waitcnt vmcnt(0)` right after the loads were issued.
    // Both instances execute the same barriers (same trip counts).
    auto run = [&](auto role) {
        constexpr bool EXTRA = decltype(role)::value;
        auto set_staging_item = [&](int item) {
            KArgs k = kargs();
            int n, x0, y0, z0, co0;
            decode(k, item, n, x0, y0, z0, co0);
            sg_z0 = z0;
            const int kC1 = k->src.C1, kC2 = k->src.C2;
            sg_row1 = n * kC1;
            sg_row2 = n * kC2 - kC1;
            const int gx = x0 - 1 + i_hx;
            if (EXTRA) {             // one element per lane (and pass): its validity (plane included)
    #pragma unroll
                for (int e = 0; e < NEX; ++e) {
                    const int gy = y0 - 1 + 2 * i_ty + e_r, gz = z0 - 1 + 2 * e + e_q;
                    const bool ok = (unsigned)gx < (unsigned)W && (unsigned)gy < (unsigned)H && (unsigned)gz < (unsigned)D;
                    rowmask[e] = ok ? 0xffffffffu : 0u;
                }
            } else {
    #pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int gy = y0 - 1 + 2 * i_ty + r;
                    const bool ok = gx >= 0 && gx < W && gy >= 0 && gy < H;
                    rowmask[r] = ok ? 0xffffffffu : 0u;
                }
            }
            const int t = (tid >> 6) & 3, r = tid & 63;      // (the filter DMA is issued by waves 0-3: matrices t, t + 4, ...)
            wvoff = 4u * (unsigned)(((t * nchunk) * 2 + (r >> 5)) * k->Cout * 2 + co0 * 2 + 4 * (r & 31));
        };
        auto set_fetch_item = [&](int item) {
            KArgs k = kargs();
            int n, x0, y0, z0, co0;
            decode(k, item, n, x0, y0, z0, co0);
            const int kC1 = k->src.C1, kC2 = k->src.C2;
            dg_base1 = k->src.p1 + (size_t)n * kC1 * S;
            dg_base2 = k->src.p2 ? k->src.p2 + (size_t)n * kC2 * S2 : dg_base1;
            const int oz = k->src.oz, oy = k->src.oy, ox = k->src.ox, H2 = k->src.H2, W2 = k->src.W2;
    #pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = 64 * i + lane;                        // (lanes >= 48 / 24 of the last instruction: masked off at issue)
                const int pl = c / G::RAW_PL_PIECES, row = (c % G::RAW_PL_PIECES) / G::RAW_PCS, xc = c % G::RAW_PCS;
                const int gz = z0 - 1 + pl, gy = y0 - 1 + row, gx = x0 - 4 + 4 * xc;    // W % 4 == 0: a piece is inside or outside as a whole
                const bool ok = (unsigned)gx < (unsigned)W && (unsigned)gy < (unsigned)H && (unsigned)gz < (unsigned)D;
                dv1[i] = ok ? 4u * (unsigned)((gz * H + gy) * W + gx) : OOB;
                dv2[i] = ok ? 4u * (unsigned)(((gz + oz) * H2 + gy + oy) * W2 + gx + ox) : OOB;
            }
        };

        // ---- compute role: wave = (y pair ty, channel tile ct, xi_z half xh); lane = (x position j, channel parity kh) ----
        // (xh, the xi_z half, IS the staging role: waves 4 .. 7 -- a compile-time constant of this instantiation, so that the
        //  epilogue's "which half am I" selects fold away: ~100 v_cndmask per item and wave as a runtime value)
        const int ty = wave & 1, ct = (wave >> 1) & 1;
        constexpr int xh = EXTRA ? 1 : 0;
        const int j = lane & 31, kh = lane >> 5;
        // column j of the wave's tile = x position j of y pair ty (BX = 32) / x position j % 16 of y pair j / 16 of z pair ty (BX = 16)
        const int bbase = kh * KHS + 8 * xh * XI + (BX == 32 ? G::pos(0, ty, j) : G::pos(ty, j >> 4, j & 15));
        const int abase = IN_STAGE + 8 * xh * WTS + (kh * 64 + 32 * ct + j) * 2;
        // (measured: reading the two operand pairs of an iteration as four ds_read_b64 -- separate opaque bases, so that hipcc
        //  cannot merge them into ds_read2(st64)_b64 -- is 2-4 % SLOWER, although the merged form has half the LDS rate)

#ifdef DRAM_WZY_STAMPS
        unsigned long long ep_acc[6] = {};      // epilogue: transform, exchange write + barrier, read + combine + barrier, statistics, stores
#define EP_STAMP(i_) { const unsigned long long tn_ = __builtin_readcyclecounter(); ep_acc[i_] += tn_ - ep_t; ep_t = tn_; }
#else
#define EP_STAMP(i_)
#endif
        f32x16 acc[8];
    #pragma unroll
        for (int t = 0; t < 8; ++t)
    #pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

        float rin[4][4];               // prefetched raw patch [z plane q][row r] of the chunk being staged
        f32x2 lc_ab = {1.f, 0.f};      // its channel's {a, b} (normalise on load) ...
        float lc_lo = 0.f;             // ... and ReLU floor (0 / -inf)
        bool lc_has = false;
        const bool lazy = a.coef1 != nullptr || a.coef2 != nullptr;
        // The coefficient pointers and ReLU flags of the two sources, parked in VECTOR registers: the scalar file is full in
        // this kernel, and the compiler would re-load them from the kernarg segment every chunk -- two dependent scalar
        // loads whose waits (the counter is shared with the LDS operand reads) sit in the middle of the MFMA stream.
        unsigned cfp[4] = {(unsigned)(unsigned long long)a.coef1, (unsigned)((unsigned long long)a.coef1 >> 32),
                           (unsigned)(unsigned long long)a.coef2, (unsigned)((unsigned long long)a.coef2 >> 32)};
        unsigned relu_bits = (a.relu1 ? 1u : 0u) | (a.relu2 ? 2u : 0u);
        asm volatile("" : "+v"(cfp[0]), "+v"(cfp[1]), "+v"(cfp[2]), "+v"(cfp[3]), "+v"(relu_bits));

        // The raw rows of chunk c0 of the fetched item -> raw buffer `rawbuf`: out-of-volume pieces arrive as zeros (range
        // check of the lane offset), a channel past Cin as zeros (empty descriptor).
        auto fetch_issue = [&](int c0, float* rawbuf, int i) {
            if (!EXTRA) return;
            const int ci = c0 + d_ci;                            // wave-uniform
            const bool first = ci < C1;
            const unsigned sx = (unsigned)(first ? S : S2);
            const float* base = (first ? dg_base1 : dg_base2) + (size_t)(unsigned)(first ? ci : ci - C1) * sx;
            const __amdgpu_buffer_rsrc_t srd = make_rsrc(uniform_ptr(base), ci < Cin ? 4u * sx : 0u);
            float* dst = rawbuf + d_ci * G::RAW_CI + 256 * i;
            if (i < 3 || lane < G::RAW_PIECES - 192)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lds_ptr_t)dst, 16, (int)(first ? dv1[i] : dv2[i]), 0, 0, 0);
        };
        // the patch of the lane (waves 0-3) / its one element (waves 4-7) out of the raw buffer
        auto read_raw = [&](const float* rawbuf) {
            if (EXTRA) {
    #pragma unroll
                for (int e = 0; e < NEX; ++e)
                    rin[0][e] = rawbuf[s_ci * G::RAW_CI + ((2 * e + e_q) * 6 + 2 * i_ty + e_r) * G::RAW_ROW + 3 + i_hx];
            } else {
                const float* pr = rawbuf + s_ci * G::RAW_CI + (2 * i_tz * 6 + 2 * i_ty) * G::RAW_ROW + 3 + i_hx;
    #pragma unroll
                for (int q = 0; q < 4; ++q)
    #pragma unroll
                    for (int r = 0; r < 4; ++r) rin[q][r] = pr[(q * 6 + r) * G::RAW_ROW];
            }
        };
        // the staged chunk's {a, b} of channel c0 + s_ci (normalise on load): one (wave-uniform address) vector load -- keeps
        // the scalar memory path, which shares its counter with the LDS operand reads, out of the loop
        auto load_coef = [&](int c0) {
            const int ci = c0 + s_ci;                            // wave-uniform
            const bool first = ci < C1;
            const unsigned plo = __builtin_amdgcn_readfirstlane(first ? cfp[0] : cfp[2]);
            const unsigned phi = __builtin_amdgcn_readfirstlane(first ? cfp[1] : cfp[3]);
            const float* cf = (const float*)(((unsigned long long)phi << 32) | plo);
            const unsigned rb = __builtin_amdgcn_readfirstlane(relu_bits);
            lc_has = cf != nullptr && ci < Cin;
            const unsigned row = (unsigned)((first ? sg_row1 : sg_row2) + ci);
            const __amdgpu_buffer_rsrc_t csrd = make_rsrc(cf, lc_has ? 0x7ffffff0u : 0u);
            const auto raw = __builtin_amdgcn_raw_buffer_load_b64(csrd, 0, (int)(8u * row), 0);
            lc_ab = __builtin_bit_cast(f32x2, raw);
            lc_lo = (lc_has && (rb & (first ? 1u : 2u))) ? 0.f : -INFINITY;
        };
        // the staged chunk's filter tile, in two halves: global -> LDS directly
        auto load_filters = [&](int c0, float* nstage, int p) {       // piece p of FPIECES: filter matrix wave + 4 p (waves 0-3)
            if (EXTRA) return;
            const unsigned cb = (unsigned)(c0 >> 2) * wchunk_bytes;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrd, (lds_ptr_t)(nstage + IN_STAGE + WTS * (wave + 4 * p)), 16,
                                                     (int)wvoff, (int)(cb + (unsigned)p * wpass_bytes), 0, 0);
        };
        // normalise + ReLU of the raw patch; the zero padding belongs to the ACTIVATED tensor (a plain source / the channel
        // tail: identity resp. zeros in, zeros out).  Branch-free per element: row validity as bit masks on b and lo.
        auto activate = [&](int half) {                      // z planes 2*half, 2*half + 1
            // The loaded {a, b} become visible to the optimiser HERE: left to itself it hoists the selects below to just
            // behind the load (iteration 2), and with them an `s_waitcnt vmcnt(0)` on every load of the chunk just issued --
            // a whole memory latency in the MFMA stream (+1700 cycles per chunk, with or without a lazy operand).
            if (half == 0) asm volatile("" : "+v"(lc_ab), "+v"(lc_lo));
            const float ca = lc_has ? lc_ab[0] : 1.f, cb = lc_has ? lc_ab[1] : 0.f;
            if (EXTRA) {             // the lane's one element (an element outside the volume was loaded as 0 and stays 0)
                if (half < NEX) {
                    const float b1 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, cb) & rowmask[half]);
                    const float l1 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, lc_lo) & rowmask[half]);
                    asm("v_fma_f32 %0, %1, %0, %2\n\tv_max_f32 %0, %0, %3" : "+v"(rin[0][half]) : "v"(ca), "v"(b1), "v"(l1));
                }
                return;
            }
            float br[4], lor[4];
    #pragma unroll
            for (int r = 0; r < 4; ++r) {
                br[r] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, cb) & rowmask[r]);
                lor[r] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, lc_lo) & rowmask[r]);
            }
            // exactly two instructions per element (as C++, fmaxf() on the masked floor costs a canonicalising v_max each,
            // and an if/else over the z validity was flattened into both arms plus moves: ~85 instructions per half)
    #pragma unroll
            for (int q = 2 * half; q < 2 * half + 2; ++q)
    #pragma unroll
                for (int r = 0; r < 4; ++r)
                    asm("v_fma_f32 %0, %1, %0, %2\n\tv_max_f32 %0, %0, %3" : "+v"(rin[q][r]) : "v"(ca), "v"(br[r]), "v"(lor[r]));
            if (!(sg_z0 >= 1 && sg_z0 + G::BZ < D)) {        // a plane outside the volume (first / last box along z only): zeros
    #pragma unroll
                for (int q = 2 * half; q < 2 * half + 2; ++q) {
                    const int gz = sg_z0 + 2 * i_tz - 1 + q;
                    if (!(gz >= 0 && gz < D)) {
    #pragma unroll
                        for (int r = 0; r < 4; ++r) rin[q][r] = 0.f;
                    }
                }
            }
        };
        auto transform_z = [&]() {                           // B^T d along z (in place)
            if (EXTRA) {             // across the quad's four lanes (planes): partner = quad_perm [2, 2, 1, 1]
    #pragma unroll
                for (int e = 0; e < NEX; ++e) {
                    const float t = dpp_mov<0x5A>(rin[0][e]);
                    rin[0][e] = fmaf(e_sbz, t, e_saz * rin[0][e]);
                }
                return;
            }
    #pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float d0 = rin[0][r], d1 = rin[1][r], d2 = rin[2][r], d3 = rin[3][r];
                rin[0][r] = d0 - d2; rin[1][r] = d1 + d2; rin[2][r] = d2 - d1; rin[3][r] = d1 - d3;
            }
        };
        auto transform_y_store = [&](float* o, int q) {      // (.) B along y, row xi_z = q -> LDS
            if (EXTRA) {             // across the four quads (rows) of a 16-lane DPP row; one store per lane
                if (q < NEX) {              // (BX = 16: the second pass rides one slice later)
                    const float v = rin[0][q];
                    const float t2 = dpp_mov<0x128>(v);                     // row_ror:8: the row two further (0 <-> 2, 1 <-> 3)
                    const float t1 = dpp_mov<0x1B>(dpp_mov<0x140>(v));      // row_mirror, quads reversed back: rows 1 <-> 2 (0 <-> 3)
                    const float t = e_mid ? t1 : t2;                        // partner row (2, 2, 1, 1)
                    o[q * EX_STEP] = fmaf(e_sby, t, e_say * v);
                }
                return;
            }
            const float d0 = rin[q][0], d1 = rin[q][1], d2 = rin[q][2], d3 = rin[q][3];
            o[(4 * q + 0) * XI] = d0 - d2;
            o[(4 * q + 1) * XI] = d1 + d2;
            o[(4 * q + 2) * XI] = d2 - d1;
            o[(4 * q + 3) * XI] = d1 - d3;
        };

        // ---- epilogue of one item: Y = A^T M A; `xch` = an idle LDS stage for the swap between the xi_z halves ----
        auto epilogue = [&](int item, float* xch) {
            KArgs k = kargs();
            int n, x0, y0, z0, co0;
            decode(k, item, n, x0, y0, z0, co0);
#ifdef DRAM_WZY_STAMPS
            unsigned long long ep_t = __builtin_readcyclecounter();
#endif
            // rows of A^T M (y part) for the wave's two xi_z: p[q][yy]
            float pq[2][2][16];
    #pragma unroll
            for (int q = 0; q < 2; ++q)
    #pragma unroll
                for (int r = 0; r < 16; ++r) {
                    pq[q][0][r] = (acc[4 * q][r] + acc[4 * q + 1][r]) + acc[4 * q + 2][r];
                    pq[q][1][r] = (acc[4 * q + 1][r] - acc[4 * q + 2][r]) - acc[4 * q + 3][r];
                }
    #pragma unroll
            for (int t = 0; t < 8; ++t)
    #pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
            // z part: Y[0] = (p0 + p1) + p2, Y[1] = (p1 - p2) - p3.  The xh = 0 wave (p0, p1) finishes plane 0 and needs p2;
            // the xh = 1 wave (p2, p3) finishes plane 1 and needs p1.  Slot layout [wave][register][lane]: conflict-free.
            EP_STAMP(0)
            float* mine = xch + wave * 32 * 64 + lane;
            const float* theirs = xch + (wave ^ 4) * 32 * 64 + lane;
    #pragma unroll
            for (int yy = 0; yy < 2; ++yy)
    #pragma unroll
                for (int r = 0; r < 16; ++r) mine[(16 * yy + r) * 64] = xh == 0 ? pq[1][yy][r] : pq[0][yy][r];
            __syncthreads();
            EP_STAMP(1)
            float yv[2][16];
    #pragma unroll
            for (int yy = 0; yy < 2; ++yy)
    #pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float got = theirs[(16 * yy + r) * 64];
                    yv[yy][r] = xh == 0 ? (pq[0][yy][r] + pq[1][yy][r]) + got      // (p0 + p1) + p2
                                        : (got - pq[0][yy][r]) - pq[1][yy][r];      // (p1 - p2) - p3
                }
            __syncthreads();           // every wave has read: the stage may be filled again
            EP_STAMP(2)
            const int gx = x0 + (BX == 32 ? j : (j & 15)), gz = z0 + (BX == 32 ? 0 : 2 * ty) + xh, gy = y0 + 2 * (BX == 32 ? ty : (j >> 4));
            const bool ok0 = gx < W && gz < D && gy < H;
            const bool ok1 = gx < W && gz < D && (gy + 1) < H;
            const int kCout = k->Cout;
            float* const kstats = k->stats;
            if (kstats) {
                __builtin_amdgcn_sched_barrier(0);
                // the box's number inside its sample, from the decoded coordinates (no division)
                const int box = (int)((((unsigned)z0 / G::BZ) * (unsigned)k->nby + ((unsigned)y0 >> 2)) * (unsigned)k->nbx + (unsigned)x0 / BX);
                stats_epilogue<1>([&](int t, int i) { return yv[t][i]; }, ok0, ok1, lane, kstats, (int64_t)n * kCout,
                                  co0 + 32 * ct, kCout, k->nparts, box * 4 + 2 * ty + xh);
                __builtin_amdgcn_sched_barrier(0);
            }
            EP_STAMP(3)
            // stores: the wave's 32 channels lie in ONE destination tensor (host: dst.C1 % 32 == 0) -> one descriptor over
            // them, lane offset = (its 4 kh channels, its voxel), the register's channel as a scalar offset
            const int cw = co0 + 32 * ct;
            const int dC1 = k->dst.C1, dC2 = k->dst.C2, dH2 = k->dst.H2, dW2 = k->dst.W2;
            const int dS2 = k->dst.D2 * dH2 * dW2;
            const bool t1 = cw < dC1;
            const float* dbase = t1 ? k->dst.p1 + ((size_t)n * dC1 + cw) * S
                                    : k->dst.p2 + ((size_t)n * dC2 + (cw - dC1)) * dS2;
            const unsigned cstride = 4u * (unsigned)(t1 ? S : dS2);
            const unsigned rowb = 4u * (unsigned)(t1 ? W : dW2);
            const unsigned sp = t1 ? (unsigned)((gz * H + gy) * W + gx)
                                   : (unsigned)(((gz + k->dst.oz) * dH2 + gy + k->dst.oy) * dW2 + gx + k->dst.ox);
            const __amdgpu_buffer_rsrc_t dsrd = make_rsrc(uniform_ptr(dbase), 32u * cstride);
            const unsigned v0 = 4u * (unsigned)kh * cstride + 4u * sp;
            const float* kbias = k->bias;
            if (kstats == nullptr && kbias != nullptr) {       // (a conv that feeds a norm has no bias: host-checked)
                float bv16[16];
                const __amdgpu_buffer_rsrc_t bsrd = make_rsrc(kbias, 4u * (unsigned)kCout);
    #pragma unroll
                for (int r = 0; r < 16; ++r) bv16[r] = buf_load(bsrd, 4u * (unsigned)(cw + 4 * kh), 4u * (unsigned)((r & 3) + 8 * (r >> 2)));
    #pragma unroll
                for (int r = 0; r < 16; ++r) { yv[0][r] += bv16[r]; yv[1][r] += bv16[r]; }
            }
    #pragma unroll
            for (int yy = 0; yy < 2; ++yy) {
                if (yy == 0 ? ok0 : ok1) {
    #pragma unroll
                    for (int r = 0; r < 16; ++r)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, yv[yy][r]), dsrd, (int)(v0 + yy * rowb),
                                                              (int)((unsigned)((r & 3) + 8 * (r >> 2)) * cstride), 0);
                }
            }
            EP_STAMP(4)
#ifdef DRAM_WZY_STAMPS
            ep_acc[5] += 1;
#endif
        };

        // ---- the pipeline.  Cursors: (c_item, c_c0) = the chunk being multiplied, (s_item, s_c0) = the chunk being staged
        // (raw rows -> transformed tile, filters), exactly one chunk ahead, (d_item, d_c0) = the chunk whose raw rows are being
        // fetched, two chunks ahead.  Raw buffer of a chunk = parity of its number in the stream = stage parity.
        float* const raw0 = lds + 2 * STAGE;
        int s_item = item_lo, s_c0 = 0;
        auto advance_staging = [&]() {
            s_c0 += 4;
            if (s_c0 >= Cin) {
                s_c0 = 0;
                s_item += item_step;
                if (s_item < item_hi) set_staging_item(s_item);
            }
        };
        int d_item = item_lo, d_c0 = 0;
        auto advance_fetch = [&]() {
            d_c0 += 4;
            if (d_c0 >= Cin) {
                d_c0 = 0;
                d_item += item_step;
                if (d_item < item_hi) set_fetch_item(d_item);
            }
        };
        // prologue: raw rows of the first two chunks, filters of the first; the first chunk transformed into stage 0
        set_staging_item(s_item);
        set_fetch_item(d_item);
    #pragma unroll
        for (int i = 0; i < 4; ++i) fetch_issue(0, raw0, i);
        advance_fetch();
        if (d_item < item_hi) {
    #pragma unroll
            for (int i = 0; i < 4; ++i) fetch_issue(d_c0, raw0 + G::RAW_STAGE, i);
            advance_fetch();
        }
        if (lazy) load_coef(0);
    #pragma unroll
        for (int p = 0; p < FPIECES; ++p) load_filters(0, lds, p);
        __syncthreads();                            // (waits for the wave's own loads first: the raw rows of all waves are in)
        read_raw(raw0);
        if (lazy) { activate(0); activate(1); }
        transform_z();
    #pragma unroll
        for (int q = 0; q < 4; ++q) transform_y_store(lds + st_idx, q);
        advance_staging();
        __syncthreads();

        int c_item = item_lo, c_c0 = 0, cur = 0;
        bool pending = false;                       // the previous chunk's last four MFMAs are still to be issued
        f32x2 av[2][2], bv[2][2];
    #ifdef DRAM_WZY_STAMPS
        unsigned long long st_acc[10] = {};
    #endif
        // The work that rides between the MFMAs of iteration `it`, in two pieces (after the 2nd / the 3rd MFMA), so that no
        // gap between two MFMAs of a wave is much longer than it has to be: the two waves of a SIMD run in step, and whatever
        // both do between two MFMAs is matrix-pipe idle time.  Staging slices run unconditionally: without a next chunk
        // they move stale data into the idle stage.
        auto ride = [&](int it, int half, bool has_next, bool has_fetch, float* nstage, int cur_) {
    #ifndef DRAM_WZY_DIAG_NOPATCH       // (diagnostic builds, scripts/diag_wzy_stamps.py: what a part of the staging costs)
            if (it < 2 && has_fetch) fetch_issue(d_c0, raw0 + cur_ * G::RAW_STAGE, 2 * it + half);
    #endif
            if (it == LD_IN && half == 1 && has_next && lazy) load_coef(s_c0);
    #ifndef DRAM_WZY_DIAG_NOFILT
            if (it >= LD_W && it < LD_W + FPIECES / 2 && has_next) load_filters(s_c0, nstage, 2 * (it - LD_W) + half);
    #endif
    #ifndef DRAM_WZY_DIAG_NOSTAGE
            if (it == SL0 - 1 && half == 1) read_raw(raw0 + (cur_ ^ 1) * G::RAW_STAGE);
            if (it == SL0 && lazy && lc_has) activate(half);       // (a plain channel of a launch with a lazy source: nothing to do)
            if (it == SL0 + 1 && half == 0) transform_z();
            const int q = 2 * (it - SL0 - 1) + half - 1;            // SL0+1: -, 0;  SL0+2: 1, 2;  SL0+3: 3, -
            if (q >= 0 && q < 4) transform_y_store(nstage + st_idx, q);
    #endif
        };
        for (;;) {
            const bool c_valid = c_item < item_hi;
            const bool has_next = s_item < item_hi;
            const bool has_fetch = d_item < item_hi;
            const bool boundary = pending && (c_c0 == 0);     // the previous chunk completed an item
            const float* stage = lds + cur * STAGE;
            float* nstage = lds + (cur ^ 1) * STAGE;
    #ifdef DRAM_WZY_STAMPS
            unsigned long long tp = __builtin_readcyclecounter();
    #endif
    #pragma unroll
            for (int it = 0; it < 12; ++it) {
                const int i0 = (it + 11) % 12;                // the MFMAs issued in this iteration belong to iteration i0
                const bool go = it > 0 || pending;
                if (it > 0 || c_valid) {
    #pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int u = 2 * it + h, kx = u >> 3, xi = u & 7;
                        av[it & 1][h] = *reinterpret_cast<const f32x2*>(stage + abase + (16 * kx + xi) * WTS);
                        bv[it & 1][h] = *reinterpret_cast<const f32x2*>(stage + bbase + xi * XI + KXS * kx);
                    }
                }
    #pragma unroll
                for (int m = 0; m < 4; ++m) {                 // k = m / 2, h = m % 2
                    if (go) {
                        const int h = m & 1, k = m >> 1, xi = (2 * i0 + h) & 7;
                        acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i0 & 1][h][k], bv[i0 & 1][h][k], acc[xi], 0, 0, 0);
                    }
                    if (m == 1 || m == 2) ride(it, m - 1, has_next, has_fetch, nstage, cur);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (it == 0) {
                    // (the MFMAs of the previous chunk's last iteration are queued behind the barrier: they cover the latency
                    //  of the operand reads above; the next chunk's filter loads follow in iteration LD_W)
                    if (boundary) epilogue(c_item - item_step, nstage);     // (the stage to fill is idle: its loads come after)
    #ifdef DRAM_WZY_STAMPS
                    if (!c_valid && lane == 0)
                        for (int i = 0; i < 10; ++i) atomicAdd(&g_wzy_stamps[(EXTRA ? 16 : 0) + i], st_acc[i]);
                    if (!c_valid && lane == 0)
                        for (int i = 0; i < 6; ++i) atomicAdd(&g_wzy_stamps[(EXTRA ? 16 : 0) + 10 + i], ep_acc[i]);
    #endif
                    if (!c_valid) return;
                    __builtin_amdgcn_sched_barrier(0);
                }
    #ifdef DRAM_WZY_STAMPS
                if (it == 0 || it == 2 || it == 6 || it == 11) {
                    const unsigned long long tn = __builtin_readcyclecounter();
                    st_acc[it == 0 ? (boundary ? 6 : 0) : it == 2 ? 1 : it == 6 ? 2 : 3] += tn - tp;
                    if (it == 0) st_acc[boundary ? 8 : 7] += 1;
                    tp = tn;
                }
    #endif
            }
            // The last iteration's four MFMAs are issued behind the barrier, after the next chunk's first operand reads, whose
            // latency they cover (measured, MFMAs before / behind the barrier: 0/4 is 1-2 % faster than 2/2, 4/0 1 % slower).
            pending = true;
            if (has_next) advance_staging();
            if (has_fetch) advance_fetch();
    #ifdef DRAM_WZY_STAMPS
            { const unsigned long long tb = __builtin_readcyclecounter(); st_acc[4] += tb - tp; tp = tb; }
    #endif
            __syncthreads();
    #ifdef DRAM_WZY_STAMPS
            st_acc[5] += __builtin_readcyclecounter() - tp;
    #endif
            cur ^= 1;
            c_c0 += 4;
            if (c_c0 >= Cin) { c_c0 = 0; c_item += item_step; }
        }
    };
    if (st_extra) run(std::true_type{}); else run(std::false_type{});
}

__global__ __launch_bounds__(512, 1) void conv3d_k3_fwd_wzy_kernel(ConvArgs a, int total_items) { fwd_wzy_body<32>(a, total_items); }
__global__ __launch_bounds__(512, 1) void conv3d_k3_fwd_wzy16_kernel(ConvArgs a, int total_items) { fwd_wzy_body<16>(a, total_items); }

// ---------------------------------------------------------------------------------------------
struct WgradArgs {
    CatView src;      // x (possibly a virtual concatenation)
    const float* dy;  // [N][Cout][D][H][W]
    float* slabs;     // [SPLIT][Cout][Cin][27]
    int N, Cin, Cout, D, H, W;
    int nbx, nby, nbz, nboxes, split, ci_tiles, co_tiles;
    int ci_tile0;     // first 16-channel ci tile of this launch ((z,y) kernel: a launch may cover the tiles of ONE source only)
    // normalise + ReLU on load of x (see ConvArgs::coef1): source k holds the RAW conv output, the operand is
    // act(coefk[row][0] * x + coefk[row][1]); Winograd kernel only (the host materialises for the others)
    const float* coef1;
    const float* coef2;
    int relu1, relu2;
};

template <int HVv>
struct PadTo2Mod32 {
    static constexpr int value = HVv + ((2 - (HVv % 32)) + 32) % 32;
};

// backward-weights kernel: one 512-thread block (8 waves, two per SIMD) per CU.
//   block tile = (16*COS output channels) x (16*CIT input channels) x 27 taps, COS*CIT = 8 waves;
//   each wave owns one 16x16 (co,ci) sub-tile for all 27 taps (27 accumulators of 4 registers).
//   COS=8,CIT=1 for wide layers; COS=4,CIT=2 when Cout <= 64 (the full-resolution layers).
// Software pipeline: the global loads of the next 64-voxel box (dY and X halo, <= 48 registers per
// thread, buffer loads with hardware zero fill) are in flight while the MFMAs of the current box
// run; they are written to the single-buffered LDS tile between two barriers afterwards.  Inside
// a box the LDS operand reads of k-step s+1 are issued ahead of the MFMAs of k-step s.
template <int BX, int BY, int BZ, int COS, int CIT>
struct WgradGeom {
    static constexpr int T = 512;
    static constexpr int VOX = BX * BY * BZ;
    static constexpr int PA = VOX + 2;  // co stride of the dY tile: == 2 (mod 32) -> conflict-free A reads
    static constexpr int HX = BX + 2, HY = BY + 2, HZ = BZ + 2;
    static constexpr int HV = HX * HY * HZ;
    static constexpr int PB = PadTo2Mod32<HV>::value;  // ci stride of the X halo tile
    static constexpr int NQ = (HV + T - 1) / T;
    static constexpr int CO_B = 16 * COS, CI_B = 16 * CIT;
    static constexpr int ROWS_PER_PASS = T / VOX;       // dY channel rows staged per pass
    static constexpr int DYQ = CO_B / ROWS_PER_PASS;    // dY registers per thread
    static constexpr size_t LDS_BYTES = (size_t)(CO_B * PA + CI_B * PB) * sizeof(float);
};

template <int BX, int BY, int BZ, int COS, int CIT>
__global__ __launch_bounds__(512, 2) void conv3d_k3_wgrad_kernel(WgradArgs a) {
    using G = WgradGeom<BX, BY, BZ, COS, CIT>;
    constexpr int T = G::T, VOX = G::VOX, PA = G::PA, HX = G::HX, HY = G::HY, HV = G::HV, PB = G::PB, NQ = G::NQ;
    constexpr int CO_B = G::CO_B, CI_B = G::CI_B, RPP = G::ROWS_PER_PASS, DYQ = G::DYQ;
    static_assert(COS * CIT == 8 && T % VOX == 0 && VOX % 64 == 0 && BX % 4 == 0 && PA % 32 == 2 && NQ <= 2 &&
                      CO_B % RPP == 0, "block geometry");

    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* ldy = lds;             // [CO_B][PA]
    float* lx = lds + CO_B * PA;  // [CI_B][PB]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wco = wave % COS, wci = wave / COS;   // this wave's (co, ci) sub-tile
    int b = xcd_remap(blockIdx.x, gridDim.x);       // logical item = (split, co tile, ci tile), ci tile fastest
    const int ci_t = b % a.ci_tiles; b /= a.ci_tiles;
    const int co_t = b % a.co_tiles;
    const int sp = b / a.co_tiles;
    const int ci0 = ci_t * CI_B, co0 = co_t * CO_B;
    const int D = a.D, H = a.H, W = a.W;
    const int S = D * H * W;
    const int S2 = a.src.D2 * a.src.H2 * a.src.W2;

    f32x4 acc[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int i = lane & 15, k = lane >> 4;
    // dY staging role: voxel v_st of the box, channel rows co_st + RPP*q (co_st is wave-uniform)
    const int v_st = tid % VOX;
    const int co_st = __builtin_amdgcn_readfirstlane(tid / VOX);
    const int svx = v_st % BX, svy = (v_st / BX) % BY, svz = v_st / (BX * BY);

    float rdy[DYQ];
    float rx[CI_B][NQ];

    // One buffer descriptor per tensor and sample (num_records = the sample's bytes); the channel plane
    // is folded into the per-lane byte offset, which runs from channel to channel with one VALU add.
    // No per-channel condition exists: a channel beyond the tensor's last one is an offset beyond
    // num_records and the hardware returns 0, and an out-of-volume halo lane keeps the OOB offset (its
    // increment is 0).  (Per-plane descriptors / scalar offsets / per-channel predicates are all loop
    // invariant, get hoisted out of the box loop and need >100 SGPRs: hipcc then spills them through
    // v_writelane/v_readlane, ~700 extra VALU instructions per box, measured.)
    const unsigned S4 = 4u * (unsigned)S, S24 = 4u * (unsigned)S2;
    // which tensor(s) this block's CI_B channels come from: 0 = x1 only, 1 = x2 only, 2 = straddles both
    const int src_mode = (a.src.p2 == nullptr || ci0 + CI_B <= a.src.C1) ? 0 : (ci0 >= a.src.C1 ? 1 : 2);
    auto load_box = [&](int box) {
        int bb = box;
        const int bx = bb % a.nbx; bb /= a.nbx;
        const int by = bb % a.nby; bb /= a.nby;
        const int bz = bb % a.nbz;
        const int n = bb / a.nbz;
        const int x0 = bx * BX, y0 = by * BY, z0 = bz * BZ;
        {   // dY[CO_B][VOX]
            const int gx = x0 + svx, gy = y0 + svy, gz = z0 + svz;
            const bool vok = gx < W && gy < H && gz < D;
            const __amdgpu_buffer_rsrc_t srd =
                make_rsrc(uniform_ptr(a.dy + (size_t)n * a.Cout * S), (unsigned)a.Cout * S4);
            unsigned run = vok ? 4u * (unsigned)((gz * H + gy) * W + gx) + (unsigned)(co0 + co_st) * S4 : OOB;
            const unsigned inc = vok ? (unsigned)RPP * S4 : 0u;
#pragma unroll
            for (int q = 0; q < DYQ; ++q) {
                rdy[q] = buf_load(srd, run, 0);
                run += inc;
            }
        }
        {   // X[CI_B][halo]
            unsigned run1[NQ], run2[NQ], inc1[NQ], inc2[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int e = tid + T * q;
                const int hx = e % HX, hy = (e / HX) % HY, hz = e / (HX * HY);
                const int gx = x0 - 1 + hx, gy = y0 - 1 + hy, gz = z0 - 1 + hz;
                const bool ok = (e < HV) && gx >= 0 && gx < W && gy >= 0 && gy < H && gz >= 0 && gz < D;
                run1[q] = ok ? 4u * (unsigned)((gz * H + gy) * W + gx) + (unsigned)ci0 * S4 : OOB;
                // (ci0 - C1) may be negative: the unsigned offset wraps and is out of range until ci reaches C1
                run2[q] = ok ? 4u * (unsigned)(((gz + a.src.oz) * a.src.H2 + gy + a.src.oy) * a.src.W2 + gx + a.src.ox) +
                                   (unsigned)(ci0 - a.src.C1) * S24 : OOB;
                inc1[q] = ok ? S4 : 0u;
                inc2[q] = ok ? S24 : 0u;
            }
            const __amdgpu_buffer_rsrc_t srd1 =
                make_rsrc(uniform_ptr(a.src.p1 + (size_t)n * a.src.C1 * S), (unsigned)a.src.C1 * S4);
            if (src_mode == 0) {
#pragma unroll
                for (int c = 0; c < CI_B; ++c)
#pragma unroll
                    for (int q = 0; q < NQ; ++q) { rx[c][q] = buf_load(srd1, run1[q], 0); run1[q] += inc1[q]; }
            } else {
                const __amdgpu_buffer_rsrc_t srd2 =
                    make_rsrc(uniform_ptr(a.src.p2 + (size_t)n * a.src.C2 * S2), (unsigned)a.src.C2 * S24);
                if (src_mode == 1) {
#pragma unroll
                    for (int c = 0; c < CI_B; ++c)
#pragma unroll
                        for (int q = 0; q < NQ; ++q) { rx[c][q] = buf_load(srd2, run2[q], 0); run2[q] += inc2[q]; }
                } else {   // rare: the tile straddles the two tensors -> load from both, one of them returns 0
#pragma unroll
                    for (int c = 0; c < CI_B; ++c)
#pragma unroll
                        for (int q = 0; q < NQ; ++q) {
                            rx[c][q] = buf_load(srd1, run1[q], 0) + buf_load(srd2, run2[q], 0);
                            run1[q] += inc1[q];
                            run2[q] += inc2[q];
                        }
                }
            }
        }
    };
    auto store_box = [&]() {
#pragma unroll
        for (int q = 0; q < DYQ; ++q) ldy[(co_st + RPP * q) * PA + v_st] = rdy[q];
#pragma unroll
        for (int c = 0; c < CI_B; ++c)
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                if (tid + T * q < HV) lx[c * PB + tid + T * q] = rx[c][q];
    };
    auto compute = [&]() {
        // VOX/4 k-steps (4 voxels along x each) x 27 taps of 16x16x4 MFMAs
        const float* ap = ldy + (wco * 16 + i) * PA + k;
        const float* bp = lx + (wci * 16 + i) * PB + k;
        constexpr int NS = VOX / 4;
        float av[2], bv[2][27];
        // two operand sets live; one operand read of k-step s beside each MFMA of k-step s-1 (see the vec kernel)
#pragma unroll
        for (int s = 0; s <= NS; ++s) {
            const int x4 = s % (BX / 4), vy = (s / (BX / 4)) % BY, vz = s / ((BX / 4) * BY);
            const float* bq = bp + (vz * HY + vy) * HX + 4 * x4;
            if (s < NS) av[s & 1] = ap[(vz * BY + vy) * BX + 4 * x4];
#pragma unroll
            for (int tap = 0; tap < 27; ++tap) {
                const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
                if (s < NS) bv[s & 1][tap] = bq[(dz * HY + dy) * HX + dx];
                if (s > 0)
                    acc[tap] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[(s - 1) & 1], bv[(s - 1) & 1][tap], acc[tap], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    if (sp < a.nboxes) {
        load_box(sp);
        store_box();
    }
    __syncthreads();
    for (int box = sp; box < a.nboxes; box += a.split) {
        const bool has_next = (box + a.split) < a.nboxes;
        if (has_next) load_box(box + a.split);      // global loads in flight during the MFMAs
        __builtin_amdgcn_sched_barrier(0);
        compute();
        __builtin_amdgcn_sched_barrier(0);
        if (has_next) {
            __syncthreads();                         // every wave is done reading the tile
            store_box();
            __syncthreads();
        }
    }

    // ---- partial slab: D row = co (4*(lane>>4)+r), col = ci (lane&15) ----
    const int ci = ci0 + wci * 16 + i;
    if (ci < a.Cin) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = co0 + wco * 16 + 4 * k + r;
            if (co < a.Cout) {
                float* o = a.slabs + (((size_t)sp * a.Cout + co) * a.Cin + ci) * 27;
#pragma unroll
                for (int tap = 0; tap < 27; ++tap) o[tap] = acc[tap][r];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// backward-weights, 16-byte staging variant (W a multiple of the box width, channel tile inside one
// source tensor).  Same tiling and MFMA loop as conv3d_k3_wgrad_kernel; only the way a box gets into
// LDS differs: s_memtime stamps showed 3-5k of the 34k cycles per box going into the issue of the 40
// dword loads per thread (320 wave-level VMEM instructions per CU and box through the vector-memory
// path) plus 1.4k into the 40 ds_write_b32.  Here a thread moves 16-byte pieces: dY rows as float4
// (buffer_load_dwordx4 + ds_write_b128), X halo rows as 8 aligned float4 of interior + 2 edge dwords
// (rows padded to BX+8 floats so that the interior starts 16-byte aligned) -> 10-12 VMEM instructions.
template <int BX, int BY, int BZ, int COS, int CIT>
struct WgradVecGeom {
    static constexpr int T = 512;
    static constexpr int VOX = BX * BY * BZ;
    static constexpr int PA = VOX + 4;                  // multiple of 4: ds_write_b128 rows (A reads 2-way conflicted: 1 of 28 reads)
    static constexpr int HY = BY + 2, HZ = BZ + 2;
    static constexpr int HXP = BX + 8;                  // padded halo row: [3 pad][left][BX interior][right][3 pad]
    static constexpr int RPC = HY * HZ;                 // halo rows per channel
    static constexpr int PB = PadTo2Mod32<RPC * HXP>::value;
    static constexpr int CO_B = 16 * COS, CI_B = 16 * CIT;
    // a thread keeps the same (row, column) role in every pass and only the channel advances, by a fixed
    // number of channels per pass: one base offset and one validity test per kind of slot instead of one
    // per slot (registers are what this kernel is short of)
    static constexpr int DY_TPC = VOX / 4, DY_CPP = T / DY_TPC, DYP = (CO_B + DY_CPP - 1) / DY_CPP;       // dY float4 slots
    static constexpr int XI_TPC = RPC * (BX / 4), XI_CPP = T / XI_TPC, XIP = (CI_B + XI_CPP - 1) / XI_CPP; // X interior float4 slots
    static constexpr int XE_TPC = RPC * 2, XE_CPP = T / XE_TPC, XEP = (CI_B + XE_CPP - 1) / XE_CPP;       // X edge dword slots
    static constexpr size_t LDS_BYTES = (size_t)(CO_B * PA + CI_B * PB) * sizeof(float);
};

template <int BX, int BY, int BZ, int COS, int CIT>
__global__ __launch_bounds__(512, 2) void conv3d_k3_wgrad_vec_kernel(WgradArgs a) {
    using G = WgradVecGeom<BX, BY, BZ, COS, CIT>;
    constexpr int VOX = G::VOX, PA = G::PA, HY = G::HY, HXP = G::HXP, PB = G::PB;
    constexpr int CO_B = G::CO_B, CI_B = G::CI_B, DYP = G::DYP, XIP = G::XIP, XEP = G::XEP;
    constexpr int DY_TPC = G::DY_TPC, DY_CPP = G::DY_CPP, XI_TPC = G::XI_TPC, XI_CPP = G::XI_CPP, XE_TPC = G::XE_TPC, XE_CPP = G::XE_CPP;
    static_assert(COS * CIT == 8 && BX % 4 == 0 && PB % 2 == 0 && VOX % 4 == 0 && DY_CPP >= 1 && XI_CPP >= 1 && XE_CPP >= 1,
                  "block geometry");

    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* ldy = lds;             // [CO_B][PA]
    float* lx = lds + CO_B * PA;  // [CI_B][PB], row r of channel c at c*PB + r*HXP, halo x index h at column h+3

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wco = wave % COS, wci = wave / COS;
    int b = xcd_remap(blockIdx.x, gridDim.x);
    const int ci_t = b % a.ci_tiles; b /= a.ci_tiles;
    const int co_t = b % a.co_tiles;
    const int sp = b / a.co_tiles;
    const int ci0 = ci_t * CI_B, co0 = co_t * CO_B;
    const int D = a.D, H = a.H, W = a.W;
    const int S = D * H * W;
    // the source tensor of this block's channel tile (the host guarantees the tile does not straddle)
    const bool use2 = a.src.p2 != nullptr && ci0 >= a.src.C1;
    const int Hs = use2 ? a.src.H2 : H, Ws = use2 ? a.src.W2 : W;
    const int Ss = use2 ? a.src.D2 * a.src.H2 * a.src.W2 : S;
    const int oz = use2 ? a.src.oz : 0, oy = use2 ? a.src.oy : 0, ox = use2 ? a.src.ox : 0;
    const int Cs = use2 ? a.src.C2 : a.src.C1;
    const int cs0 = use2 ? ci0 - a.src.C1 : ci0;          // first channel of the tile inside its tensor
    const float* xs = use2 ? a.src.p2 : a.src.p1;
    const unsigned S4 = 4u * (unsigned)S, Ss4 = 4u * (unsigned)Ss;

    f32x4 acc[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int i = lane & 15, k = lane >> 4;

    f32x4 rdy[DYP];
    f32x4 rxi[XIP];
    float rxe[XEP];

    // thread-constant slot roles
    // dY: channel d_c + p*DY_CPP, 4 voxels starting at (d_vz, d_vy, d_vx)
    const int d_c = tid / DY_TPC, d_v = 4 * (tid % DY_TPC);
    const int d_vx = d_v % BX, d_vy = (d_v / BX) % BY, d_vz = d_v / (BX * BY);
    const bool d_act = d_c < DY_CPP;
    const unsigned d_rel = (unsigned)(co0 + d_c) * S4 + 4u * (unsigned)((d_vz * H + d_vy) * W + d_vx);
    // X interior: channel i_c + p*XI_CPP, halo row (i_hz, i_hy), columns 4*i_j..+3
    const int i_c = tid / XI_TPC, i_r = (tid % XI_TPC) / (BX / 4), i_j = tid % (BX / 4);
    const int i_hy = i_r % HY, i_hz = i_r / HY;
    const bool i_act = i_c < XI_CPP;
    const unsigned i_rel = (unsigned)(cs0 + i_c) * Ss4 + 4u * (unsigned)((i_hz * Hs + i_hy) * Ws + 4 * i_j);
    // X edges: channel e_c + p*XE_CPP, halo row (e_hz, e_hy), side 0 = x0-1, 1 = x0+BX
    const int e_c = tid / XE_TPC, e_r = (tid % XE_TPC) / 2, e_side = tid % 2;
    const int e_hy = e_r % HY, e_hz = e_r / HY;
    const bool e_act = e_c < XE_CPP;
    const unsigned e_rel = (unsigned)(cs0 + e_c) * Ss4 + 4u * (unsigned)((e_hz * Hs + e_hy) * Ws + (e_side ? BX : -1));

    auto load_box = [&](int box) {
        int bb = box;
        const int bx = bb % a.nbx; bb /= a.nbx;
        const int by = bb % a.nby; bb /= a.nby;
        const int bz = bb % a.nbz;
        const int n = bb / a.nbz;
        const int x0 = bx * BX, y0 = by * BY, z0 = bz * BZ;
        {   // dY: rows of the box are full in x (W % BX == 0); rows beyond H / D read 0, and so do channels
            // >= Cout (their offset is beyond the descriptor's num_records)
            const __amdgpu_buffer_rsrc_t srd = make_rsrc(uniform_ptr(a.dy + (size_t)n * a.Cout * S), (unsigned)a.Cout * S4);
            const bool ok = d_act && (y0 + d_vy) < H && (z0 + d_vz) < D;
            unsigned run = ok ? 4u * (unsigned)((z0 * H + y0) * W + x0) + d_rel : OOB;
            const unsigned inc = ok ? (unsigned)DY_CPP * S4 : 0u;
#pragma unroll
            for (int p = 0; p < DYP; ++p) {
                rdy[p] = buf_load4(srd, (DYP * DY_CPP == CO_B || d_c + p * DY_CPP < CO_B) ? run : OOB, 0);
                run += inc;
            }
        }
        {   // X halo rows: halo origin = (z0-1, y0-1, x0) of the (cropped) source plane
            const __amdgpu_buffer_rsrc_t srd = make_rsrc(uniform_ptr(xs + (size_t)n * Cs * Ss), (unsigned)Cs * Ss4);
            const unsigned origin = 4u * (unsigned)(((z0 - 1 + oz) * Hs + (y0 - 1 + oy)) * Ws + x0 + ox);   // may wrap for rows above the volume: those are invalid
            {
                const bool ok = i_act && (unsigned)(y0 - 1 + i_hy) < (unsigned)H && (unsigned)(z0 - 1 + i_hz) < (unsigned)D;
                unsigned run = ok ? origin + i_rel : OOB;
                const unsigned inc = ok ? (unsigned)XI_CPP * Ss4 : 0u;
#pragma unroll
                for (int p = 0; p < XIP; ++p) {
                    rxi[p] = buf_load4(srd, (XIP * XI_CPP == CI_B || i_c + p * XI_CPP < CI_B) ? run : OOB, 0);
                    run += inc;
                }
            }
            {
                const bool ok = e_act && (unsigned)(y0 - 1 + e_hy) < (unsigned)H && (unsigned)(z0 - 1 + e_hz) < (unsigned)D &&
                                (e_side ? (x0 + BX) < W : x0 > 0);
                unsigned run = ok ? origin + e_rel : OOB;
                const unsigned inc = ok ? (unsigned)XE_CPP * Ss4 : 0u;
#pragma unroll
                for (int p = 0; p < XEP; ++p) {
                    rxe[p] = buf_load(srd, (XEP * XE_CPP == CI_B || e_c + p * XE_CPP < CI_B) ? run : OOB, 0);
                    run += inc;
                }
            }
        }
    };
    auto store_box = [&]() {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int p = 0; p < DYP; ++p)
            if (d_act && (DYP * DY_CPP == CO_B || d_c + p * DY_CPP < CO_B))
                *reinterpret_cast<f32x4*>(ldy + (d_c + p * DY_CPP) * PA + d_v) = rdy[p];
#pragma unroll
        for (int p = 0; p < XIP; ++p)
            if (i_act && (XIP * XI_CPP == CI_B || i_c + p * XI_CPP < CI_B)) {
                float* d = lx + (i_c + p * XI_CPP) * PB + i_r * HXP + 4 + 4 * i_j;       // 8-byte aligned
                *reinterpret_cast<f32x2*>(d) = f32x2{rxi[p][0], rxi[p][1]};
                *reinterpret_cast<f32x2*>(d + 2) = f32x2{rxi[p][2], rxi[p][3]};
            }
#pragma unroll
        for (int p = 0; p < XEP; ++p)
            if (e_act && (XEP * XE_CPP == CI_B || e_c + p * XE_CPP < CI_B))
                lx[(e_c + p * XE_CPP) * PB + e_r * HXP + (e_side ? BX + 4 : 3)] = rxe[p];
    };
    auto compute = [&]() {
        const float* ap = ldy + (wco * 16 + i) * PA + k;
        const float* bp = lx + (wci * 16 + i) * PB + k + 3;
        constexpr int NS = VOX / 4;
        float av[2], bv[2][27];
        // one operand read of k-step s beside each MFMA of k-step s-1: the 28 LDS reads of a step are spread over
        // its 27 MFMA slots instead of being issued as one burst that the 4-bit lgkmcnt counter throttles
#pragma unroll
        for (int s = 0; s <= NS; ++s) {
            const int x4 = s % (BX / 4), vy = (s / (BX / 4)) % BY, vz = s / ((BX / 4) * BY);
            const float* bq = bp + (vz * HY + vy) * HXP + 4 * x4;
            if (s < NS) av[s & 1] = ap[(vz * BY + vy) * BX + 4 * x4];
#pragma unroll
            for (int tap = 0; tap < 27; ++tap) {
                const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
                if (s < NS) bv[s & 1][tap] = bq[(dz * HY + dy) * HXP + dx];
                if (s > 0)
                    acc[tap] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[(s - 1) & 1], bv[(s - 1) & 1][tap], acc[tap], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    if (sp < a.nboxes) {
        load_box(sp);
        store_box();
    }
    __syncthreads();
    for (int box = sp; box < a.nboxes; box += a.split) {
        const bool has_next = (box + a.split) < a.nboxes;
        if (has_next) load_box(box + a.split);
        __builtin_amdgcn_sched_barrier(0);
        compute();
        __builtin_amdgcn_sched_barrier(0);
        if (has_next) {
            __syncthreads();
            store_box();
            __syncthreads();
        }
    }

    const int ci = ci0 + wci * 16 + i;
    if (ci < a.Cin) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = co0 + wco * 16 + 4 * k + r;
            if (co < a.Cout) {
                float* o = a.slabs + (((size_t)sp * a.Cout + co) * a.Cin + ci) * 27;
#pragma unroll
                for (int tap = 0; tap < 27; ++tap) o[tap] = acc[tap][r];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// backward-weights with the transpose of Winograd F(2,3) along z (exact fp32 arithmetic, 2/3 of the MFMAs).
// For a plane pair (z0, z0+1) with gradients e0, e1 and inputs d0..d3 = x[z0-1..z0+2] the three z taps of a
// (ky,kx) column receive  dg0 = e0 d0 + e1 d1,  dg1 = e0 d1 + e1 d2,  dg2 = e0 d2 + e1 d3  (6 products).
// Transposing the bilinear algorithm of the forward kernel:
//     p0 = e0 (d0-d2),  p1 = (e0+e1)(d1+d2),  p2 = (e0-e1)(d2-d1),  p3 = -e1 (d1-d3)      (4 products)
//     dg0 = p0 + (p1+p2)/2,   dg1 = (p1-p2)/2,   dg2 = (p1+p2)/2 + p3.
// So: four GEMMs (one per transformed plane xi) dP[xi][co][ci][ky,kx] += E_xi[co][pos] * T_xi[ci][pos+(ky,kx)] with
// K = the (y,x) positions of the plane pair; 9 x 4 = 36 accumulators of 16x16 per wave instead of 27, and 36
// MFMAs per 2 voxels instead of 54.  Staging as in the 16-byte kernel above, except that a thread fetches its
// slot from both (dY) / all four (X) planes and stores the transformed planes.
template <int BX, int BY, int COS, int CIT>
struct WgradWzGeom {
    static constexpr int T = 64 * COS * CIT;            // 512: one block per CU; 256: two blocks per CU
    static constexpr int POS = BX * BY;                 // positions of the plane pair = K per box and xi
    static constexpr int RA = 4 * POS + 2;              // dY channel stride (4 planes): == 2 (mod 32) -> conflict-free A reads
    static constexpr int HY = BY + 2;
    static constexpr int HXP = BX + 4;                  // halo row: [left][BX interior][right][2 pad]
    static constexpr int PLX = HY * HXP;                // one transformed X plane
    static constexpr int PB = PadTo2Mod32<4 * PLX>::value;
    static constexpr int CO_B = 16 * COS, CI_B = 16 * CIT;
    static constexpr int DY_TPC = POS / 4, DY_CPP = T / DY_TPC, DYP = (CO_B + DY_CPP - 1) / DY_CPP;
    static constexpr int XI_TPC = HY * (BX / 4), XI_CPP = T / XI_TPC, XIP = (CI_B + XI_CPP - 1) / XI_CPP;
    static constexpr int XE_TPC = HY * 2, XE_CPP = T / XE_TPC, XEP = (CI_B + XE_CPP - 1) / XE_CPP;
    static constexpr int STAGE = CO_B * RA + CI_B * PB;  // floats of one (dY, X) tile pair
    // two stages when they fit (one barrier per box, stores not serialised behind the slowest wave); otherwise
    // (128 co tiles) a single stage with the stores between two barriers
    static constexpr bool DOUBLE = 2 * (size_t)STAGE * sizeof(float) <= 160 * 1024;
    static constexpr size_t LDS_BYTES = (DOUBLE ? 2 : 1) * (size_t)STAGE * sizeof(float);
};

#ifdef DRAM_WZY_STAMPS      // diagnostics build only (scripts/diag_wgrad_wz_stamps.py)
__device__ unsigned long long g_wgrad_stamps[8];
#endif

template <int BX, int BY, int COS, int CIT, bool LAZY = false>
__global__ __launch_bounds__(64 * COS * CIT, 2) void conv3d_k3_wgrad_wz_kernel(WgradArgs a) {
    using G = WgradWzGeom<BX, BY, COS, CIT>;
    constexpr int POS = G::POS, RA = G::RA, HXP = G::HXP, PLX = G::PLX, PB = G::PB, STAGE = G::STAGE;
    constexpr int CO_B = G::CO_B, CI_B = G::CI_B, DYP = G::DYP, XIP = G::XIP, XEP = G::XEP;
    constexpr int DY_TPC = G::DY_TPC, DY_CPP = G::DY_CPP, XI_TPC = G::XI_TPC, XI_CPP = G::XI_CPP, XE_TPC = G::XE_TPC, XE_CPP = G::XE_CPP;
    constexpr bool DOUBLE = G::DOUBLE;
    static_assert((COS * CIT == 8 || COS * CIT == 4) && BX % 4 == 0 && POS % 8 == 0 && PB % 2 == 0 && RA % 2 == 0 && DY_CPP >= 1 &&
                      XI_CPP >= 1 && XE_CPP >= 1, "block geometry");
    typedef float f32x2 __attribute__((ext_vector_type(2)));

    extern __shared__ __attribute__((aligned(16))) float lds[];
    // stage layout: dY [CO_B][4][POS] (+2), then X [CI_B][4][HY][HXP]; halo x index h of a row at column h

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wco = wave % COS, wci = wave / COS;
    int b = xcd_remap(blockIdx.x, gridDim.x);
    const int ci_t = b % a.ci_tiles; b /= a.ci_tiles;
    const int co_t = b % a.co_tiles;
    const int sp = b / a.co_tiles;
    const int ci0 = ci_t * CI_B, co0 = co_t * CO_B;
    const int D = a.D, H = a.H, W = a.W;
    const int S = D * H * W;
    const bool use2 = a.src.p2 != nullptr && ci0 >= a.src.C1;
    const int Hs = use2 ? a.src.H2 : H, Ws = use2 ? a.src.W2 : W;
    const int Ss = use2 ? a.src.D2 * a.src.H2 * a.src.W2 : S;
    const int oz = use2 ? a.src.oz : 0, oy = use2 ? a.src.oy : 0, ox = use2 ? a.src.ox : 0;
    const int Cs = use2 ? a.src.C2 : a.src.C1;
    const int cs0 = use2 ? ci0 - a.src.C1 : ci0;
    const float* xs = use2 ? a.src.p2 : a.src.p1;
    const unsigned S4 = 4u * (unsigned)S, Ss4 = 4u * (unsigned)Ss;
    const unsigned zs4 = 4u * (unsigned)(H * W), zss4 = 4u * (unsigned)(Hs * Ws);   // plane strides in bytes

    f32x4 acc[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[t][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int i = lane & 15, k = lane >> 4;

    f32x4 rdy[DYP][2];
    f32x4 rxi[XIP][4];
    float rxe[XEP][4];
    // LAZY: per staged slot the {a, b} of its channel row and the validity of its four planes (the zero padding
    // belongs to the ACTIVATED tensor: out-of-volume halo elements must stay 0, not act(b))
    const float* cf = LAZY ? (use2 ? a.coef2 : a.coef1) : nullptr;   // block-uniform; null: this source is plain
    const float lo = (LAZY && (use2 ? a.relu2 : a.relu1)) ? 0.f : -INFINITY;
    float cia[XIP], cib[XIP], cea[XEP], ceb[XEP];
    unsigned imask = 0, emask = 0;

    // thread-constant slot roles (see conv3d_k3_wgrad_vec_kernel)
    const int d_c = tid / DY_TPC, d_v = 4 * (tid % DY_TPC);
    const int d_vx = d_v % BX, d_vy = d_v / BX;
    const bool d_act = d_c < DY_CPP;
    const unsigned d_rel = (unsigned)(co0 + d_c) * S4 + 4u * (unsigned)(d_vy * W + d_vx);
    const int i_c = tid / XI_TPC, i_r = (tid % XI_TPC) / (BX / 4), i_j = tid % (BX / 4);     // i_r = halo row
    const bool i_act = i_c < XI_CPP;
    const unsigned i_rel = (unsigned)(cs0 + i_c) * Ss4 + 4u * (unsigned)(i_r * Ws + 4 * i_j);
    const int e_c = tid / XE_TPC, e_r = (tid % XE_TPC) / 2, e_side = tid % 2;
    const bool e_act = e_c < XE_CPP;
    const unsigned e_rel = (unsigned)(cs0 + e_c) * Ss4 + 4u * (unsigned)(e_r * Ws + (e_side ? BX : -1));

    // NB every validity test below is folded into ONE select per load (non-short-circuit & on ints): a chain of &&
    // makes hipcc split a load into two exec-masked loads of the same registers, with an s_waitcnt vmcnt(0) between
    // them -- a full memory latency inside the issue sequence, per plane and pass
    auto load_box = [&](int box) {
        int bb = box;
        const int bx = bb % a.nbx; bb /= a.nbx;
        const int by = bb % a.nby; bb /= a.nby;
        const int bz = bb % a.nbz;
        const int n = bb / a.nbz;
        const int x0 = bx * BX, y0 = by * BY, z0 = 2 * bz;
        {   // dY, planes z0 and z0+1 (full rows in x: W % BX == 0)
            const __amdgpu_buffer_rsrc_t srd = make_rsrc(uniform_ptr(a.dy + (size_t)n * a.Cout * S), (unsigned)a.Cout * S4);
            const int ok = (int)d_act & (int)((y0 + d_vy) < H);
            const int ok1 = ok & (int)((z0 + 1) < D);
            const unsigned base = 4u * (unsigned)((z0 * H + y0) * W + x0) + d_rel;
            unsigned run0 = ok ? base : OOB, run1 = ok1 ? base + zs4 : OOB;
            const unsigned inc0 = ok ? (unsigned)DY_CPP * S4 : 0u, inc1 = ok1 ? (unsigned)DY_CPP * S4 : 0u;
#pragma unroll
            for (int p = 0; p < DYP; ++p) {
                const int in = (DYP * DY_CPP == CO_B) | (int)(d_c + p * DY_CPP < CO_B);
                rdy[p][0] = buf_load4(srd, in ? run0 : OOB, 0);
                rdy[p][1] = buf_load4(srd, in ? run1 : OOB, 0);
                run0 += inc0;
                run1 += inc1;
            }
        }
        {   // X halo rows of the four planes z0-1 .. z0+2
            const __amdgpu_buffer_rsrc_t srd = make_rsrc(uniform_ptr(xs + (size_t)n * Cs * Ss), (unsigned)Cs * Ss4);
            const unsigned origin = 4u * (unsigned)(((z0 - 1 + oz) * Hs + (y0 - 1 + oy)) * Ws + x0 + ox);   // may wrap: invalid rows are masked
            {
                const int okr = (int)i_act & (int)((unsigned)(y0 - 1 + i_r) < (unsigned)H);
                unsigned run[4], inc[4];
                if (LAZY) imask = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int okq = okr & (int)((unsigned)(z0 - 1 + q) < (unsigned)D);
                    run[q] = okq ? origin + i_rel + (unsigned)q * zss4 : OOB;
                    inc[q] = okq ? (unsigned)XI_CPP * Ss4 : 0u;
                    if (LAZY) imask |= (unsigned)okq << q;
                }
#pragma unroll
                for (int p = 0; p < XIP; ++p) {
                    const int in = (XIP * XI_CPP == CI_B) | (int)(i_c + p * XI_CPP < CI_B);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        rxi[p][q] = buf_load4(srd, in ? run[q] : OOB, 0);
                        run[q] += inc[q];
                    }
                    if (LAZY && cf) {
                        const int ch = min(cs0 + i_c + p * XI_CPP, Cs - 1);       // (clamped: tail slots are never stored)
                        const float2 ab = *reinterpret_cast<const float2*>(cf + 2 * ((int64_t)n * Cs + ch));
                        cia[p] = ab.x; cib[p] = ab.y;
                    }
                }
            }
            {
                const int okr = (int)e_act & (int)((unsigned)(y0 - 1 + e_r) < (unsigned)H) &
                                (e_side ? (int)((x0 + BX) < W) : (int)(x0 > 0));
                unsigned run[4], inc[4];
                if (LAZY) emask = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int okq = okr & (int)((unsigned)(z0 - 1 + q) < (unsigned)D);
                    run[q] = okq ? origin + e_rel + (unsigned)q * zss4 : OOB;
                    inc[q] = okq ? (unsigned)XE_CPP * Ss4 : 0u;
                    if (LAZY) emask |= (unsigned)okq << q;
                }
#pragma unroll
                for (int p = 0; p < XEP; ++p) {
                    const int in = (XEP * XE_CPP == CI_B) | (int)(e_c + p * XE_CPP < CI_B);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        rxe[p][q] = buf_load(srd, in ? run[q] : OOB, 0);
                        run[q] += inc[q];
                    }
                    if (LAZY && cf) {
                        const int ch = min(cs0 + e_c + p * XE_CPP, Cs - 1);
                        const float2 ab = *reinterpret_cast<const float2*>(cf + 2 * ((int64_t)n * Cs + ch));
                        cea[p] = ab.x; ceb[p] = ab.y;
                    }
                }
            }
        }
    };
    // the staged registers go to LDS in NPIECE pieces (one pass of one slot kind each)
    constexpr int NPIECE = DYP + XIP + XEP;
    auto store_piece = [&](float* st, int piece) {
        float* ldy = st;
        float* lx = st + CO_B * RA;
        if (piece < DYP) {
            const int p = piece;
            if (d_act && (DYP * DY_CPP == CO_B || d_c + p * DY_CPP < CO_B)) {
                const f32x4 e0 = rdy[p][0], e1 = rdy[p][1];
                const f32x4 pl[4] = {e0, e0 + e1, e0 - e1, -e1};
                float* d = ldy + (d_c + p * DY_CPP) * RA + d_v;        // 8-byte aligned
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    *reinterpret_cast<f32x2*>(d + q * POS) = f32x2{pl[q][0], pl[q][1]};
                    *reinterpret_cast<f32x2*>(d + q * POS + 2) = f32x2{pl[q][2], pl[q][3]};
                }
            }
        } else if (piece < DYP + XIP) {
            const int p = piece - DYP;
            if (i_act && (XIP * XI_CPP == CI_B || i_c + p * XI_CPP < CI_B)) {
                f32x4 dq[4] = {rxi[p][0], rxi[p][1], rxi[p][2], rxi[p][3]};
                if (LAZY && cf) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const bool okq = (imask >> q) & 1u;
#pragma unroll
                        for (int u = 0; u < 4; ++u) dq[q][u] = okq ? fmaxf(fmaf(cia[p], dq[q][u], cib[p]), lo) : 0.f;
                    }
                }
                const f32x4 d0 = dq[0], d1 = dq[1], d2 = dq[2], d3 = dq[3];
                const f32x4 pl[4] = {d0 - d2, d1 + d2, d2 - d1, d1 - d3};
                float* d = lx + (i_c + p * XI_CPP) * PB + i_r * HXP + 1 + 4 * i_j;       // interior starts at column 1
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    d[q * PLX] = pl[q][0];
                    d[q * PLX + 1] = pl[q][1];
                    d[q * PLX + 2] = pl[q][2];
                    d[q * PLX + 3] = pl[q][3];
                }
            }
        } else {
            const int p = piece - DYP - XIP;
            if (e_act && (XEP * XE_CPP == CI_B || e_c + p * XE_CPP < CI_B)) {
                float eq[4] = {rxe[p][0], rxe[p][1], rxe[p][2], rxe[p][3]};
                if (LAZY && cf) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) eq[q] = ((emask >> q) & 1u) ? fmaxf(fmaf(cea[p], eq[q], ceb[p]), lo) : 0.f;
                }
                const float d0 = eq[0], d1 = eq[1], d2 = eq[2], d3 = eq[3];
                float* d = lx + (e_c + p * XE_CPP) * PB + e_r * HXP + (e_side ? BX + 1 : 0);
                d[0] = d0 - d2;
                d[PLX] = d1 + d2;
                d[2 * PLX] = d2 - d1;
                d[3 * PLX] = d1 - d3;
            }
        }
    };
    auto store_box = [&](float* st) {
#pragma unroll
        for (int piece = 0; piece < NPIECE; ++piece) store_piece(st, piece);
    };
    auto compute = [&](const float* st) {
        const float* ap = st + (wco * 16 + i) * RA + k;
        const float* bp = st + CO_B * RA + (wci * 16 + i) * PB + k;
        constexpr int NU = POS;     // (POS/4 k-steps) x (4 planes)
        float av[2], bv[2][9];
        // one operand read of step u beside each MFMA of step u-1
#pragma unroll
        for (int u = 0; u <= NU; ++u) {
            const int s = u / 4, xi = u % 4;
            const int x4 = s % (BX / 4), vy = s / (BX / 4);
            const float* bq = bp + xi * PLX + vy * HXP + 4 * x4;
            const int xp = (u - 1) % 4;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                if (u < NU) {
                    if (tap == 0) av[u & 1] = ap[xi * POS + 4 * s];
                    bv[u & 1][tap] = bq[(tap / 3) * HXP + tap % 3];
                }
                if (u > 0)
                    acc[tap][xp] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[(u - 1) & 1], bv[(u - 1) & 1][tap], acc[tap][xp], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    if (sp < a.nboxes) {
        load_box(sp);
        store_box(lds);
    }
    __syncthreads();
    int cur = 0;
#ifdef DRAM_WZY_STAMPS
    unsigned long long st_acc[5] = {};
#define WG_STAMP(i_) { const unsigned long long tn_ = __builtin_readcyclecounter(); st_acc[i_] += tn_ - tp_; tp_ = tn_; }
#else
#define WG_STAMP(i_)
#endif
    for (int box = sp; box < a.nboxes; box += a.split) {
        const bool has_next = (box + a.split) < a.nboxes;
#ifdef DRAM_WZY_STAMPS
        unsigned long long tp_ = __builtin_readcyclecounter();
        st_acc[4] += 1;
#endif
        if (has_next) load_box(box + a.split);
        __builtin_amdgcn_sched_barrier(0);
        WG_STAMP(0)
        if (DOUBLE) {
            // a wave writes the next box into the idle stage as soon as ITS MFMAs are done -- while slower waves still
            // compute -- and a box costs one barrier.  What the stamps (-DDRAM_WZY_STAMPS, scripts/diag_wgrad_wz_stamps.py)
            // show: every wave issues its 288 MFMAs in 9,400 cycles -- the pipe's full rate -- so the two waves of a SIMD
            // run their MFMA loops one after the other (the older first; the younger makes no progress meanwhile, not even
            // through its load phase), and a box takes 21,700 cycles = both loops (18,700) + the first wave's load phase +
            // the second wave's store phase + the barrier.  Letting the partners take turns on purpose (waves 0-3 compute
            // first and fetch / store afterwards, waves 4-7 fetch / store first) made a box 25,000 cycles: the younger wave's
            // fetch only starts when the older one's loop is over, and its latency is then exposed.
            compute(lds + cur * STAGE);
            __builtin_amdgcn_sched_barrier(0);
            WG_STAMP(1)
            if (has_next) store_box(lds + (cur ^ 1) * STAGE);
            __builtin_amdgcn_sched_barrier(0);
            WG_STAMP(2)
            __syncthreads();
            WG_STAMP(3)
            cur ^= 1;
        } else {
            compute(lds);
            __builtin_amdgcn_sched_barrier(0);
            if (has_next) {
                __syncthreads();
                store_box(lds);
                __syncthreads();
            }
        }
    }

#ifdef DRAM_WZY_STAMPS
    if (lane == 0)
        for (int q = 0; q < 5; ++q) atomicAdd(&g_wgrad_stamps[q], st_acc[q]);
#endif
#undef WG_STAMP
    // G^T p per accumulator element -> the three z taps of each (ky,kx) column
    const int ci = ci0 + wci * 16 + i;
    if (ci < a.Cin) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = co0 + wco * 16 + 4 * k + r;
            if (co < a.Cout) {
                float* o = a.slabs + (((size_t)sp * a.Cout + co) * a.Cin + ci) * 27;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const float p0 = acc[tap][0][r], p1 = acc[tap][1][r], p2 = acc[tap][2][r], p3 = acc[tap][3][r];
                    const float h = 0.5f * (p1 + p2);
                    o[tap] = p0 + h;
                    o[9 + tap] = 0.5f * (p1 - p2);
                    o[18 + tap] = h + p3;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// backward-weights of the FIRST layer (Cin == 1, e.g. DC3D ds_modules.0 conv 1->32).  The generic
// kernel would run its 16/32-wide input-channel tile 1/32 full; here the 27 taps of the single input
// channel play the role of the GEMM's N dimension instead:
//     dW[co][tap] = sum_v dY[co][v] * X[v + tap]        M = co (32), N = tap (27 of 32), K = voxels
// on v_mfma_f32_32x32x2_f32 (A[co][k] = dY[co][v+k], B[k][tap] = X[v+k+off(tap)], both from LDS).
// HBM-bound (reads dY once: 4*Cout B/voxel).  Block = 256 voxels (32x4x2 box) x 32 co, 4 waves x 64
// voxels; per-wave partial slabs, reduced in a fixed order by slab_reduce_kernel.
struct WgradC1Args {
    const float* x;   // [N][1][D][H][W]
    const float* dy;  // [N][Cout][D][H][W]
    float* slabs;     // [4*gridDim.x][Cout][27]
    int N, Cout, D, H, W;
    int nbx, nby, nbz, nboxes;
};

__global__ __launch_bounds__(256, 2) void conv3d_k3_wgrad_c1_kernel(WgradC1Args a) {
    constexpr int BX = 32, BY = 4, BZ = 2, VOX = 256;
    constexpr int HX = BX + 2, HY = BY + 2, HZ = BZ + 2, HV = HX * HY * HZ;   // 816
    constexpr int PA = VOX + 1;   // odd co stride: conflict-free A reads
    constexpr int NQ = (HV + 255) / 256;
    __shared__ float ldy[32 * PA];
    __shared__ float lx[HV];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int co0 = blockIdx.y * 32;
    const int D = a.D, H = a.H, W = a.W;
    const int S = D * H * W;
    const unsigned S4 = 4u * (unsigned)S;
    const int i = lane & 31, k = lane >> 5;     // A: co = i, voxel k ; B: tap = i, voxel k
    const int tap = i < 27 ? i : 0;
    const int tapoff = ((tap / 9) * HY + (tap / 3) % 3) * HX + tap % 3;
    const bool tap_ok = i < 27;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    for (int box = blockIdx.x; box < a.nboxes; box += gridDim.x) {
        int bb = box;
        const int bx = bb % a.nbx; bb /= a.nbx;
        const int by = bb % a.nby; bb /= a.nby;
        const int bz = bb % a.nbz;
        const int n = bb / a.nbz;
        const int x0 = bx * BX, y0 = by * BY, z0 = bz * BZ;
        __syncthreads();
        {   // dY[32][256]: thread = voxel, 32 channel rows (rows beyond Cout are out of the descriptor's range -> 0)
            const int vx = tid % BX, vy = (tid / BX) % BY, vz = tid / (BX * BY);
            const bool vok = (x0 + vx) < W && (y0 + vy) < H && (z0 + vz) < D;
            const __amdgpu_buffer_rsrc_t srd = make_rsrc(uniform_ptr(a.dy + (size_t)n * a.Cout * S), (unsigned)a.Cout * S4);
            unsigned run = vok ? 4u * (unsigned)(((z0 + vz) * H + y0 + vy) * W + x0 + vx) + (unsigned)co0 * S4 : OOB;
            const unsigned inc = vok ? S4 : 0u;
#pragma unroll 8
            for (int c = 0; c < 32; ++c) {
                ldy[c * PA + tid] = buf_load(srd, run, 0);
                run += inc;
            }
            const __amdgpu_buffer_rsrc_t srx = make_rsrc(uniform_ptr(a.x + (size_t)n * S), S4);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int e = tid + 256 * q;
                const int hx = e % HX, hy = (e / HX) % HY, hz = e / (HX * HY);
                const int gx = x0 - 1 + hx, gy = y0 - 1 + hy, gz = z0 - 1 + hz;
                const bool ok = e < HV && gx >= 0 && gx < W && gy >= 0 && gy < H && gz >= 0 && gz < D;
                const float v = buf_load(srx, ok ? 4u * (unsigned)((gz * H + gy) * W + gx) : OOB, 0);
                if (e < HV) lx[e] = v;
            }
        }
        __syncthreads();
        // wave w: voxels [64w, 64w+64) = 2 x-rows; 32 k-steps of 2 voxels
        const float* ap = ldy + i * PA + 64 * wave + k;
#pragma unroll 8
        for (int s = 0; s < 32; ++s) {
            const int v = 64 * wave + 2 * s + k;            // this lane's voxel of the k-step
            const int vx = v % BX, vy = (v / BX) % BY, vz = v / (BX * BY);
            const float av = ap[2 * s];
            float bv = lx[(vz * HY + vy) * HX + vx + tapoff];
            bv = tap_ok ? bv : 0.f;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
    }
    // D[row = co][col = tap]: lane col = lane&31, rows (r&3)+8(r>>2)+4k
    if (tap_ok) {
        float* slab = a.slabs + ((size_t)(blockIdx.x * 4 + wave) * a.Cout) * 27;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + (r & 3) + 8 * (r >> 2) + 4 * k;
            if (co < a.Cout) slab[(size_t)co * 27 + i] = acc[r];
        }
    }
}

#include "wgrad_wzy.inc"

static inline int wgrad_c1_blocks(int nboxes) { return nboxes < 1024 ? nboxes : 1024; }

// out[co][ci][tap] = sum over the first `split` slabs (input channels >= C1: over the first `split2` slabs -- backward-weights of a
// virtual concat run as one launch per source, each with its own split; C1 = Cin: one split)
__global__ void slab_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ out, int64_t E, int split, int Cin,
                                   int C1, int split2) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    int n = split;
    if (Cin > 0 && (int)((e / 27) % Cin) >= C1) n = split2;
    float s = 0.f;
    for (int p = 0; p < n; ++p) s += slabs[(size_t)p * E + e];
    out[e] = s;
}

__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ wt, int Cout, int Cin, int mode) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t E = (int64_t)27 * Cin * Cout;
    if (e >= E) return;
    if (mode == 0) {  // wt[t][ci][co] = w[co][ci][t]
        const int co = e % Cout;
        const int ci = (e / Cout) % Cin;
        const int t = e / ((int64_t)Cout * Cin);
        wt[e] = w[((size_t)co * Cin + ci) * 27 + t];
    } else {  // wt[t][co][ci] = w[co][ci][26-t]
        const int ci = e % Cin;
        const int co = (e / Cin) % Cout;
        const int t = e / ((int64_t)Cout * Cin);
        wt[e] = w[((size_t)co * Cin + ci) * 27 + (26 - t)];
    }
}

// Transformed filters of the Winograd-z kernel: wz[(ky*3+kx)*4 + xi][ci'][co'] = (G g)[xi] of the three z taps g of
// column (ky,kx); mode as above (1: the transposed convolution's filter, taps flipped, channel roles swapped).
__global__ void pack_weights_wz_kernel(const float* __restrict__ w, float* __restrict__ wz, int Cout, int Cin, int mode) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t E = (int64_t)36 * Cin * Cout;
    if (e >= E) return;
    const int t36 = (int)(e / ((int64_t)Cout * Cin));
    const int tap9 = t36 / 4, xi = t36 % 4;
    int co, ci;
    if (mode == 0) { co = e % Cout; ci = (e / Cout) % Cin; }
    else { ci = e % Cin; co = (e / Cin) % Cout; }
    const float* g = w + ((size_t)co * Cin + ci) * 27;
    float g0, g1, g2;
    if (mode == 0) { g0 = g[tap9]; g1 = g[9 + tap9]; g2 = g[18 + tap9]; }
    else { g0 = g[18 + (8 - tap9)]; g1 = g[9 + (8 - tap9)]; g2 = g[8 - tap9]; }
    wz[e] = xi == 0 ? g0 : xi == 1 ? 0.5f * ((g0 + g1) + g2) : xi == 2 ? 0.5f * ((g0 - g1) + g2) : g2;
}

// Transformed filters of the Winograd-(z,y) kernel: wzy[t = 16 kx + 4 xi_z + xi_y][chunk][kh][Ko][kk] =
// (G g[:,:,kx] G^T)[xi_z][xi_y] for GEMM input channel 4 chunk + 2 kk + kh (zero beyond Ki) and GEMM output channel
// 0..Ko-1; mode 0: Ko = Cout, Ki = Cin; mode 1 (backward-data): Ko = Cin, Ki = Cout, taps flipped.
__device__ __forceinline__ float wino_g(float g0, float g1, float g2, int xi) {
    return xi == 0 ? g0 : xi == 1 ? 0.5f * ((g0 + g1) + g2) : xi == 2 ? 0.5f * ((g0 - g1) + g2) : g2;
}
__global__ void pack_weights_wzy_kernel(const float* __restrict__ w, float* __restrict__ wzy, int Cout, int Cin, int mode) {
    const int Ko = mode == 0 ? Cout : Cin, Ki = mode == 0 ? Cin : Cout;
    const int nchunk = (Ki + 3) / 4;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t E = (int64_t)48 * nchunk * 4 * Ko;
    if (e >= E) return;
    const int kk = (int)(e % 2);
    const int ko = (int)((e / 2) % Ko);
    const int kh = (int)((e / (2 * (int64_t)Ko)) % 2);
    const int chunk = (int)((e / (4 * (int64_t)Ko)) % nchunk);
    const int t = (int)(e / (4 * (int64_t)Ko * nchunk));
    const int ki = 4 * chunk + 2 * kk + kh;
    if (ki >= Ki) { wzy[e] = 0.f; return; }
    const int kx = t / 16, xz = (t % 16) / 4, xy = t % 4;
    const float* g = mode == 0 ? w + ((size_t)ko * Cin + ki) * 27 : w + ((size_t)ki * Cin + ko) * 27;
    float gz[3];        // G along z for each ky
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        float g0, g1, g2;
        if (mode == 0) { g0 = g[ky * 3 + kx]; g1 = g[9 + ky * 3 + kx]; g2 = g[18 + ky * 3 + kx]; }
        else { const int f = (2 - ky) * 3 + (2 - kx); g0 = g[18 + f]; g1 = g[9 + f]; g2 = g[f]; }
        gz[ky] = wino_g(g0, g1, g2, xz);
    }
    wzy[e] = wino_g(gz[0], gz[1], gz[2], xy);
}

static inline size_t wzy_floats(int Ko, int Ki) { return (size_t)48 * ((Ki + 3) / 4) * 4 * Ko; }

// ---------------------------------------------------------------------------------------------
static int persistent_blocks() {       // one block per CU of the current device
    static int cus[DRAM_MAX_DEVICES] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= DRAM_MAX_DEVICES) dev = 0;
    if (cus[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[dev] = n;
    }
    return cus[dev];
}

template <int BX>
static int launch_fwd_wzy_f(ConvArgs& a, unsigned nblk, hipStream_t st) {
    using G = FwdWzyGeomT<BX>;
    const auto kern = BX == 32 ? conv3d_k3_fwd_wzy_kernel : conv3d_k3_fwd_wzy16_kernel;
    static LdsAttrOnce lds_once;
    if (const int rc = ensure_dynamic_lds((const void*)kern, G::LDS_BYTES, lds_once, "conv3d_k3_fwd(wzy)")) return rc;
    a.co_tiles = a.Cout / 64;
    const int64_t total = (int64_t)nblk * a.co_tiles;
    if (total > 0x7fffffffLL) {
        set_error("conv3d_k3_fwd: grid too large");
        return DRAM_EINVAL;
    }
    a.dv_d[0] = (unsigned)a.co_tiles; a.dv_d[1] = (unsigned)a.nbx; a.dv_d[2] = (unsigned)a.nby; a.dv_d[3] = (unsigned)a.nbz;
    for (int i = 0; i < 4; ++i) fast_div_prepare(a.dv_d[i], a.dv_m[i], a.dv_s[i]);
    const int64_t cus = persistent_blocks();
    const unsigned grid = (unsigned)(total < cus ? total : cus);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), G::LDS_BYTES, st, a, (int)total);
    return check_launch("conv3d_k3_fwd(wzy)");
}

template <int BX>
static int launch_fwd_wzy(ConvArgs& a, hipStream_t st) {
    a.nbx = cdiv(a.W, BX);
    a.nby = cdiv(a.H, 4);
    a.nbz = cdiv(a.D, FwdWzyGeomT<BX>::BZ);
    const int64_t nblk = (int64_t)a.N * a.nbx * a.nby * a.nbz;
    if (nblk > 0x7fffffffLL) {
        set_error("conv3d_k3_fwd: grid too large");
        return DRAM_EINVAL;
    }
    return launch_fwd_wzy_f<BX>(a, (unsigned)nblk, st);
}

// ---------------------------------------------------------------------------------------------
template <int BX, int BY, int COT, bool FUSED>
static int launch_fwd_wz_cot(ConvArgs& a, unsigned nblk, hipStream_t st) {
    using G = FwdWzGeom<BX, BY, COT>;
    static LdsAttrOnce lds_once;
    if (const int rc = ensure_dynamic_lds((const void*)conv3d_k3_fwd_wz_kernel<BX, BY, COT, FUSED>, G::LDS_BYTES, lds_once, "conv3d_k3_fwd(wz)")) return rc;
    a.co_tiles = cdiv(a.Cout, 32 * COT);
    const int64_t total = (int64_t)nblk * a.co_tiles;
    if (total > 0x7fffffffLL) {
        set_error("conv3d_k3_fwd: grid too large");
        return DRAM_EINVAL;
    }
    hipLaunchKernelGGL((conv3d_k3_fwd_wz_kernel<BX, BY, COT, FUSED>), dim3((unsigned)total), dim3(256), G::LDS_BYTES, st, a);
    return check_launch("conv3d_k3_fwd(wz)");
}

template <int BX, int BY>
static int launch_fwd_wz(ConvArgs& a, hipStream_t st) {
    a.nbx = cdiv(a.W, BX);
    a.nby = cdiv(a.H, BY);
    a.nbz = cdiv(a.D, 2);
    const int64_t nblk = (int64_t)a.N * a.nbx * a.nby * a.nbz;
    if (nblk > 0x7fffffffLL) {
        set_error("conv3d_k3_fwd: grid too large");
        return DRAM_EINVAL;
    }
    // the fused variant (operand transform on load / statistics epilogue) is a separate instantiation: the plain
    // kernel keeps its registers and schedule
    const bool fused = a.coef1 || a.coef2 || a.stats;
    if (a.Cout <= 32) return fused ? launch_fwd_wz_cot<BX, BY, 1, true>(a, (unsigned)nblk, st) : launch_fwd_wz_cot<BX, BY, 1, false>(a, (unsigned)nblk, st);
    return fused ? launch_fwd_wz_cot<BX, BY, 2, true>(a, (unsigned)nblk, st) : launch_fwd_wz_cot<BX, BY, 2, false>(a, (unsigned)nblk, st);
}

// ---------------------------------------------------------------------------------------------
template <int BX, int BY, int BZ, int COT, bool FUSED>
static int launch_fwd_cot(ConvArgs& a, unsigned nblk, hipStream_t st) {
    using G = FwdGeom<BX, BY, BZ, COT>;
    static LdsAttrOnce lds_once;
    if (const int rc = ensure_dynamic_lds((const void*)conv3d_k3_fwd_kernel<BX, BY, BZ, COT, FUSED>, G::LDS_BYTES, lds_once, "conv3d_k3_fwd")) return rc;
    a.co_tiles = cdiv(a.Cout, 32 * COT);
    const int64_t total = (int64_t)nblk * a.co_tiles;
    if (total > 0x7fffffffLL) {
        set_error("conv3d_k3_fwd: grid too large");
        return DRAM_EINVAL;
    }
    hipLaunchKernelGGL((conv3d_k3_fwd_kernel<BX, BY, BZ, COT, FUSED>), dim3((unsigned)total), dim3(256), G::LDS_BYTES, st, a);
    return check_launch("conv3d_k3_fwd");
}

template <int BX, int BY, int BZ>
static int launch_fwd(ConvArgs& a, hipStream_t st) {
    a.nbx = cdiv(a.W, BX);
    a.nby = cdiv(a.H, BY);
    a.nbz = cdiv(a.D, BZ);
    const int64_t nblk = (int64_t)a.N * a.nbx * a.nby * a.nbz;
    if (nblk > 0x7fffffffLL) {
        set_error("conv3d_k3_fwd: grid too large");
        return DRAM_EINVAL;
    }
    const bool fused = a.coef1 || a.coef2 || a.stats;
    if (a.Cout <= 32) return fused ? launch_fwd_cot<BX, BY, BZ, 1, true>(a, (unsigned)nblk, st) : launch_fwd_cot<BX, BY, BZ, 1, false>(a, (unsigned)nblk, st);
    return fused ? launch_fwd_cot<BX, BY, BZ, 2, true>(a, (unsigned)nblk, st) : launch_fwd_cot<BX, BY, BZ, 2, false>(a, (unsigned)nblk, st);
}

// Box shape (all 256 / 64 voxels): the one that pads the volume least; ties go to the first listed.  The reference trains and infers on 80^3 chunks (RESAMPLE_SIZE), whose
// pyramid 80/40/20/10 is covered exactly by 16-, 8-wide boxes but only to 83 % / 62 % by 32-wide ones.
static int pick_box(int D, int H, int W, const int (*boxes)[3], int nboxes, const char* env) {
    if (const char* f = getenv(env)) {          // experiments only
        for (int i = 0; i < nboxes; ++i)
            if (atoi(f) == boxes[i][0]) return i;
    }
    int best = 0;
    int64_t best_vol = -1;
    for (int i = 0; i < nboxes; ++i) {
        const int64_t vol = (int64_t)cdiv(W, boxes[i][0]) * boxes[i][0] * cdiv(H, boxes[i][1]) * boxes[i][1] *
                            cdiv(D, boxes[i][2]) * boxes[i][2];
        if (best_vol < 0 || vol < best_vol) { best = i; best_vol = vol; }
    }
    return best;
}

// box tables of the z-only Winograd kernel ((x, y) positions of a plane pair) and of the direct kernel
static const int kFwdWzBoxes[4][2] = {{32, 4}, {16, 8}, {8, 16}, {10, 10}};
static const int kFwdBoxes[3][3] = {{32, 4, 2}, {16, 4, 4}, {8, 8, 4}};

// The Winograd-z kernel serves every layer with enough input channels to amortise its filter tile; the first
// layer (Cin = 1) and DRAM_CONV_DIRECT=1 (experiments, A/B tests) use the direct kernel.
static bool use_wz(const ConvArgs& a) {
    static const bool direct = getenv("DRAM_CONV_DIRECT") != nullptr;
    return !direct && a.Cin >= 8 && a.D >= 2;
}

// Which forward kernel / box shape a shape gets (shared by the launch and by dram_conv3d_k3_stats_parts).
// The Winograd-(z,y) kernel: whole 64-channel output tiles, enough input channels, and a volume that one of its two boxes
// covers with little padding (it executes 2/3 of the z-only kernel's MFMAs: worth up to ~1.3x padding):
//   32 x 4 x 2 (128-byte rows: preferred) where it pads at most 1.2x and no more than the 16-wide box;
//   16 x 4 x 4 where it pads less than that, or at most 1.2x, or at most 1.3x of what the z-only kernel's own best box pads
//   (the 20^3 level of the reference's 80^3 chunks: 1.6 against 1.28 for 10x10 positions -- 38 against 46 MFMAs per output).
// Returns the box width (32 / 16) or 0.
static double wz_padding(int D, int H, int W);
static int wzy_box(const ConvArgs& a) {       // what the SHAPE allows (dram_conv3d_k3_stats_parts sizes the partials by it)
    const bool off = getenv("DRAM_CONV_NO_WZY") != nullptr;     // (read per call: A/B tests toggle it inside one process)
    if (off || !use_wz(a) || a.Cout % 64 != 0 || a.Cin < 8) return 0;
    if (a.dst.C2 > 0 && a.dst.C1 % 32 != 0) return 0;                           // a wave's 32 channels: one destination tensor
    const int64_t dmax = (int64_t)a.D * a.H * a.W > (int64_t)a.dst.D2 * a.dst.H2 * a.dst.W2 ? (int64_t)a.D * a.H * a.W
                                                                                              : (int64_t)a.dst.D2 * a.dst.H2 * a.dst.W2;
    if (dmax * 4 * 36 > 0xffffffffLL) return 0;                                 // ... behind one 32-bit descriptor
    // input rows are fetched as aligned 16-byte pieces (LDS-DMA) behind one descriptor per channel
    if (a.W % 4 != 0 || (int64_t)a.D * a.H * a.W * 4 >= (int64_t)OOB) return 0;
    const double vol = (double)a.W * a.H * a.D;
    const double pad32 = (double)cdiv(a.W, 32) * 32 * cdiv(a.H, 4) * 4 * cdiv(a.D, 2) * 2 / vol;
    const double pad16 = (double)cdiv(a.W, 16) * 16 * cdiv(a.H, 4) * 4 * cdiv(a.D, 4) * 4 / vol;
    if (const char* f = getenv("DRAM_WZY_BX")) {                                // experiments / A-B tests only (read per call)
        const int bx = atoi(f);
        if (bx == 32) return pad32 <= 1.2 ? 32 : 0;
        if (bx == 16) return 16;
    }
    if (pad32 <= 1.2 && pad32 <= pad16) return 32;
    if (pad16 <= 1.2 || pad16 <= 1.3 * wz_padding(a.D, a.H, a.W)) return 16;
    return 0;
}
// ... and what the SOURCE of one launch allows on top of that: 16-byte aligned bases, a cropped second source whose rows and
// window start on 16-byte boundaries.  A launch that fails only this runs the z-only kernel instead -- on the 32x4 positions of
// the 32-wide box, on 16x8 positions for the 16-wide one -- and the number of statistics partials the caller sizes its buffer
// for (dram_conv3d_k3_stats_parts: a function of the shape alone) covers both; unused slots are zero-filled by the launch.
static bool wzy_source_ok(const ConvArgs& a) {
    if ((((unsigned long long)a.src.p1) | ((unsigned long long)a.src.p2)) & 15ull) return false;
    if (a.src.p2 && (a.src.W2 % 4 != 0 || a.src.ox % 4 != 0 || (int64_t)a.src.D2 * a.src.H2 * a.src.W2 * 4 >= (int64_t)OOB)) return false;
    return true;
}

struct FwdChoice {
    bool c1;            // first-layer kernel (Cin = 1, plain source and destination)
    bool c1w;           // ... its wide form (a lane owns four consecutive x: W % 4 == 0, W >= 96)
    bool wz;
    bool wzy;
    int wzy_bx;         // box width of the (z,y) kernel (32 / 16)
    int box;            // index into the kernel family's box table
    int nbx, nby, nbz;  // boxes per sample
    int parts_per_box;  // statistics partials a box writes per row
    int parts_cap;      // partial slots per row the caller provides for this SHAPE (>= what any launch of it writes)
};
// padding of the z-only kernel's best box: lane slots spent per voxel
static int wz_best_box(int H, int W) {
    int best = 0;
    double best_cost = -1.0;
    for (int i = 0; i < 4; ++i) {
        const double cost = (double)cdiv(W, kFwdWzBoxes[i][0]) * cdiv(H, kFwdWzBoxes[i][1]) * 128.0;   // lane slots spent on a plane
        if (best_cost < 0 || cost < best_cost) { best = i; best_cost = cost; }
    }
    return best;
}
static double wz_padding(int D, int H, int W) {
    const int b = wz_best_box(H, W);
    return (double)cdiv(W, kFwdWzBoxes[b][0]) * cdiv(H, kFwdWzBoxes[b][1]) * 256.0 * cdiv(D, 2) / ((double)W * H * D);   // (a box: 128 positions x 2 planes)
}
static FwdChoice fwd_choice(const ConvArgs& a) {
    FwdChoice c;
    c.wz = use_wz(a);
    const int shape_bx = wzy_box(a);
    c.wzy = shape_bx != 0 && wzy_source_ok(a);
    c.wzy_bx = c.wzy ? shape_bx : 0;
    c.parts_per_box = 4;
    static const bool direct = getenv("DRAM_CONV_DIRECT") != nullptr;
    c.c1 = !direct && a.Cin == 1 && a.src.p2 == nullptr && a.dst.p2 == nullptr && a.coef1 == nullptr;
    c.c1w = c.c1 && a.W % 4 == 0 && a.W >= 96 && getenv("DRAM_C1_NARROW") == nullptr;
    if (c.c1w) {
        c.box = 0;
        c.parts_per_box = 16;           // (wave, row)
        c.nbx = cdiv(a.W, FwdC1WGeom::BX); c.nby = cdiv(a.H, FwdC1WGeom::BY); c.nbz = cdiv(a.D, FwdC1WGeom::BZ);
    } else if (c.c1) {
        c.box = 0;
        c.parts_per_box = 8;            // (wave, row group)
        c.nbx = cdiv(a.W, FwdC1Geom::BX); c.nby = cdiv(a.H, FwdC1Geom::BY); c.nbz = cdiv(a.D, FwdC1Geom::BZ);
    } else if (c.wzy) {
        c.box = 0;
        c.nbx = cdiv(a.W, c.wzy_bx); c.nby = cdiv(a.H, 4); c.nbz = cdiv(a.D, c.wzy_bx == 32 ? 2 : 4);
    } else if (c.wz) {
        // position boxes: the padded plane area, weighted by the lanes a box leaves idle (10x10 uses 100 of 128: the
        // 20^3 and 10^3 levels of the reference's 80^3 chunks fit it exactly)
        const int (*boxes2)[2] = kFwdWzBoxes;
        int best = wz_best_box(a.H, a.W);
        if (const char* f = getenv("DRAM_FWD_BX"))
            for (int i = 0; i < 4; ++i)
                if (atoi(f) == boxes2[i][0]) best = i;
        if (shape_bx == 32) best = 0;           // the (z,y) kernel's shape, refused for its source only (wzy_source_ok)
        if (shape_bx == 16) best = 1;
        c.box = best;
        c.nbx = cdiv(a.W, boxes2[best][0]); c.nby = cdiv(a.H, boxes2[best][1]); c.nbz = cdiv(a.D, 2);
    } else {
        const int (*boxes)[3] = kFwdBoxes;
        c.box = pick_box(a.D, a.H, a.W, boxes, 3, "DRAM_FWD_BX");
        c.nbx = cdiv(a.W, boxes[c.box][0]); c.nby = cdiv(a.H, boxes[c.box][1]); c.nbz = cdiv(a.D, boxes[c.box][2]);
    }
    const int64_t produced = (int64_t)c.nbx * c.nby * c.nbz * c.parts_per_box;
    int64_t cap = produced;
    if (shape_bx == 16 && !c.c1) {               // either kernel of a 16-wide (z,y) shape: (z,y) on 16x4x4, z-only on 16x8x2 boxes
        const int64_t zy = (int64_t)cdiv(a.W, 16) * cdiv(a.H, 4) * cdiv(a.D, 4) * 4, z = (int64_t)cdiv(a.W, 16) * cdiv(a.H, 8) * cdiv(a.D, 2) * 4;
        cap = zy > z ? zy : z;
    }
    c.parts_cap = cap > 0x7fffffffLL ? 0 : (int)cap;
    return c;
}

// What ran: launches per kernel family since the library was loaded (dram_conv3d_k3_launch_counts).  The tests of the
// benchmarked shapes assert on these, so that "the (z,y) kernel was verified" means the (z,y) kernel was launched.
static std::atomic<unsigned long long> g_launches[DRAM_K3_KINDS];

// kind + instantiation name (as rocprofv3 prints it, without namespace and argument list) of the forward /
// backward-data kernel that conv_fwd_dispatch launches for `a`
static int fwd_kernel_id(const ConvArgs& a, const FwdChoice& c, char* name, size_t cap) {
    const bool fused = a.coef1 || a.coef2 || a.stats;
    const int cot = a.Cout <= 32 ? 1 : 2;
    int kind;
    char buf[96];
    if (c.c1) {
        kind = DRAM_K3_FWD_C1;
        snprintf(buf, sizeof(buf), c.c1w ? "conv3d_k3_fwd_c1w_kernel" : "conv3d_k3_fwd_c1_kernel");
    } else if (c.wzy) {
        kind = DRAM_K3_FWD_WZY;
        snprintf(buf, sizeof(buf), c.wzy_bx == 32 ? "conv3d_k3_fwd_wzy_kernel" : "conv3d_k3_fwd_wzy16_kernel");
    } else if (c.wz) {
        kind = DRAM_K3_FWD_WZ;
        snprintf(buf, sizeof(buf), "conv3d_k3_fwd_wz_kernel<%d, %d, %d, %s>", kFwdWzBoxes[c.box][0], kFwdWzBoxes[c.box][1], cot,
                 fused ? "true" : "false");
    } else {
        kind = DRAM_K3_FWD_DIRECT;
        snprintf(buf, sizeof(buf), "conv3d_k3_fwd_kernel<%d, %d, %d, %d, %s>", kFwdBoxes[c.box][0], kFwdBoxes[c.box][1],
                 kFwdBoxes[c.box][2], cot, fused ? "true" : "false");
    }
    if (name && cap) snprintf(name, cap, "%s", buf);
    return kind;
}

static int conv_fwd_dispatch(ConvArgs& a, hipStream_t st) {
    const FwdChoice c = fwd_choice(a);
    g_launches[fwd_kernel_id(a, c, nullptr, 0)].fetch_add(1, std::memory_order_relaxed);
    if (a.stats) {
        // the caller's slots per row (dram_conv3d_k3_stats_parts) must hold what this launch writes; slots it leaves unused
        // (a 16-wide (z,y) shape whose two kernels box the volume differently) are zero-filled: count 0, ignored by the combine
        const int produced = c.nbx * c.nby * c.nbz * c.parts_per_box;
        DRAM_REQUIRE(a.nparts >= produced, "conv3d_k3_fwd: statistics buffer sized for %d partials per row, this shape produces %d "
                     "(dram_conv3d_k3_stats_parts)", a.nparts, produced);
        if (a.nparts > produced)
            (void)hipMemsetAsync(a.stats, 0, (size_t)a.N * a.Cout * a.nparts * 3 * sizeof(float), st);
    }
    if (c.c1) {
        a.nbx = c.nbx; a.nby = c.nby; a.nbz = c.nbz;
        a.co_tiles = cdiv(a.Cout, 32);
        const int64_t total = (int64_t)a.N * c.nbx * c.nby * c.nbz * a.co_tiles;
        if (total > 0x7fffffffLL) {
            set_error("conv3d_k3_fwd: grid too large");
            return DRAM_EINVAL;
        }
        if (c.c1w) hipLaunchKernelGGL(conv3d_k3_fwd_c1w_kernel, dim3((unsigned)total), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(conv3d_k3_fwd_c1_kernel, dim3((unsigned)total), dim3(256), 0, st, a);
        return check_launch("conv3d_k3_fwd(c1)");
    }
    if (c.wzy) {
        a.wt += (size_t)63 * a.Cin * a.Cout;      // ... and the (z,y)-transformed ones follow those
        return c.wzy_bx == 32 ? launch_fwd_wzy<32>(a, st) : launch_fwd_wzy<16>(a, st);
    }
    if (c.wz) {
        a.wt += (size_t)27 * a.Cin * a.Cout;      // the transformed filters follow the direct ones in the packed buffer
        switch (c.box) {
            case 0: return launch_fwd_wz<32, 4>(a, st);
            case 1: return launch_fwd_wz<16, 8>(a, st);
            case 2: return launch_fwd_wz<8, 16>(a, st);
            default: return launch_fwd_wz<10, 10>(a, st);
        }
    }
    switch (c.box) {
        case 0: return launch_fwd<32, 4, 2>(a, st);
        case 1: return launch_fwd<16, 4, 4>(a, st);
        default: return launch_fwd<8, 8, 4>(a, st);
    }
}

// Split of the (z,y) backward-weights kernel's box range over blocks: one 512-thread block per CU; contiguous box ranges, so the
// split only has to fill the device evenly (`tiles` = (ci tile, co tile) pairs of the launch, `slab` = bytes of one partial dW)
static int wzy_split(int tiles, int nboxes, size_t slab, size_t ws_cap = (size_t)1 << 30) {
    const int resident = 256;
    int best = 1;
    double best_score = -1.0;
    for (int sp = 1; sp <= nboxes && sp <= 4096; ++sp) {
        const int64_t blocks = (int64_t)tiles * sp;
        if (blocks > 4LL * resident && sp > 1) break;
        if ((size_t)sp * slab > ws_cap && sp > 1) break;
        const int64_t rounds = (blocks + resident - 1) / resident;
        double util = (double)blocks / (double)(rounds * resident);
        const int64_t per = (nboxes + sp - 1) / sp;
        util *= (double)nboxes / (double)(per * sp);
        const double score = util + 1e-3 * (blocks >= 2LL * resident ? 1.0 : (double)blocks / (2.0 * resident));
        if (score > best_score) { best_score = score; best = sp; }
    }
    return best;
}

constexpr size_t kWgradSubWsCap = (size_t)256 << 20;      // partial slabs of a per-source launch of a virtual concat

struct WgradPlan {
    int variant;  // 0: block = 64 co x 32 ci (Cout <= 64);  1: block = 128 co x 16 ci
    int wz;       // 1: Winograd-z kernel (boxes = bx x by positions of a plane pair)
    int wzy;      // 1: Winograd-(z,y) kernel (boxes = 16 x positions of a y pair and z pair); cit = 16-channel ci tiles per block
    int cit;
    int bx, by, bz, nbx, nby, nbz, nboxes, ci_tiles, co_tiles, split;
};

static WgradPlan wgrad_plan(int N, int Cin, int Cout, int D, int H, int W, int C1 = 0) {
    WgradPlan p;
    p.variant = Cout > 64 ? 1 : 0;
    p.wz = 0;
    p.wzy = 0;
    p.cit = 1;
    {   // Winograd-(z,y): full 16-wide boxes along x, whole y and z pairs, a ci tile inside one source tensor
        static const bool direct = getenv("DRAM_CONV_DIRECT") != nullptr;
        const bool off = getenv("DRAM_WGRAD_NO_WZY") != nullptr;      // (read per call: A/B tests toggle it inside one process)
        const int cit = 1;
        // 16-wide boxes along x, the last one of a row ragged where W % 16 != 0 (16-byte pieces: W % 4 == 0) as long as it pads at
        // most 1.6x (measured at the reference's 40^3 level, 1.2x padding: 233 TFLOP/s direct-equivalent against 186 for the
        // z-only kernel on exact boxes; the 20^3 level pads 1.6x: ~175 against the z-only kernel's 151 there)
        const bool wide_ok = W % 4 == 0 && cdiv(W, 16) * 16 * 10 <= W * 16;
        if (!direct && !off && wide_ok && H % 2 == 0 && D % 2 == 0 && D >= 4 && (C1 == 0 || C1 % (16 * cit) == 0)) {   // (D >= 4: two boxes per z column, the raw plane ring counts on it)
            p.wzy = 1;
            p.cit = cit;
            p.bx = 16; p.by = 2; p.bz = 2;
            p.nbx = cdiv(W, 16); p.nby = H / 2; p.nbz = D / 2;
            p.nboxes = (int)((int64_t)N * p.nbx * p.nby * p.nbz);
            p.ci_tiles = cdiv(Cin, 16 * cit);
            p.co_tiles = cdiv(Cout, 64);
            p.split = wzy_split(p.ci_tiles * p.co_tiles, p.nboxes, (size_t)Cout * Cin * 27 * sizeof(float));
            return p;
        }
    }
    {   // Winograd-z: rows must be full boxes along x and a channel tile must lie inside one source tensor.  Its 64 co x
        // 32 ci tile (two LDS stages, one barrier per box) serves wide layers too: measured 197-200 TFLOP/s direct-equivalent
        // against 186-190 for the 128 co x 16 ci tile at 384->128, 256->256, 768->256; the latter remains for a concat
        // boundary that is a multiple of 16 only.
        static const bool direct = getenv("DRAM_CONV_DIRECT") != nullptr;
        const int bx = (W % 16 == 0) ? 16 : ((W % 8 == 0) ? 8 : ((W % 4 == 0) ? 4 : 0));
        int variant = -1;
        if (C1 == 0 || C1 % 32 == 0) variant = 0;
        else if (Cout > 64 && C1 % 16 == 0) variant = 1;
        if (const char* f = getenv("DRAM_WGRAD_VARIANT")) {                              // experiments only
            const int v = atoi(f) ? 1 : 0;
            if (C1 == 0 || C1 % (v == 1 ? 16 : 32) == 0) variant = v;
        }
        if (!direct && bx && D >= 2 && variant >= 0) {
            p.wz = 1;
            p.variant = variant;
            p.bx = bx; p.by = 32 / bx; p.bz = 2;
        }
    }
    if (!p.wz) {
    // ties go to the 16-wide box: its X halo (18x4x4 = 288 elements per 64 voxels, against 408 for 34x4x3)
    // needs the fewest staging loads (measured +2.6 % at 128^3 / 64^3 / 32^3)
    static const int boxes[3][3] = {{16, 2, 2}, {32, 2, 1}, {8, 4, 2}};
    const int bi = pick_box(D, H, W, boxes, 3, "DRAM_WGRAD_BX");
    p.bx = boxes[bi][0]; p.by = boxes[bi][1]; p.bz = boxes[bi][2];
    }
    p.nbx = cdiv(W, p.bx); p.nby = cdiv(H, p.by); p.nbz = cdiv(D, p.bz);
    const int64_t nb = (int64_t)N * p.nbx * p.nby * p.nbz;
    p.nboxes = (int)nb;
    p.ci_tiles = cdiv(Cin, p.variant == 0 ? 32 : 16);
    p.co_tiles = cdiv(Cout, p.variant == 1 ? 128 : 64);
    // Blocks = tiles x split, one 512-thread block resident per CU (256 at a time).  A last round
    // that is nearly empty costs a whole block duration (measured: 1032 blocks on 512 slots ran 1.5x
    // longer than 1020), so pick the split with the best slot utilisation, preferring >= 2 rounds.
    const int tiles = p.ci_tiles * p.co_tiles;
    const int resident = 256;
    const size_t slab = (size_t)Cout * Cin * 27 * sizeof(float);
    int best = 1;
    double best_score = -1.0;
    for (int sp = 1; sp <= p.nboxes && sp <= 4096; ++sp) {
        const int64_t blocks = (int64_t)tiles * sp;
        if (blocks > 4LL * resident && sp > 1) break;
        if ((size_t)sp * slab > ((size_t)1 << 30) && sp > 1) break;
        const int64_t rounds = (blocks + resident - 1) / resident;
        double util = (double)blocks / (double)(rounds * resident);
        const int64_t per = (p.nboxes + sp - 1) / sp;   // uneven box counts per block also idle slots
        util *= (double)p.nboxes / (double)(per * sp);
        const double score = util + 1e-3 * (blocks >= 2LL * resident ? 1.0 : (double)blocks / (2.0 * resident));
        if (score > best_score) { best_score = score; best = sp; }
    }
    p.split = best;
    return p;
}

template <int BX, int BY, int BZ, int COS, int CIT>
static int launch_wgrad_vec(WgradArgs& a, hipStream_t st) {
    using G = WgradVecGeom<BX, BY, BZ, COS, CIT>;
    static LdsAttrOnce lds_once;
    if (const int rc = ensure_dynamic_lds((const void*)conv3d_k3_wgrad_vec_kernel<BX, BY, BZ, COS, CIT>, G::LDS_BYTES, lds_once, "conv3d_k3_wgrad")) return rc;
    const unsigned grid = (unsigned)(a.split * a.ci_tiles * a.co_tiles);
    hipLaunchKernelGGL((conv3d_k3_wgrad_vec_kernel<BX, BY, BZ, COS, CIT>), dim3(grid), dim3(512), G::LDS_BYTES, st, a);
    return check_launch("conv3d_k3_wgrad(vec)");
}

template <int BX, int BY, int COS, int CIT, bool LAZY>
static int launch_wgrad_wz_l(WgradArgs& a, hipStream_t st) {
    using G = WgradWzGeom<BX, BY, COS, CIT>;
    static LdsAttrOnce lds_once;
    if (const int rc = ensure_dynamic_lds((const void*)conv3d_k3_wgrad_wz_kernel<BX, BY, COS, CIT, LAZY>, G::LDS_BYTES, lds_once, "conv3d_k3_wgrad")) return rc;
    const unsigned grid = (unsigned)(a.split * a.ci_tiles * a.co_tiles);
    hipLaunchKernelGGL((conv3d_k3_wgrad_wz_kernel<BX, BY, COS, CIT, LAZY>), dim3(grid), dim3(G::T), G::LDS_BYTES, st, a);
    return check_launch("conv3d_k3_wgrad(wz)");
}
template <int BX, int BY, int COS, int CIT>
static int launch_wgrad_wz(WgradArgs& a, hipStream_t st) {
    // the lazy-operand variant is a separate instantiation: the plain kernel keeps its registers and schedule
    return (a.coef1 || a.coef2) ? launch_wgrad_wz_l<BX, BY, COS, CIT, true>(a, st) : launch_wgrad_wz_l<BX, BY, COS, CIT, false>(a, st);
}

template <bool LAZY>
static int launch_wgrad_wzy_l(WgradArgs& a, hipStream_t st) {
    using G = WgradWzyGeom;
    static LdsAttrOnce lds_once;
    if (const int rc = ensure_dynamic_lds((const void*)conv3d_k3_wgrad_wzy_kernel<LAZY>, G::LDS_BYTES, lds_once, "conv3d_k3_wgrad(wzy)")) return rc;
    const unsigned grid = (unsigned)(a.split * a.ci_tiles * a.co_tiles);
    const int per = cdiv(a.nboxes, a.split);
    hipLaunchKernelGGL((conv3d_k3_wgrad_wzy_kernel<LAZY>), dim3(grid), dim3(512), G::LDS_BYTES, st, a, per);
    return check_launch("conv3d_k3_wgrad(wzy)");
}
static int launch_wgrad_wzy(WgradArgs& a, hipStream_t st) {
    return (a.coef1 || a.coef2) ? launch_wgrad_wzy_l<true>(a, st) : launch_wgrad_wzy_l<false>(a, st);
}

template <int BX, int BY, int BZ, int COS, int CIT>
static int launch_wgrad(WgradArgs& a, hipStream_t st) {
    using G = WgradGeom<BX, BY, BZ, COS, CIT>;
    static LdsAttrOnce lds_once;
    if (const int rc = ensure_dynamic_lds((const void*)conv3d_k3_wgrad_kernel<BX, BY, BZ, COS, CIT>, G::LDS_BYTES, lds_once, "conv3d_k3_wgrad")) return rc;
    const unsigned grid = (unsigned)(a.split * a.ci_tiles * a.co_tiles);
    hipLaunchKernelGGL((conv3d_k3_wgrad_kernel<BX, BY, BZ, COS, CIT>), dim3(grid), dim3(512), G::LDS_BYTES, st, a);
    return check_launch("conv3d_k3_wgrad");
}

// The backward-weights kernel wgrad_run launches for a plan: kind + the instantiation's template arguments.
struct WgradKernel {
    int kind;               // DRAM_K3_WGRAD_*
    int bx, by, bz, cos, cit;
    bool lazy;
};
static WgradKernel wgrad_kernel(const WgradPlan& p, int C1, bool has_x2, int W, bool lazy) {
    WgradKernel k;
    k.bx = p.bx; k.by = p.by; k.bz = p.bz;
    k.cos = p.variant == 1 ? 8 : 4;
    k.cit = p.variant == 1 ? 1 : 2;
    k.lazy = false;
    if (p.wzy) {
        k.kind = DRAM_K3_WGRAD_WZY;
        k.cit = p.cit;
        k.lazy = lazy;
        return k;
    }
    if (p.wz) {
        k.kind = lazy ? DRAM_K3_WGRAD_WZ_LAZY : DRAM_K3_WGRAD_WZ;
        k.lazy = lazy;
        return k;
    }
    // 16-byte staging needs full boxes along x and a channel tile that lies inside one source tensor
    const int ci_b = 16 * k.cit;
    const bool vec = (W % p.bx == 0) && (!has_x2 || C1 % ci_b == 0) && getenv("DRAM_WGRAD_NOVEC") == nullptr;
    k.kind = vec ? DRAM_K3_WGRAD_VEC : DRAM_K3_WGRAD_DIRECT;
    return k;
}
static void wgrad_kernel_name(const WgradKernel& k, char* name, size_t cap) {
    if (!name || !cap) return;
    if (k.kind == DRAM_K3_WGRAD_C1) snprintf(name, cap, "conv3d_k3_wgrad_c1_kernel");
    else if (k.kind == DRAM_K3_WGRAD_WZY) snprintf(name, cap, "conv3d_k3_wgrad_wzy_kernel<%s>", k.lazy ? "true" : "false");
    else if (k.kind == DRAM_K3_WGRAD_WZ || k.kind == DRAM_K3_WGRAD_WZ_LAZY)
        snprintf(name, cap, "conv3d_k3_wgrad_wz_kernel<%d, %d, %d, %d, %s>", k.bx, k.by, k.cos, k.cit, k.lazy ? "true" : "false");
    else
        snprintf(name, cap, "conv3d_k3_%s_kernel<%d, %d, %d, %d, %d>", k.kind == DRAM_K3_WGRAD_VEC ? "wgrad_vec" : "wgrad", k.bx, k.by,
                 k.bz, k.cos, k.cit);
}

static int check_conv_shape(const char* who, int N, int Cin, int Cout, int D, int H, int W) {
    DRAM_REQUIRE(N > 0 && Cin > 0 && Cout > 0 && D > 0 && H > 0 && W > 0, "%s: non-positive dimension", who);
    DRAM_REQUIRE((int64_t)D * H * W * (int64_t)(Cin > Cout ? Cin : Cout) < 0x1fffffffLL,
                 "%s: one sample exceeds 2^29 elements (2 GiB; 32-bit buffer offsets with an out-of-range sentinel)", who);
    return DRAM_OK;
}

static int check_cat(const char* who, const CatView& v, int D, int H, int W) {
    if (v.p2 == nullptr) return DRAM_OK;
    DRAM_REQUIRE(v.C2 > 0 && v.oz >= 0 && v.oy >= 0 && v.ox >= 0 && v.oz + D <= v.D2 && v.oy + H <= v.H2 &&
                     v.ox + W <= v.W2,
                 "%s: crop window (%d,%d,%d)+(%d,%d,%d) outside the second tensor (%d,%d,%d)", who, v.oz, v.oy, v.ox,
                 D, H, W, v.D2, v.H2, v.W2);
    return DRAM_OK;
}

}  // namespace dram

using namespace dram;

#ifdef DRAM_WZY_STAMPS
extern "C" int dram_debug_wzy_stamps(unsigned long long* out, int reset) {
    unsigned long long z[32] = {};
    if (reset) return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_wzy_stamps), z, sizeof(z));
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wzy_stamps), sizeof(z));
}
extern "C" int dram_debug_wgrad_stamps(unsigned long long* out, int reset) {
    unsigned long long z[8] = {};
    if (reset) return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_wgrad_stamps), z, sizeof(z));
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wgrad_stamps), sizeof(z));
}
#endif

extern "C" size_t dram_conv3d_k3_packed_floats(int Cout, int Cin) {
    if (Cout <= 0 || Cin <= 0) return 0;
    const size_t a = wzy_floats(Cout, Cin), b = wzy_floats(Cin, Cout);     // forward / backward-data orientation
    return (size_t)(27 + 36) * Cin * Cout + (a > b ? a : b);
}

extern "C" int dram_conv3d_k3_pack_weights(const float* w, float* wt, int Cout, int Cin, int mode, void* stream) {
    DRAM_REQUIRE(w && wt, "conv3d_k3_pack_weights: null pointer");
    DRAM_REQUIRE(Cout > 0 && Cin > 0 && (mode == 0 || mode == 1), "conv3d_k3_pack_weights: bad arguments");
    const int64_t E = (int64_t)27 * Cin * Cout;
    hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)cdiv64(E, 256)), dim3(256), 0, (hipStream_t)stream, w, wt,
                       Cout, Cin, mode);
    const int64_t Ez = (int64_t)36 * Cin * Cout;
    hipLaunchKernelGGL(pack_weights_wz_kernel, dim3((unsigned)cdiv64(Ez, 256)), dim3(256), 0, (hipStream_t)stream, w,
                       wt + E, Cout, Cin, mode);
    const int64_t Ezy = (int64_t)(mode == 0 ? wzy_floats(Cout, Cin) : wzy_floats(Cin, Cout));
    hipLaunchKernelGGL(pack_weights_wzy_kernel, dim3((unsigned)cdiv64(Ezy, 256)), dim3(256), 0, (hipStream_t)stream, w,
                       wt + E + Ez, Cout, Cin, mode);
    return check_launch("conv3d_k3_pack_weights");
}

static int conv_fwd_fill(ConvArgs& a, const float* x1, int C1, const float* x2, int C2, int D2, int H2, int W2, int oz, int oy,
                         int ox, const float* wt, const float* bias, float* y1, int Co1, float* y2, int Co2, int yD2, int yH2,
                         int yW2, int yoz, int yoy, int yox, int N, int D, int H, int W) {
    DRAM_REQUIRE(x1 && wt && y1, "conv3d_k3_fwd: null pointer");
    DRAM_REQUIRE(C1 > 0 && Co1 > 0 && (x2 == nullptr || C2 > 0) && (y2 == nullptr || Co2 > 0),
                 "conv3d_k3_fwd: bad channel counts");
    a = ConvArgs{};
    a.src = CatView{const_cast<float*>(x1), const_cast<float*>(x2), C1, x2 ? C2 : 0, x2 ? D2 : 1, x2 ? H2 : 1,
                    x2 ? W2 : 1, x2 ? oz : 0, x2 ? oy : 0, x2 ? ox : 0};
    a.dst = CatView{y1, y2, Co1, y2 ? Co2 : 0, y2 ? yD2 : 1, y2 ? yH2 : 1, y2 ? yW2 : 1, y2 ? yoz : 0, y2 ? yoy : 0,
                    y2 ? yox : 0};
    a.wt = wt;
    a.bias = bias;
    a.N = N;
    a.Cin = a.src.C1 + a.src.C2;
    a.Cout = a.dst.C1 + a.dst.C2;
    a.D = D; a.H = H; a.W = W;
    int rc = check_conv_shape("conv3d_k3_fwd", N, a.Cin, a.Cout, D, H, W);
    if (rc) return rc;
    if ((rc = check_cat("conv3d_k3_fwd(src)", a.src, D, H, W))) return rc;
    if ((rc = check_cat("conv3d_k3_fwd(dst)", a.dst, D, H, W))) return rc;
    return DRAM_OK;
}

// Generic entry used by the Python side for forward (dst plain or plain) and
// backward-data (src plain, dst possibly split into two tensors).
extern "C" int dram_conv3d_k3_fwd_ex(const float* x1, int C1, const float* x2, int C2, int D2, int H2, int W2, int oz,
                                     int oy, int ox, const float* wt, const float* bias, float* y1, int Co1, float* y2,
                                     int Co2, int yD2, int yH2, int yW2, int yoz, int yoy, int yox, int N, int D,
                                     int H, int W, void* stream) {
    ConvArgs a;
    const int rc = conv_fwd_fill(a, x1, C1, x2, C2, D2, H2, W2, oz, oy, ox, wt, bias, y1, Co1, y2, Co2, yD2, yH2, yW2, yoz,
                                 yoy, yox, N, D, H, W);
    if (rc) return rc;
    return conv_fwd_dispatch(a, (hipStream_t)stream);
}

extern "C" int dram_conv3d_k3_stats_parts(int Cin, int Cout, int D, int H, int W) {
    if (Cin <= 0 || Cout <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
    ConvArgs a = {};
    a.Cin = Cin; a.Cout = Cout; a.D = D; a.H = H; a.W = W;
    return fwd_choice(a).parts_cap;
}

// Which kernel a forward / backward-data call of this shape launches (the same fwd_choice the launch uses).  The plain query
// assumes a source the (z,y) kernel accepts (16-byte aligned, no cropped second tensor); dram_conv3d_k3_fwd_choice_src takes
// the facts of the source that can send a launch to the z-only kernel instead (wzy_source_ok).
extern "C" int dram_conv3d_k3_fwd_choice_src(int Cin, int Cout, int D, int H, int W, int dstC1, int dstC2, int dstD2, int dstH2,
                                             int dstW2, int fused, int srcC2, int srcD2, int srcH2, int srcW2, int srcox,
                                             int src_misaligned, char* name, size_t cap) {
    DRAM_REQUIRE(Cin > 0 && Cout > 0 && D > 0 && H > 0 && W > 0, "conv3d_k3_fwd_choice: non-positive dimension");
    DRAM_REQUIRE(dstC2 == 0 || dstC1 + dstC2 == Cout, "conv3d_k3_fwd_choice: the destination split does not add up to Cout");
    DRAM_REQUIRE(srcC2 >= 0 && srcC2 < Cin, "conv3d_k3_fwd_choice: the second source holds %d of %d input channels", srcC2, Cin);
    ConvArgs a = {};
    a.Cin = Cin; a.Cout = Cout; a.D = D; a.H = H; a.W = W;
    a.dst.C1 = dstC2 > 0 ? dstC1 : Cout;
    a.dst.C2 = dstC2 > 0 ? dstC2 : 0;
    a.dst.D2 = dstC2 > 0 ? dstD2 : 1; a.dst.H2 = dstC2 > 0 ? dstH2 : 1; a.dst.W2 = dstC2 > 0 ? dstW2 : 1;
    static const float dummy[8] = {};
    // (pointers are only tested for null-ness / alignment by the choice: a 16-byte aligned dummy, or one 4 bytes off)
    const float* al = (const float*)(((unsigned long long)dummy + 15ull) & ~15ull);
    a.src.p1 = const_cast<float*>(src_misaligned ? al + 1 : al);
    a.src.C1 = Cin - srcC2;
    if (srcC2 > 0) {
        a.src.p2 = const_cast<float*>(al);
        a.src.C2 = srcC2; a.src.D2 = srcD2; a.src.H2 = srcH2; a.src.W2 = srcW2; a.src.ox = srcox;
    }
    if (fused) a.stats = const_cast<float*>(dummy);          // (only tested for null-ness by fwd_kernel_id)
    return fwd_kernel_id(a, fwd_choice(a), name, cap);
}
extern "C" int dram_conv3d_k3_fwd_choice(int Cin, int Cout, int D, int H, int W, int dstC1, int dstC2, int dstD2, int dstH2,
                                         int dstW2, int fused, char* name, size_t cap) {
    return dram_conv3d_k3_fwd_choice_src(Cin, Cout, D, H, W, dstC1, dstC2, dstD2, dstH2, dstW2, fused, 0, 0, 0, 0, 0, 0, name, cap);
}

// Which kernel a backward-weights call of this shape launches (the same wgrad_plan / wgrad_kernel the launch uses).
extern "C" int dram_conv3d_k3_wgrad_choice(int N, int C1, int C2, int Cout, int D, int H, int W, int lazy, char* name, size_t cap) {
    DRAM_REQUIRE(N > 0 && C1 > 0 && C2 >= 0 && Cout > 0 && D > 0 && H > 0 && W > 0, "conv3d_k3_wgrad_choice: bad dimension");
    WgradKernel k = {};
    if (C1 + C2 == 1) {
        k.kind = DRAM_K3_WGRAD_C1;
    } else {
        const WgradPlan p = wgrad_plan(N, C1 + C2, Cout, D, H, W, C2 > 0 ? C1 : 0);
        k = wgrad_kernel(p, C1, C2 > 0, W, lazy != 0 && (p.wz || p.wzy));
    }
    wgrad_kernel_name(k, name, cap);
    return k.kind;
}

extern "C" int dram_conv3d_k3_launch_counts(unsigned long long* counts, int n) {
    DRAM_REQUIRE(counts && n > 0, "conv3d_k3_launch_counts: null pointer");
    for (int i = 0; i < n; ++i) counts[i] = i < DRAM_K3_KINDS ? g_launches[i].load(std::memory_order_relaxed) : 0ull;
    return DRAM_OK;
}

// Fused forward: each source may be the RAW output of the previous conv with its norm (+ReLU) applied on load
// (coefK = per-row {a, b}, null = plain tensor), and the statistics of the output are accumulated in the epilogue.
extern "C" int dram_conv3d_k3_fwd_fused(const float* x1, int C1, const float* coef1, int relu1, const float* x2, int C2,
                                        const float* coef2, int relu2, int D2, int H2, int W2, int oz, int oy, int ox,
                                        const float* wt, const float* bias, float* y, float* stats, int nparts, int N,
                                        int Cout, int D, int H, int W, void* stream) {
    ConvArgs a;
    const int rc = conv_fwd_fill(a, x1, C1, x2, C2, D2, H2, W2, oz, oy, ox, wt, bias, y, Cout, nullptr, 0, 0, 0, 0, 0, 0, 0, N,
                                 D, H, W);
    if (rc) return rc;
    DRAM_REQUIRE(coef2 == nullptr || x2 != nullptr, "conv3d_k3_fwd_fused: coef2 without a second source");
    DRAM_REQUIRE(stats == nullptr || nparts > 0, "conv3d_k3_fwd_fused: statistics buffer without a partial count");
    DRAM_REQUIRE(stats == nullptr || bias == nullptr, "conv3d_k3_fwd_fused: output statistics and a bias exclude each other "
                 "(a convolution that feeds a norm layer has no bias, reference models.py:78)");
    a.coef1 = coef1; a.coef2 = coef2;
    a.relu1 = relu1; a.relu2 = relu2;
    a.stats = stats; a.nparts = nparts;
    return conv_fwd_dispatch(a, (hipStream_t)stream);
}

extern "C" int dram_conv3d_k3_fwd(const float* x, const float* wt, const float* bias, float* y, int N, int Cin,
                                  int Cout, int D, int H, int W, void* stream) {
    return dram_conv3d_k3_fwd_ex(x, Cin, nullptr, 0, 0, 0, 0, 0, 0, 0, wt, bias, y, Cout, nullptr, 0, 0, 0, 0, 0, 0, 0,
                                 N, D, H, W, stream);
}

extern "C" int dram_conv3d_k3_fwd_cat(const float* x1, int C1, const float* x2, int C2, int D2, int H2, int W2, int oz,
                                      int oy, int ox, const float* wt, const float* bias, float* y, int N, int Cout,
                                      int D, int H, int W, void* stream) {
    DRAM_REQUIRE(x2 != nullptr, "conv3d_k3_fwd_cat: second tensor is null");
    return dram_conv3d_k3_fwd_ex(x1, C1, x2, C2, D2, H2, W2, oz, oy, ox, wt, bias, y, Cout, nullptr, 0, 0, 0, 0, 0, 0,
                                 0, N, D, H, W, stream);
}

extern "C" size_t dram_conv3d_k3_wgrad_ws_bytes(int N, int Cin, int Cout, int D, int H, int W) {
    if (N <= 0 || Cin <= 0 || Cout <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
    if (Cin == 1) {
        const int64_t nboxes = (int64_t)N * cdiv(W, 32) * cdiv(H, 4) * cdiv(D, 2);
        return (size_t)4 * wgrad_c1_blocks((int)(nboxes < 0x7fffffff ? nboxes : 0x7fffffff)) * Cout * 27 * sizeof(float);
    }
    // the split depends on whether a virtual-concat input lets the Winograd kernel run: size for the larger -- and, where the
    // (z,y) kernel runs, for a per-source launch of a concat with one lazy source (fewer tiles, a finer split: wgrad_run)
    const WgradPlan p = wgrad_plan(N, Cin, Cout, D, H, W, 0), q = wgrad_plan(N, Cin, Cout, D, H, W, 1);
    const size_t slab = (size_t)Cout * Cin * 27 * sizeof(float);
    int split = p.split > q.split ? p.split : q.split;
    if (p.wzy && Cin > 16) {
        const int sub = wzy_split(p.co_tiles, p.nboxes, slab, kWgradSubWsCap);
        split = sub > split ? sub : split;
    }
    return (size_t)split * slab;
}

// 1 if backward-weights of this shape can take its x operand lazily (normalise + ReLU on load): the Winograd kernel
// runs (W % 4 == 0, D >= 2, channel tiles inside one source tensor) and it is not the first-layer kernel.
extern "C" int dram_conv3d_k3_wgrad_lazy_ok(int N, int C1, int C2, int Cout, int D, int H, int W) {
    if (N <= 0 || C1 <= 0 || C2 < 0 || Cout <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
    if (C1 + C2 == 1) return 0;
    const WgradPlan p = wgrad_plan(N, C1 + C2, Cout, D, H, W, C2 > 0 ? C1 : 0);
    return p.wz | p.wzy;
}

static int wgrad_run(const float* x1, int C1, const float* coef1, int relu1, const float* x2, int C2, const float* coef2,
                     int relu2, int D2, int H2, int W2, int oz, int oy, int ox, const float* dy, float* dw, void* ws,
                     size_t ws_bytes, int N, int Cout, int D, int H, int W, void* stream) {
    DRAM_REQUIRE(x1 && dy && dw && ws, "conv3d_k3_wgrad: null pointer");
    WgradArgs a = {};
    a.coef1 = coef1; a.coef2 = x2 ? coef2 : nullptr;
    a.relu1 = relu1; a.relu2 = relu2;
    a.src = CatView{const_cast<float*>(x1), const_cast<float*>(x2), C1, x2 ? C2 : 0, x2 ? D2 : 1, x2 ? H2 : 1,
                    x2 ? W2 : 1, x2 ? oz : 0, x2 ? oy : 0, x2 ? ox : 0};
    a.dy = dy;
    a.slabs = (float*)ws;
    a.N = N;
    a.Cin = a.src.C1 + a.src.C2;
    a.Cout = Cout;
    a.D = D; a.H = H; a.W = W;
    int rc = check_conv_shape("conv3d_k3_wgrad", N, a.Cin, Cout, D, H, W);
    if (rc) return rc;
    if ((rc = check_cat("conv3d_k3_wgrad(src)", a.src, D, H, W))) return rc;
    if (a.Cin == 1 && x2 == nullptr) {   // first layer: dedicated kernel
        DRAM_REQUIRE(a.coef1 == nullptr, "conv3d_k3_wgrad_fused: the first-layer kernel (Cin = 1) takes a plain input");
        DRAM_REQUIRE(((int64_t)Cout + 32) * (int64_t)D * H * W < 0x3fffffffLL,
                     "conv3d_k3_wgrad: (channels + 32) * voxels per sample exceeds 2^30 (32-bit buffer offsets)");
        WgradC1Args c;
        c.x = x1; c.dy = dy; c.slabs = (float*)ws;
        c.N = N; c.Cout = Cout; c.D = D; c.H = H; c.W = W;
        c.nbx = cdiv(W, 32); c.nby = cdiv(H, 4); c.nbz = cdiv(D, 2);
        const int64_t nb = (int64_t)N * c.nbx * c.nby * c.nbz;
        DRAM_REQUIRE(nb < 0x7fffffffLL, "conv3d_k3_wgrad: too many boxes");
        c.nboxes = (int)nb;
        const int blocks = wgrad_c1_blocks(c.nboxes);
        const size_t need1 = (size_t)4 * blocks * Cout * 27 * sizeof(float);
        if (ws_bytes < need1) {
            set_error("conv3d_k3_wgrad: workspace %zu < %zu bytes", ws_bytes, need1);
            return DRAM_EWS;
        }
        hipStream_t st1 = (hipStream_t)stream;
        g_launches[DRAM_K3_WGRAD_C1].fetch_add(1, std::memory_order_relaxed);
        hipLaunchKernelGGL(conv3d_k3_wgrad_c1_kernel, dim3(blocks, cdiv(Cout, 32)), dim3(256), 0, st1, c);
        const int64_t E1 = (int64_t)Cout * 27;
        hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)cdiv64(E1, 256)), dim3(256), 0, st1, c.slabs, dw, E1, 4 * blocks, 0, 0, 0);
        return check_launch("conv3d_k3_wgrad(c1)");
    }
    // running 32-bit offsets walk up to 128 channel planes past the last one: they must not wrap
    DRAM_REQUIRE(((int64_t)(a.Cin > Cout ? a.Cin : Cout) + 128) * (int64_t)D * H * W < 0x3fffffffLL,
                 "conv3d_k3_wgrad: (channels + 128) * voxels per sample exceeds 2^30 (32-bit buffer offsets)");
    const WgradPlan p = wgrad_plan(N, a.Cin, Cout, D, H, W, x2 ? C1 : 0);
    DRAM_REQUIRE(p.wz || p.wzy || (a.coef1 == nullptr && a.coef2 == nullptr),
                 "conv3d_k3_wgrad_fused: this shape runs a kernel without the lazy-operand path "
                 "(dram_conv3d_k3_wgrad_lazy_ok): materialise x first");
    const size_t need = (size_t)p.split * Cout * a.Cin * 27 * sizeof(float);
    if (ws_bytes < need) {
        set_error("conv3d_k3_wgrad: workspace %zu < %zu bytes", ws_bytes, need);
        return DRAM_EWS;
    }
    a.nbx = p.nbx; a.nby = p.nby; a.nbz = p.nbz; a.nboxes = p.nboxes;
    a.split = p.split; a.ci_tiles = p.ci_tiles; a.co_tiles = p.co_tiles;
    hipStream_t st = (hipStream_t)stream;
    const WgradKernel wk = wgrad_kernel(p, C1, x2 != nullptr, W, a.coef1 || a.coef2);
    g_launches[wk.kind].fetch_add(1, std::memory_order_relaxed);
    const bool vec = wk.kind == DRAM_K3_WGRAD_VEC;
    const int64_t E = (int64_t)Cout * a.Cin * 27;
    if (p.wzy && x2 && (a.coef1 != nullptr) != (a.coef2 != nullptr) && getenv("DRAM_WGRAD_ONE_LAUNCH") == nullptr) {
        // A virtual concat with exactly ONE lazy source (the first conv of an UpsampleConvBlock5d in the fused engine: the
        // upsampled part plain, the skip part lazy): one launch per source, each on its own instantiation and with its own
        // split.  In one launch the lazy tiles' blocks run ~9 % longer per box than the plain tiles', the blocks of a split
        // drift apart and stop sharing dY in L2: measured [4,128+64->64,128^3] 23.3 ms as one launch (all-lazy 22.0, plain 20.7).
        // The launches write disjoint input-channel columns of the same partial slabs; the reduce takes each column's split.
        const int t1 = a.src.C1 / 16, t2 = cdiv(a.src.C2, 16);
        const size_t cap = ws_bytes < kWgradSubWsCap ? ws_bytes : kWgradSubWsCap;   // (ws_bytes >= need: checked above)
        const int s1 = wzy_split(t1 * p.co_tiles, p.nboxes, (size_t)E * sizeof(float), cap);
        const int s2 = wzy_split(t2 * p.co_tiles, p.nboxes, (size_t)E * sizeof(float), cap);
        WgradArgs b = a;
        a.ci_tile0 = 0; a.ci_tiles = t1; a.split = s1; a.coef2 = nullptr;
        b.ci_tile0 = t1; b.ci_tiles = t2; b.split = s2; b.coef1 = nullptr;
        g_launches[wk.kind].fetch_add(1, std::memory_order_relaxed);
        rc = launch_wgrad_wzy(a, st);
        if (rc) return rc;
        rc = launch_wgrad_wzy(b, st);
        if (rc) return rc;
        hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)cdiv64(E, 256)), dim3(256), 0, st, a.slabs, dw, E, s1, a.Cin, a.src.C1, s2);
        return check_launch("conv3d_k3_wgrad(reduce)");
    }
    if (p.wzy) {
        rc = launch_wgrad_wzy(a, st);
    } else if (p.wz) {
        if (p.variant == 1)
            rc = p.bx == 16 ? launch_wgrad_wz<16, 2, 8, 1>(a, st) : p.bx == 8 ? launch_wgrad_wz<8, 4, 8, 1>(a, st) : launch_wgrad_wz<4, 8, 8, 1>(a, st);
        else
            rc = p.bx == 16 ? launch_wgrad_wz<16, 2, 4, 2>(a, st) : p.bx == 8 ? launch_wgrad_wz<8, 4, 4, 2>(a, st) : launch_wgrad_wz<4, 8, 4, 2>(a, st);
    } else if (vec) {
        if (p.variant == 1) {
            if (p.bx == 32) rc = launch_wgrad_vec<32, 2, 1, 8, 1>(a, st);
            else if (p.bx == 16) rc = launch_wgrad_vec<16, 2, 2, 8, 1>(a, st);
            else rc = launch_wgrad_vec<8, 4, 2, 8, 1>(a, st);
        } else {
            if (p.bx == 32) rc = launch_wgrad_vec<32, 2, 1, 4, 2>(a, st);
            else if (p.bx == 16) rc = launch_wgrad_vec<16, 2, 2, 4, 2>(a, st);
            else rc = launch_wgrad_vec<8, 4, 2, 4, 2>(a, st);
        }
    } else if (p.variant == 1) {
        if (p.bx == 32) rc = launch_wgrad<32, 2, 1, 8, 1>(a, st);
        else if (p.bx == 16) rc = launch_wgrad<16, 2, 2, 8, 1>(a, st);
        else rc = launch_wgrad<8, 4, 2, 8, 1>(a, st);
    } else {
        if (p.bx == 32) rc = launch_wgrad<32, 2, 1, 4, 2>(a, st);
        else if (p.bx == 16) rc = launch_wgrad<16, 2, 2, 4, 2>(a, st);
        else rc = launch_wgrad<8, 4, 2, 4, 2>(a, st);
    }
    if (rc) return rc;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)cdiv64(E, 256)), dim3(256), 0, st, a.slabs, dw, E, p.split, 0, 0, 0);
    return check_launch("conv3d_k3_wgrad(reduce)");
}

extern "C" int dram_conv3d_k3_wgrad_ex(const float* x1, int C1, const float* x2, int C2, int D2, int H2, int W2,
                                       int oz, int oy, int ox, const float* dy, float* dw, void* ws, size_t ws_bytes,
                                       int N, int Cout, int D, int H, int W, void* stream) {
    return wgrad_run(x1, C1, nullptr, 0, x2, C2, nullptr, 0, D2, H2, W2, oz, oy, ox, dy, dw, ws, ws_bytes, N, Cout, D, H, W,
                     stream);
}

// Backward-weights whose x operand is act(coef * raw + ...) applied on load (see dram_conv3d_k3_fwd_fused).
extern "C" int dram_conv3d_k3_wgrad_fused(const float* x1, int C1, const float* coef1, int relu1, const float* x2, int C2,
                                          const float* coef2, int relu2, int D2, int H2, int W2, int oz, int oy, int ox,
                                          const float* dy, float* dw, void* ws, size_t ws_bytes, int N, int Cout, int D,
                                          int H, int W, void* stream) {
    return wgrad_run(x1, C1, coef1, relu1, x2, C2, coef2, relu2, D2, H2, W2, oz, oy, ox, dy, dw, ws, ws_bytes, N, Cout, D, H, W,
                     stream);
}

extern "C" int dram_conv3d_k3_wgrad(const float* x, const float* dy, float* dw, void* ws, size_t ws_bytes, int N,
                                    int Cin, int Cout, int D, int H, int W, void* stream) {
    return dram_conv3d_k3_wgrad_ex(x, Cin, nullptr, 0, 0, 0, 0, 0, 0, 0, dy, dw, ws, ws_bytes, N, Cout, D, H, W,
                                   stream);
}
