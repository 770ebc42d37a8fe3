// 3x3x3 / stride 1 / pad 1 convolution for fp32 NCDHW tensors on gfx950.
//
// Replaces the ATen conv3d dispatch (and its two backward kernels) issued by the
// nn.Conv3d modules of reference dram/parts.py:95,105,133,142,177,185.
//
// Design (MI355X): every 3x3x3 conv of DC3D with Cin >= 32 is FP32-FLOP bound
// (200-2500 FLOP/B), so all three kernels are implicit GEMMs on the
// exact-fp32 matrix cores (v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32,
// 157 TFLOP/s peak = the fp32 vector peak but one VGPR per operand and the VALU
// left free).  Operand tiles are staged in LDS straight from NCDHW memory
// (rows along x are contiguous -> coalesced), zero padding is materialised in
// the LDS halo, and the 27 taps become compile-time LDS offsets.
//
//   forward / backward-data  D[co][voxel] += W[co][ci] * X[ci][voxel+tap]
//       block = 256 voxels (BX x BY x BZ box) x 32*COT output channels, 4 waves,
//       each wave 2 voxel tiles x COT channel tiles of 32x32 (fp32 accumulators),
//       K loop over channel chunks of 4 with all 27 taps unrolled.
//       The voxel index sits on the MFMA column (lane) axis so that every
//       accumulator register is a 128-byte run along x of one output channel.
//   backward-weights         dW[co][ci][tap] += dY[co][voxel] * X[ci][voxel+tap]
//       block = 64 co x 16 ci x all 27 taps (27 independent 16x16 accumulators
//       per wave), K loop over 128-voxel boxes, partial slabs + ordered reduce
//       (deterministic, no float atomics).
//
// The input of forward and the output of backward-data may be a *virtual*
// channel concatenation of two tensors (crop_concat_5d fused away).
#include "common.h"

namespace dram {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

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
    int nbx, nby, nbz;
};

constexpr int KC = 4;  // input channels per LDS stage

template <int BX, int BY, int BZ, int COT>
__global__ __launch_bounds__(256) void conv3d_k3_fwd_kernel(ConvArgs a) {
    static_assert(BX * BY * BZ == 256, "block covers 256 voxels");
    constexpr int HX = BX + 2, HY = BY + 2, HZ = BZ + 2;
    constexpr int HV = HX * HY * HZ;
    constexpr int PS = HV;
    constexpr int COB = 32 * COT;
    constexpr int NQ = (HV + 255) / 256;
    constexpr int WROWS = 27 * KC;
    constexpr int RPT = 256 / COB;  // weight rows staged per pass
    constexpr int WPASS = (WROWS + RPT - 1) / RPT;

    __shared__ float lds[KC * PS + WROWS * COB];
    float* lin = lds;
    float* lw = lds + KC * PS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int b = blockIdx.x;
    const int bx = b % a.nbx; b /= a.nbx;
    const int by = b % a.nby; b /= a.nby;
    const int bz = b % a.nbz;
    const int n = b / a.nbz;
    const int x0 = bx * BX, y0 = by * BY, z0 = bz * BZ;
    const int co0 = blockIdx.y * COB;
    const int D = a.D, H = a.H, W = a.W;
    const int S = D * H * W;
    const int S2 = a.src.D2 * a.src.H2 * a.src.W2;

    // per-thread halo elements of the input stage
    int off1[NQ], off2[NQ];
    bool ok[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int e = tid + 256 * q;
        const int hx = e % HX, hy = (e / HX) % HY, hz = e / (HX * HY);
        const int gx = x0 - 1 + hx, gy = y0 - 1 + hy, gz = z0 - 1 + hz;
        ok[q] = (e < HV) && gx >= 0 && gx < W && gy >= 0 && gy < H && gz >= 0 && gz < D;
        off1[q] = (gz * H + gy) * W + gx;
        off2[q] = ((gz + a.src.oz) * a.src.H2 + gy + a.src.oy) * a.src.W2 + gx + a.src.ox;
    }
    const float* s1 = a.src.p1 + (size_t)n * a.src.C1 * S;
    const float* s2 = a.src.p2 ? a.src.p2 + (size_t)n * a.src.C2 * S2 : nullptr;

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

    const int wrow = tid / COB, wcol = tid % COB;
    const bool wcol_ok = (co0 + wcol) < a.Cout;

    for (int c0 = 0; c0 < a.Cin; c0 += KC) {
        __syncthreads();
        // ---- stage KC input channels (zero padded halo) ----
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            const int ci = c0 + kc;
            float* dstp = lin + kc * PS;
            // (two statically indexed copies: a runtime select between off1/off2 would push them to scratch)
            if (ci >= a.Cin) {
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    if (tid + 256 * q < HV) dstp[tid + 256 * q] = 0.f;
            } else if (ci < a.src.C1) {
                const float* base = s1 + (size_t)ci * S;
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    if (tid + 256 * q < HV) dstp[tid + 256 * q] = ok[q] ? base[off1[q]] : 0.f;
            } else {
                const float* base = s2 + (size_t)(ci - a.src.C1) * S2;
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    if (tid + 256 * q < HV) dstp[tid + 256 * q] = ok[q] ? base[off2[q]] : 0.f;
            }
        }
        // ---- stage the weights of those channels: rows (tap,kc) x COB columns ----
#pragma unroll
        for (int p = 0; p < WPASS; ++p) {
            const int row = p * RPT + wrow;
            if (row < WROWS) {
                const int tap = row / KC, kc = row % KC;
                const int ci = c0 + kc;
                float v = 0.f;
                if (ci < a.Cin && wcol_ok) v = a.wt[((size_t)tap * a.Cin + ci) * a.Cout + co0 + wcol];
                lw[row * COB + wcol] = v;
            }
        }
        __syncthreads();
        // ---- 27 taps x KC/2 k-steps of 32x32x2 MFMAs ----
#pragma unroll
        for (int tap = 0; tap < 27; ++tap) {
            const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
            const int toff = (dz * HY + dy) * HX + dx;
#pragma unroll
            for (int kk = 0; kk < KC / 2; ++kk) {
                float av[COT], bv[2];
#pragma unroll
                for (int c = 0; c < COT; ++c) av[c] = lw[abase + (tap * KC + 2 * kk) * COB + 32 * c];
#pragma unroll
                for (int t = 0; t < 2; ++t) bv[t] = lin[bbase[t] + 2 * kk * PS + toff];
#pragma unroll
                for (int c = 0; c < COT; ++c)
#pragma unroll
                    for (int t = 0; t < 2; ++t)
                        acc[c][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c], bv[t], acc[c][t], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: accumulator register r of lane (j,kh) = channel (r&3)+8(r>>2)+4kh, voxel j ----
    const int dS2 = a.dst.D2 * a.dst.H2 * a.dst.W2;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int v = 32 * (2 * wave + t) + j;
        const int vx = v % BX, vy = (v / BX) % BY, vz = v / (BX * BY);
        const int gx = x0 + vx, gy = y0 + vy, gz = z0 + vz;
        if (gx >= W || gy >= H || gz >= D) continue;
        const int sp1 = (gz * H + gy) * W + gx;
        const int sp2 = ((gz + a.dst.oz) * a.dst.H2 + gy + a.dst.oy) * a.dst.W2 + gx + a.dst.ox;
#pragma unroll
        for (int c = 0; c < COT; ++c) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + 32 * c + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (co < a.Cout) {
                    float val = acc[c][t][r];
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

// ---------------------------------------------------------------------------------------------
struct WgradArgs {
    CatView src;      // x (possibly a virtual concatenation)
    const float* dy;  // [N][Cout][D][H][W]
    float* slabs;     // [SPLIT][Cout][Cin][27]
    int N, Cin, Cout, D, H, W;
    int nbx, nby, nbz, nboxes, split, ci_tiles, co_tiles;
};

template <int HVv>
struct PadTo2Mod32 {
    static constexpr int value = HVv + ((2 - (HVv % 32)) + 32) % 32;
};

template <int BX, int BY, int BZ>
__global__ __launch_bounds__(256) void conv3d_k3_wgrad_kernel(WgradArgs a) {
    constexpr int VOX = BX * BY * BZ;
    static_assert(VOX == 128 && BX % 4 == 0, "box = 128 voxels, rows a multiple of 4");
    constexpr int PA = VOX + 2;  // co stride of the dY tile: == 2 (mod 32) -> conflict-free A reads
    constexpr int HX = BX + 2, HY = BY + 2, HZ = BZ + 2;
    constexpr int HV = HX * HY * HZ;
    constexpr int PB = PadTo2Mod32<HV>::value;  // ci stride of the X halo tile
    constexpr int NQ = (HV + 255) / 256;
    constexpr int CO_B = 64, CI_B = 16;

    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* ldy = lds;             // [64][PA]
    float* lx = lds + CO_B * PA;  // [16][PB]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int b = blockIdx.x;
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
    // dY staging role: voxel v_st, channel rows co_st + 2q
    const int v_st = tid % VOX, co_st = tid / VOX;
    const int svx = v_st % BX, svy = (v_st / BX) % BY, svz = v_st / (BX * BY);

    for (int box = sp; box < a.nboxes; box += a.split) {
        int bb = box;
        const int bx = bb % a.nbx; bb /= a.nbx;
        const int by = bb % a.nby; bb /= a.nby;
        const int bz = bb % a.nbz;
        const int n = bb / a.nbz;
        const int x0 = bx * BX, y0 = by * BY, z0 = bz * BZ;

        __syncthreads();
        // ---- stage dY[64 co][128 voxels] ----
        {
            const int gx = x0 + svx, gy = y0 + svy, gz = z0 + svz;
            const bool vok = gx < W && gy < H && gz < D;
            const float* dyn = a.dy + ((size_t)n * a.Cout + co0) * S + (gz * H + gy) * W + gx;
#pragma unroll 8
            for (int q = 0; q < CO_B / 2; ++q) {
                const int co = co_st + 2 * q;
                float v = 0.f;
                if (vok && (co0 + co) < a.Cout) v = dyn[(size_t)co * S];
                ldy[co * PA + v_st] = v;
            }
        }
        // ---- stage X[16 ci][halo] ----
        {
            int off1[NQ], off2[NQ];
            bool ok[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int e = tid + 256 * q;
                const int hx = e % HX, hy = (e / HX) % HY, hz = e / (HX * HY);
                const int gx = x0 - 1 + hx, gy = y0 - 1 + hy, gz = z0 - 1 + hz;
                ok[q] = (e < HV) && gx >= 0 && gx < W && gy >= 0 && gy < H && gz >= 0 && gz < D;
                off1[q] = (gz * H + gy) * W + gx;
                off2[q] = ((gz + a.src.oz) * a.src.H2 + gy + a.src.oy) * a.src.W2 + gx + a.src.ox;
            }
            const float* s1 = a.src.p1 + (size_t)n * a.src.C1 * S;
            const float* s2 = a.src.p2 ? a.src.p2 + (size_t)n * a.src.C2 * S2 : nullptr;
#pragma unroll 4
            for (int c = 0; c < CI_B; ++c) {
                const int ci = ci0 + c;
                float* dstp = lx + c * PB;
                if (ci >= a.Cin) {
#pragma unroll
                    for (int q = 0; q < NQ; ++q)
                        if (tid + 256 * q < HV) dstp[tid + 256 * q] = 0.f;
                } else if (ci < a.src.C1) {
                    const float* base = s1 + (size_t)ci * S;
#pragma unroll
                    for (int q = 0; q < NQ; ++q)
                        if (tid + 256 * q < HV) dstp[tid + 256 * q] = ok[q] ? base[off1[q]] : 0.f;
                } else {
                    const float* base = s2 + (size_t)(ci - a.src.C1) * S2;
#pragma unroll
                    for (int q = 0; q < NQ; ++q)
                        if (tid + 256 * q < HV) dstp[tid + 256 * q] = ok[q] ? base[off2[q]] : 0.f;
                }
            }
        }
        __syncthreads();
        // ---- 32 k-steps (4 voxels along x each) x 27 taps of 16x16x4 MFMAs ----
        const float* ap = ldy + (wave * 16 + i) * PA + k;
        const float* bp = lx + i * PB + k;
#pragma unroll 2
        for (int ks = 0; ks < VOX / 4; ++ks) {
            const int x4 = ks % (BX / 4), vy = (ks / (BX / 4)) % BY, vz = ks / ((BX / 4) * BY);
            const float av = ap[(vz * BY + vy) * BX + 4 * x4];
            const float* bq = bp + (vz * HY + vy) * HX + 4 * x4;
#pragma unroll
            for (int tap = 0; tap < 27; ++tap) {
                const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
                const float bv = bq[(dz * HY + dy) * HX + dx];
                acc[tap] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[tap], 0, 0, 0);
            }
        }
    }

    // ---- partial slab: D row = co (4*(lane>>4)+r), col = ci (lane&15) ----
    const int ci = ci0 + i;
    if (ci < a.Cin) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = co0 + wave * 16 + 4 * k + r;
            if (co < a.Cout) {
                float* o = a.slabs + (((size_t)sp * a.Cout + co) * a.Cin + ci) * 27;
#pragma unroll
                for (int tap = 0; tap < 27; ++tap) o[tap] = acc[tap][r];
            }
        }
    }
}

__global__ void slab_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ out, int64_t E, int split) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    float s = 0.f;
    for (int p = 0; p < split; ++p) s += slabs[(size_t)p * E + e];
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

// ---------------------------------------------------------------------------------------------
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
    if (a.Cout <= 32) {
        dim3 grid((unsigned)nblk, cdiv(a.Cout, 32));
        hipLaunchKernelGGL((conv3d_k3_fwd_kernel<BX, BY, BZ, 1>), grid, dim3(256), 0, st, a);
    } else {
        dim3 grid((unsigned)nblk, cdiv(a.Cout, 64));
        hipLaunchKernelGGL((conv3d_k3_fwd_kernel<BX, BY, BZ, 2>), grid, dim3(256), 0, st, a);
    }
    return check_launch("conv3d_k3_fwd");
}

static int conv_fwd_dispatch(ConvArgs& a, hipStream_t st) {
    if (a.W >= 24) return launch_fwd<32, 4, 2>(a, st);
    if (a.W >= 12) return launch_fwd<16, 4, 4>(a, st);
    return launch_fwd<8, 8, 4>(a, st);
}

struct WgradPlan {
    int bx, by, bz, nbx, nby, nbz, nboxes, ci_tiles, co_tiles, split;
};

static WgradPlan wgrad_plan(int N, int Cin, int Cout, int D, int H, int W) {
    WgradPlan p;
    if (W >= 24) { p.bx = 32; p.by = 2; p.bz = 2; }
    else if (W >= 12) { p.bx = 16; p.by = 4; p.bz = 2; }
    else { p.bx = 8; p.by = 4; p.bz = 4; }
    p.nbx = cdiv(W, p.bx); p.nby = cdiv(H, p.by); p.nbz = cdiv(D, p.bz);
    const int64_t nb = (int64_t)N * p.nbx * p.nby * p.nbz;
    p.nboxes = (int)nb;
    p.ci_tiles = cdiv(Cin, 16);
    p.co_tiles = cdiv(Cout, 64);
    const int tiles = p.ci_tiles * p.co_tiles;
    int split = cdiv(1024, tiles);  // ~4 blocks per CU in flight over the whole launch
    if (split > p.nboxes) split = p.nboxes;
    if (split < 1) split = 1;
    p.split = split;
    return p;
}

template <int BX, int BY, int BZ>
static int launch_wgrad(WgradArgs& a, hipStream_t st) {
    constexpr int VOX = BX * BY * BZ;
    constexpr int HV = (BX + 2) * (BY + 2) * (BZ + 2);
    constexpr int PB = PadTo2Mod32<HV>::value;
    constexpr size_t lds_bytes = (size_t)(64 * (VOX + 2) + 16 * PB) * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)conv3d_k3_wgrad_kernel<BX, BY, BZ>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) {
            set_error("conv3d_k3_wgrad: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return DRAM_EHIP;
        }
        attr_done = true;
    }
    const unsigned grid = (unsigned)(a.split * a.ci_tiles * a.co_tiles);
    hipLaunchKernelGGL((conv3d_k3_wgrad_kernel<BX, BY, BZ>), dim3(grid), dim3(256), lds_bytes, st, a);
    return check_launch("conv3d_k3_wgrad");
}

static int check_conv_shape(const char* who, int N, int Cin, int Cout, int D, int H, int W) {
    DRAM_REQUIRE(N > 0 && Cin > 0 && Cout > 0 && D > 0 && H > 0 && W > 0, "%s: non-positive dimension", who);
    DRAM_REQUIRE((int64_t)D * H * W * (int64_t)(Cin > Cout ? Cin : Cout) < 0x7fffffffLL,
                 "%s: one sample exceeds 2^31 elements", who);
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

extern "C" int dram_conv3d_k3_pack_weights(const float* w, float* wt, int Cout, int Cin, int mode, void* stream) {
    DRAM_REQUIRE(w && wt, "conv3d_k3_pack_weights: null pointer");
    DRAM_REQUIRE(Cout > 0 && Cin > 0 && (mode == 0 || mode == 1), "conv3d_k3_pack_weights: bad arguments");
    const int64_t E = (int64_t)27 * Cin * Cout;
    hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)cdiv64(E, 256)), dim3(256), 0, (hipStream_t)stream, w, wt,
                       Cout, Cin, mode);
    return check_launch("conv3d_k3_pack_weights");
}

// Generic entry used by the Python side for forward (dst plain or plain) and
// backward-data (src plain, dst possibly split into two tensors).
extern "C" int dram_conv3d_k3_fwd_ex(const float* x1, int C1, const float* x2, int C2, int D2, int H2, int W2, int oz,
                                     int oy, int ox, const float* wt, const float* bias, float* y1, int Co1, float* y2,
                                     int Co2, int yD2, int yH2, int yW2, int yoz, int yoy, int yox, int N, int D,
                                     int H, int W, void* stream) {
    DRAM_REQUIRE(x1 && wt && y1, "conv3d_k3_fwd: null pointer");
    DRAM_REQUIRE(C1 > 0 && Co1 > 0 && (x2 == nullptr || C2 > 0) && (y2 == nullptr || Co2 > 0),
                 "conv3d_k3_fwd: bad channel counts");
    ConvArgs a;
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
    const WgradPlan p = wgrad_plan(N, Cin, Cout, D, H, W);
    return (size_t)p.split * Cout * Cin * 27 * sizeof(float);
}

extern "C" int dram_conv3d_k3_wgrad_ex(const float* x1, int C1, const float* x2, int C2, int D2, int H2, int W2,
                                       int oz, int oy, int ox, const float* dy, float* dw, void* ws, size_t ws_bytes,
                                       int N, int Cout, int D, int H, int W, void* stream) {
    DRAM_REQUIRE(x1 && dy && dw && ws, "conv3d_k3_wgrad: null pointer");
    WgradArgs a;
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
    const WgradPlan p = wgrad_plan(N, a.Cin, Cout, D, H, W);
    const size_t need = (size_t)p.split * Cout * a.Cin * 27 * sizeof(float);
    if (ws_bytes < need) {
        set_error("conv3d_k3_wgrad: workspace %zu < %zu bytes", ws_bytes, need);
        return DRAM_EWS;
    }
    a.nbx = p.nbx; a.nby = p.nby; a.nbz = p.nbz; a.nboxes = p.nboxes;
    a.split = p.split; a.ci_tiles = p.ci_tiles; a.co_tiles = p.co_tiles;
    hipStream_t st = (hipStream_t)stream;
    if (p.bx == 32) rc = launch_wgrad<32, 2, 2>(a, st);
    else if (p.bx == 16) rc = launch_wgrad<16, 4, 2>(a, st);
    else rc = launch_wgrad<8, 4, 4>(a, st);
    if (rc) return rc;
    const int64_t E = (int64_t)Cout * a.Cin * 27;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)cdiv64(E, 256)), dim3(256), 0, st, a.slabs, dw, E, p.split);
    return check_launch("conv3d_k3_wgrad(reduce)");
}

extern "C" int dram_conv3d_k3_wgrad(const float* x, const float* dy, float* dw, void* ws, size_t ws_bytes, int N,
                                    int Cin, int Cout, int D, int H, int W, void* stream) {
    return dram_conv3d_k3_wgrad_ex(x, Cin, nullptr, 0, 0, 0, 0, 0, 0, 0, dy, dw, ws, ws_bytes, N, Cout, D, H, W,
                                   stream);
}
