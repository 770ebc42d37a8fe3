// The less-travelled branches of the reference's factories, gfx950:
//   * nn.PReLU(num_parameters, init)           act_wrapper "prelu", dram/parts.py:51-52
//   * F.adaptive_max_pool3d(x, 1)              pooling_dense_features 'global_max', dram/models.py:41-42
// Both are HBM-bound streaming passes over (n,c) rows of S = D*H*W contiguous floats; every reduction is a
// fixed-order tree (deterministic, no atomics).
#include "common.h"

namespace dram {

constexpr int ACHUNK = 8192;

__global__ __launch_bounds__(256) void prelu_fwd_kernel(const float* __restrict__ x, const float* __restrict__ a,
                                                        float* __restrict__ y, int C, int nparam, int64_t S) {
    const int row = blockIdx.y;                       // n * C + c
    const float slope = a[nparam == 1 ? 0 : row % C];
    const float* xr = x + (int64_t)row * S;
    float* yr = y + (int64_t)row * S;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < S; e += stride) {
        const float v = xr[e];
        yr[e] = v > 0.f ? v : slope * v;
    }
}

// dx = dy * (x > 0 ? 1 : a);  part[row][chunk] = sum over the chunk of dy * x [x <= 0]
__global__ __launch_bounds__(256) void prelu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                        const float* __restrict__ a, float* __restrict__ dx,
                                                        float* __restrict__ part, int C, int nparam, int64_t S, int nchunks) {
    __shared__ float red[4];
    const int row = blockIdx.y, chunk = blockIdx.x;
    const float slope = a[nparam == 1 ? 0 : row % C];
    const int64_t beg = (int64_t)chunk * ACHUNK;
    const int len = (int)((S - beg) < ACHUNK ? (S - beg) : ACHUNK);
    const float* xr = x + (int64_t)row * S + beg;
    const float* gr = dy + (int64_t)row * S + beg;
    float* dr = dx ? dx + (int64_t)row * S + beg : nullptr;
    float acc = 0.f;
    for (int e = threadIdx.x; e < len; e += 256) {
        const float v = xr[e], g = gr[e];
        if (dr) dr[e] = v > 0.f ? g : slope * g;
        if (!(v > 0.f)) acc = fmaf(g, v, acc);
    }
    acc = block_sum_256(acc, red);
    if (threadIdx.x == 0) part[(size_t)row * nchunks + chunk] = acc;
}

// da[p] = sum over the rows of parameter p (all rows when nparam == 1), fixed order, fp64
__global__ __launch_bounds__(256) void prelu_da_kernel(const float* __restrict__ part, float* __restrict__ da, int N, int C,
                                                       int nparam, int nchunks) {
    __shared__ double red[256];
    const int p = blockIdx.x;
    const int64_t per_row = nchunks;
    const int64_t rows = nparam == 1 ? (int64_t)N * C : N;
    const int64_t total = rows * per_row;
    double acc = 0.0;
    for (int64_t t = threadIdx.x; t < total; t += 256) {
        const int64_t r = t / per_row, k = t % per_row;
        const int64_t row = nparam == 1 ? r : r * C + p;
        acc += part[(size_t)row * nchunks + k];
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) da[p] = (float)red[0];
}

// ---------------------------------------------------------------- global max
__device__ __forceinline__ void argmax_merge(float& v, int64_t& i, float ov, int64_t oi) {
    // ATen keeps the first maximum in scan order (and propagates NaN): larger value wins, ties -> smaller index
    const bool take = (ov > v) || (ov != ov && v == v) || (ov == v && oi < i);
    if (take) { v = ov; i = oi; }
}

__global__ __launch_bounds__(256) void global_max_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                         int64_t* __restrict__ idx, int64_t S) {
    __shared__ float sv[256];
    __shared__ int64_t si[256];
    const int row = blockIdx.x;
    const float* xr = x + (int64_t)row * S;
    float v = -INFINITY;
    int64_t i = 0x7fffffffffffffffLL;
    for (int64_t e = threadIdx.x; e < S; e += 256) argmax_merge(v, i, xr[e], e);
    sv[threadIdx.x] = v;
    si[threadIdx.x] = i;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            float a = sv[threadIdx.x];
            int64_t ai = si[threadIdx.x];
            argmax_merge(a, ai, sv[threadIdx.x + s], si[threadIdx.x + s]);
            sv[threadIdx.x] = a;
            si[threadIdx.x] = ai;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[row] = sv[0]; idx[row] = si[0]; }
}

__global__ __launch_bounds__(256) void global_max_bwd_kernel(const float* __restrict__ dout, const int64_t* __restrict__ idx,
                                                             float* __restrict__ dx, int64_t S) {
    const int row = blockIdx.y;
    const int64_t hit = idx[row];
    const float g = dout[row];
    float* dr = dx + (int64_t)row * S;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < S; e += stride) dr[e] = e == hit ? g : 0.f;
}

static inline unsigned stream_blocks(int64_t S) {
    const int64_t b = cdiv64(S, 256 * 8);
    return (unsigned)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}

}  // namespace dram

using namespace dram;

extern "C" int dram_prelu_fwd(const float* x, const float* a, float* y, int N, int C, int nparam, int64_t S, void* stream) {
    DRAM_REQUIRE(x && a && y, "prelu_fwd: null pointer");
    DRAM_REQUIRE(N > 0 && C > 0 && S > 0 && (int64_t)N * C <= 65535, "prelu_fwd: bad dimensions");
    DRAM_REQUIRE(nparam == 1 || nparam == C, "prelu_fwd: num_parameters must be 1 or C");
    hipLaunchKernelGGL(prelu_fwd_kernel, dim3(stream_blocks(S), N * C), dim3(256), 0, (hipStream_t)stream, x, a, y, C, nparam, S);
    return check_launch("prelu_fwd");
}

extern "C" size_t dram_prelu_bwd_ws_bytes(int N, int C, int64_t S) {
    if (N <= 0 || C <= 0 || S <= 0) return 0;
    return (size_t)N * C * cdiv64(S, ACHUNK) * sizeof(float);
}

extern "C" int dram_prelu_bwd(const float* dy, const float* x, const float* a, float* dx, float* da, void* ws,
                              size_t ws_bytes, int N, int C, int nparam, int64_t S, void* stream) {
    DRAM_REQUIRE(dy && x && a && da && ws, "prelu_bwd: null pointer");
    DRAM_REQUIRE(N > 0 && C > 0 && S > 0 && (int64_t)N * C <= 65535, "prelu_bwd: bad dimensions");
    DRAM_REQUIRE(nparam == 1 || nparam == C, "prelu_bwd: num_parameters must be 1 or C");
    if (ws_bytes < dram_prelu_bwd_ws_bytes(N, C, S)) {
        set_error("prelu_bwd: workspace too small");
        return DRAM_EWS;
    }
    hipStream_t st = (hipStream_t)stream;
    const int nch = (int)cdiv64(S, ACHUNK);
    hipLaunchKernelGGL(prelu_bwd_kernel, dim3(nch, N * C), dim3(256), 0, st, dy, x, a, dx, (float*)ws, C, nparam, S, nch);
    hipLaunchKernelGGL(prelu_da_kernel, dim3(nparam), dim3(256), 0, st, (const float*)ws, da, N, C, nparam, nch);
    return check_launch("prelu_bwd");
}

extern "C" int dram_global_max_fwd(const float* x, float* out, int64_t* idx, int NC, int64_t S, void* stream) {
    DRAM_REQUIRE(x && out && idx, "global_max_fwd: null pointer");
    DRAM_REQUIRE(NC > 0 && S > 0, "global_max_fwd: bad dimensions");
    hipLaunchKernelGGL(global_max_kernel, dim3(NC), dim3(256), 0, (hipStream_t)stream, x, out, idx, S);
    return check_launch("global_max_fwd");
}

extern "C" int dram_global_max_bwd(const float* dout, const int64_t* idx, float* dx, int NC, int64_t S, void* stream) {
    DRAM_REQUIRE(dout && idx && dx, "global_max_bwd: null pointer");
    DRAM_REQUIRE(NC > 0 && NC <= 65535 && S > 0, "global_max_bwd: bad dimensions");
    hipLaunchKernelGGL(global_max_bwd_kernel, dim3(stream_blocks(S), NC), dim3(256), 0, (hipStream_t)stream, dout, idx, dx, S);
    return check_launch("global_max_bwd");
}
