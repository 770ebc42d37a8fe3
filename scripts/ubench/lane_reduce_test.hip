// Checks csrc/lane_reduce.h on the device: lane_transpose_reduce / lane_transpose_broadcast for 16 and 32 registers against
// sums formed on the host (integer-valued floats: exact), the lane -> register map, and the cycle cost of one reduce.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 scripts/ubench/lane_reduce_test.hip -o scripts/ubench/lane_reduce_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>
#include <vector>
#include "../../bodyct-dram_amd/csrc/lane_reduce.h"

template <int NREG>
__global__ void k_reduce(const float* in, float* out, float* bc, long long* cyc) {
    const int lane = threadIdx.x;
    float x[NREG];
#pragma unroll
    for (int i = 0; i < NREG; ++i) x[i] = in[i * 64 + lane];
    const long long t0 = __builtin_readcyclecounter();
    dram::lane_transpose_reduce<NREG>(x, lane);
    const long long t1 = __builtin_readcyclecounter();
    out[lane] = x[0];
    float q[NREG];
    q[0] = x[0] * 0.5f;
    dram::lane_transpose_broadcast<NREG>(q, lane);
    const long long t2 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < NREG; ++i) bc[i * 64 + lane] = q[i];
    if (lane == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; }
}

template <int NREG>
int run() {
    std::vector<float> in(NREG * 64), out(64), bc(NREG * 64);
    for (int i = 0; i < NREG; ++i)
        for (int l = 0; l < 64; ++l) in[i * 64 + l] = (float)((i * 131 + l * 17 + (i * l) % 7) % 1000 - 400);
    float *din, *dout, *dbc;
    long long* dc;
    hipMalloc(&din, in.size() * 4); hipMalloc(&dout, 64 * 4); hipMalloc(&dbc, bc.size() * 4); hipMalloc(&dc, 16);
    hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice);
    long long cyc[2];
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k_reduce<NREG>, dim3(1), dim3(64), 0, 0, din, dout, dbc, dc);
    hipDeviceSynchronize();
    hipMemcpy(out.data(), dout, 64 * 4, hipMemcpyDeviceToHost);
    hipMemcpy(bc.data(), dbc, bc.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(cyc, dc, 16, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const int half = l >> 5, rp = (l >> 4) & 1, ll = l & 15;
        const int reg = NREG == 32 ? 16 * rp + ll : 8 * rp + (ll >> 1);
        float want = 0.f;
        for (int m = 0; m < 32; ++m) want += in[reg * 64 + 32 * half + m];
        if (out[l] != want) { if (bad < 8) printf("  NREG %d lane %d: register %d total %g, got %g\n", NREG, l, reg, want, out[l]); ++bad; }
    }
    int badb = 0;
    for (int i = 0; i < NREG; ++i)
        for (int l = 0; l < 64; ++l) {
            float want = 0.f;
            for (int m = 0; m < 32; ++m) want += in[i * 64 + 32 * (l >> 5) + m];
            want *= 0.5f;
            if (bc[i * 64 + l] != want) { if (badb < 8) printf("  NREG %d broadcast reg %d lane %d: want %g got %g\n", NREG, i, l, want, bc[i * 64 + l]); ++badb; }
        }
    printf("NREG %d: reduce %s (%d bad), broadcast %s (%d bad); cycles reduce %lld broadcast %lld\n", NREG, bad ? "FAIL" : "ok", bad,
           badb ? "FAIL" : "ok", badb, cyc[0], cyc[1]);
    return bad + badb;
}

int main() {
    const int b = run<16>() + run<32>();
    return b ? 1 : 0;
}
