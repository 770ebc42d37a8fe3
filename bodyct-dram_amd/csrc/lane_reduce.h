// Sums over the 32 lanes of a wave half without touching LDS (gfx950).
//
// A 32x32 MFMA accumulator tile puts 32 positions on the lanes of a wave half (lane >> 5) and 16 channels on the registers.
// The statistics epilogues need, per channel, the sum over the 32 lanes -- for 16 or 32 registers at once.  A butterfly per
// register would be 5 exchanges per register; here every step exchanges HALF of the live registers with a partner lane and
// keeps the other half ("transposing reduce": NREG - 1 exchanges), so that at the end lane j holds the total of ONE
// register, lane_register_index<NREG>(lane).  Round 2 did the exchanges with ds_swizzle (77 LDS instructions per statistics
// epilogue, five dependent LDS latencies per reduce); this form uses
//   * v_permlane16_swap_b32 (new on gfx950: swaps the odd 16-lane rows of one register with the even rows of another) for the
//     step between the two rows of a half: one swap + one add per register PAIR,
//   * DPP row_mirror / row_half_mirror with bank masks for the steps over 8 and 4 lanes: the two v_add_f32_dpp of a register
//     pair write disjoint banks of one destination (no select),
//   * DPP quad_perm + v_cndmask for the steps inside a quad (bank masks cannot tell the lanes of a quad apart).
// All lanes of the wave must be active.
#pragma once

namespace dram {

// a <- [a.row0, b.row0, a.row2, b.row2],  b <- [a.row1, b.row1, a.row3, b.row3]   (rows = 16 lanes)
// (inline asm: hipcc 7.2 maps BOTH results of __builtin_amdgcn_permlane16_swap to the first register -- the ISA of
//  scripts/ubench/lane_reduce_test.hip showed "v_permlane16_swap_b32 v4, v3; v_add_f32 v3, v4, v4".  The wait states
//  around it are the ones the compiler puts around its own: VALU write -> swap read, swap write -> VALU read.)
__device__ __forceinline__ void lr_row_swap(float& a, float& b) {
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}

template <int CTRL>
__device__ __forceinline__ float lr_dpp(float v) {     // quad_perm 0x00-0xFF, row_mirror 0x140, row_half_mirror 0x141
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}

// lanes whose partner bit is clear: lo + partner's lo; lanes whose bit is set: hi + partner's hi.
// MIRROR8 = true: partner = 15 - l inside the 16-lane row (bit 3 of l: banks 2,3);  false: partner = 7 - l inside 8 lanes
// (bit 2: banks 1,3).  (s_nop: a DPP operand written by the instruction just before needs two wait states, and the
// compiler's hazard recogniser does not look into inline asm.)
template <bool MIRROR8>
__device__ __forceinline__ float lr_pair_add(float lo, float hi) {
    float out;
    if (MIRROR8)
        asm("s_nop 1\n\t"
            "v_add_f32_dpp %0, %1, %1 row_mirror row_mask:0xf bank_mask:0x3\n\t"
            "v_add_f32_dpp %0, %2, %2 row_mirror row_mask:0xf bank_mask:0xc"
            : "=&v"(out) : "v"(lo), "v"(hi));
    else
        asm("s_nop 1\n\t"
            "v_add_f32_dpp %0, %1, %1 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %0, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xa"
            : "=&v"(out) : "v"(lo), "v"(hi));
    return out;
}

// The register whose total lane `lane` holds in x[0] after lane_transpose_reduce<NREG>.  NREG = 16: two lanes (l, l ^ 1)
// hold each total.
template <int NREG>
__device__ __forceinline__ int lane_register_index(int lane) {
    const int rp = (lane >> 4) & 1, l = lane & 15;
    return NREG == 32 ? 16 * rp + l : 8 * rp + (l >> 1);
}
template <int NREG>
__device__ __forceinline__ bool lane_writes_total(int lane) {       // one writer per register and wave half
    return NREG == 32 ? true : (lane & 1) == 0;
}

template <int NREG>
__device__ __forceinline__ void lane_transpose_reduce(float (&x)[NREG], int lane) {
    static_assert(NREG == 16 || NREG == 32, "16 or 32 registers");
    constexpr int N1 = NREG / 2, N2 = NREG / 4, N3 = NREG / 8, N4 = NREG / 16;
#pragma unroll
    for (int k = 0; k < N1; ++k) {                  // rows of the half: register k + N1 * (row parity)
        lr_row_swap(x[k], x[k + N1]);
        x[k] += x[k + N1];
    }
#pragma unroll
    for (int k = 0; k < N2; ++k) x[k] = lr_pair_add<true>(x[k], x[k + N2]);      // + N2 * bit 3
#pragma unroll
    for (int k = 0; k < N3; ++k) x[k] = lr_pair_add<false>(x[k], x[k + N3]);     // + N3 * bit 2
    {
        const bool up = (lane & 2) != 0;                                           // + N4 * bit 1
#pragma unroll
        for (int k = 0; k < N4; ++k) {
            const float keep = up ? x[k + N4] : x[k];
            const float send = up ? x[k] : x[k + N4];
            x[k] = keep + lr_dpp<0x4E>(send);       // quad_perm [2,3,0,1]
        }
    }
    if (NREG == 32) {
        const bool up = (lane & 1) != 0;                                           // + bit 0
        const float keep = up ? x[1] : x[0];
        const float send = up ? x[0] : x[1];
        x[0] = keep + lr_dpp<0xB1>(send);           // quad_perm [1,0,3,2]
    } else {
        x[0] += lr_dpp<0xB1>(x[0]);
    }
}

// The inverse: x[0] of lane j = the value of register lane_register_index(j); on return every lane of the half holds all
// NREG values, x[i] = the value of register i.
template <int CTRL, int N, int NREG>
__device__ __forceinline__ void lr_broadcast_step(float (&x)[NREG], bool up) {      // N registers -> 2 N
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const float mine = x[k], theirs = lr_dpp<CTRL>(mine);
        x[k] = up ? theirs : mine;
        x[k + N] = up ? mine : theirs;
    }
}
template <int NREG>
__device__ __forceinline__ void lane_transpose_broadcast(float (&x)[NREG], int lane) {
    static_assert(NREG == 16 || NREG == 32, "16 or 32 registers");
    constexpr int N1 = NREG / 2, N2 = NREG / 4, N3 = NREG / 8, N4 = NREG / 16;
    if (NREG == 32) lr_broadcast_step<0xB1, 1>(x, (lane & 1) != 0);
    lr_broadcast_step<0x4E, N4>(x, (lane & 2) != 0);
    lr_broadcast_step<0x141, N3>(x, (lane & 4) != 0);
    lr_broadcast_step<0x140, N2>(x, (lane & 8) != 0);
#pragma unroll
    for (int k = 0; k < N1; ++k) {                  // the two rows: one copy + one swap per register pair
        float a = x[k], b = x[k];
        lr_row_swap(a, b);
        x[k] = a;
        x[k + N1] = b;
    }
}

}  // namespace dram
