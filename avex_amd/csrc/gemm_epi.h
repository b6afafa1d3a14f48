// Row-statistics helpers of the residual epilogues (gemm.hip: gemm256p_kernel EPI 2; gemm_row.hip: the full-row kernel).  Both kernels
// call exactly these functions on exactly the same values, so the folded-LayerNorm statistics they leave agree bit for bit.
#pragma once
#include "common.h"

// Sum over the 8 consecutive lanes that share (lane >> 3), valid in the lane with (lane & 7) == 0: three DPP steps
// (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_shl:4), no LDS traffic (a __shfl_xor becomes a ds_bpermute round trip).
static __device__ __forceinline__ float seg8_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x104, 0xF, 0xF, true));
    return v;
}
// The same reduction for TWO values at once with the DPP operand INSIDE the add (v_add_f32_dpp): six adds and three one-cycle nops.  From
// the builtin hipcc makes, per step, two v_mov_b32 0 (the `old` operand), two v_mov_b32_dpp and one packed add -- fifteen instructions per
// row segment, 13 % of the residual epilogue's vector instructions.  Written as one asm block because the hazard recogniser does not look
// inside inline assembly: a DPP read needs two wait states after the VALU write of its source (the partner chain's add is one, s_nop 0 the
// other; s_nop 1 covers whatever wrote the inputs).  a + dpp(a) either way: the same bits.
#ifndef GEMM_DPP_ADD
#define GEMM_DPP_ADD 1
#endif
static __device__ __forceinline__ void seg8_sum2(float& s1, float& s2) {
#if GEMM_DPP_ADD
    float a, b;
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %1, %1, %1 row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 0"
        : "=&v"(a), "=&v"(b) : "v"(s1), "v"(s2));
    s1 = a; s2 = b;
#else
    s1 = seg8_sum(s1); s2 = seg8_sum(s2);
#endif
}

// Sum and sum of squares of a lane's 8 values x[0..3], y[0..3] (a row segment's share of the LayerNorm statistics), as PACKED operations down
// the register pairs -- (x.01 + x.23) + (y.01 + y.23), the two halves added last -- 9 instructions.  (Written as a chain of scalar adds the
// compiler packed it anyway, with two register moves per packed add to line the pairs up: ~500 v_mov per tile and wave in the residual
// epilogue.)  Both epilogues that write statistics use this one function, so they agree bit for bit.
static __device__ __forceinline__ void stats8(const f32x4& x, const f32x4& y, float& s1, float& s2) {
    const f32x2 xl = {x[0], x[1]}, xh = {x[2], x[3]}, yl = {y[0], y[1]}, yh = {y[2], y[3]};
    const f32x2 t = (xl + xh) + (yl + yh);
    const f32x2 q = __builtin_elementwise_fma(yh, yh, __builtin_elementwise_fma(yl, yl, __builtin_elementwise_fma(xh, xh, xl * xl)));
    s1 = hsum2(t);
    s2 = hsum2(q);
}

