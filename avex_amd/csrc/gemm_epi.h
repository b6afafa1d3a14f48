// Row-statistics helpers of the residual epilogues (gemm.hip: gemm256p_kernel EPI 2; gemm_row.hip: the full-row kernel).  Both kernels
// call exactly these functions on exactly the same values, so the folded-LayerNorm statistics they leave agree bit for bit.
#pragma once
#include "common.h"

// Raw-buffer access for the fast epilogues (EPI 1 / 2): the wave's 64 output rows are ONE buffer whose size ends with the last valid row,
// so rows past M are dropped (stores) or read as zero (loads) by the hardware's bounds check -- no exec masking, no row clamp -- and an
// address is a 32-bit lane offset (made once per tile) + a scalar offset per store instead of a 64-bit multiply-add per row.  (As
// global_store with a per-row "m < M" the epilogue was ~30 basic blocks of 64-bit address arithmetic: v_mul_lo_u32 / v_mad_u64_u32 are
// quarter-rate instructions.)  AUX 2 = the non-temporal hint (GEMM_NT above).
typedef int i32x4_buf __attribute__((ext_vector_type(4)));
typedef int i32x2_buf __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_rsrc(const void* base, unsigned bytes) {
    // the inputs ARE wave-uniform (kernel arguments, tile and wave indices); the readfirstlanes make that provable, or every buffer
    // instruction is wrapped in a "waterfall" loop (4 x v_readfirstlane + compares + exec juggling per store)
    const uint64_t a = (uint64_t)base;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
    const int nb = __builtin_amdgcn_readfirstlane((int)bytes);
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, nb, 0x00020000);
}
// The scalar-offset operand of these instructions stays 0, and constant parts of an offset are added to the lane offset (the compiler
// folds them into the instruction's 12-bit immediate).  With a REGISTER there, hipcc's hazard recogniser assumes that a 128-bit store
// needs no wait states before a VALU instruction overwrites its data registers (LLVM: "this hazard only exists if the instruction is not
// using a register in the soffset field") -- on gfx950 it does: `buffer_store_dwordx4 v[48:51], v120, s[28:31], s65 offen nt` directly
// followed by `v_pk_mul_f32 v[48:49], ...` stored the NEW second dword for lanes 12-15 of every 16 (profiles/r04d_store_hazard.txt).
template <int AUX, typename V>
static __device__ __forceinline__ void buf_st16(const V& v, __amdgpu_buffer_rsrc_t r, int voff) {
    static_assert(sizeof(V) == 16, "16-byte stores only");
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4_buf, v), r, voff, 0, AUX);
}
template <typename V>
static __device__ __forceinline__ V buf_ld16(__amdgpu_buffer_rsrc_t r, int voff) {
    static_assert(sizeof(V) == 16, "16-byte loads only");
    return __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0));
}

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

