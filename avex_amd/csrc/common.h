// Shared device/host helpers for the gfx950 kernels (wave64, MFMA, LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/avexhip.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------
// Operand-type traits: the same kernels are instantiated for f16 and bf16 MFMA operands.
// ---------------------------------------------------------------------------------------------
template <typename T> struct Half;
template <> struct Half<_Float16> {
    typedef f16x8 v8;
    typedef f16x4 v4;
    static __device__ __forceinline__ _Float16 from(float x) {
        // saturate instead of producing inf: activations are O(1..1e3), this only guards outliers.  Compare + select, NOT fmin / fmax: those
        // return their non-NaN operand, so a NaN came out as -65504 -- a poisoned clip (one NaN sample) got a FINITE embedding, found by
        // tests/test_gpu_e2e.py::test_nan_input_stays_nan_and_stays_in_its_clip; the reference's fp32 path returns NaN for it.  A NaN fails
        // both compares and passes through, like the hardware form below.
        const float lo = x < -65504.0f ? -65504.0f : x;
        return (_Float16)(lo > 65504.0f ? 65504.0f : lo);
    }
    static __device__ __forceinline__ _Float16 from_hw(float x) { return (_Float16)x; }      // saturates through MODE.FP16_OVFL (below)
};
template <> struct Half<__bf16> {
    typedef bf16x8 v8;
    typedef bf16x4 v4;
    static __device__ __forceinline__ __bf16 from(float x) { return (__bf16)x; }
    static __device__ __forceinline__ __bf16 from_hw(float x) { return (__bf16)x; }
};
// The same saturation in HARDWARE: with MODE.FP16_OVFL set, v_cvt_f16_f32 / v_cvt_pk_f16_f32 round a finite value beyond the f16 range
// to +-65504 instead of +-inf (measured on gfx950: scripts/micro/fp16_ovfl.hip, profiles/r04u_fp16_ovfl.txt; inf and NaN pass through,
// where from() turns them into +-65504).  A kernel that calls Half<T>::from_hw must execute AVX_F16_SATURATE_ON() first: the bit is
// per-wave state, clear at wave start.  The "memory" clobber keeps every load -- and with it every conversion of loaded or computed
// values -- behind the mode write.  One v_med3_f32 less per stored element: a quarter of the vector work of a bias-only GEMM epilogue.
// MODE.DX10_CLAMP clear: a `clamp` output modifier passes a NaN through instead of turning it into 0 (gelu_erf2_h below relies on it).  Part of both
// epilogue macros; harmless for kernels that use no clamp modifier.
#define AVX_NAN_CLAMP_OFF_ASM "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 8, 1), 0\n\t"
#define AVX_F16_SATURATE_ON() asm volatile(AVX_NAN_CLAMP_OFF_ASM "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1" ::: "memory")
// MEASURED ON gfx950 (scripts/micro/mfma_nan.hip, round 5): WITH MODE.FP16_OVFL SET, v_mfma_f32_{16x16x32,32x32x16}_f16 TREATS A NaN OPERAND AS 0 AND
// AN INFINITE ONE AS A FINITE MAXIMUM -- a clip with one NaN sample went through every GEMM as ordinary numbers and came out with a finite
// embedding (the reference's fp32 path returns NaN; tests/test_gpu_e2e.py::test_nan_input_stays_nan_and_stays_in_its_clip).  Kernels with MFMAs
// therefore keep the bit CLEAR while their MFMAs run and set it around their conversions only: AVX_F16_SAT_BEGIN() at the start of an
// epilogue, AVX_F16_SAT_END() behind it when more MFMAs follow.  These use the s_setreg builtin, which the compiler knows to write MODE: a
// scheduling boundary no floating-point instruction crosses.  Kernels without MFMAs (depthwise, element-wise) keep the one-time form above.
#define AVX_MODE_FP16_OVFL_HWREG 1473      // hwreg(HW_REG_MODE = 1, offset 23, size 1): id | offset << 6 | (size - 1) << 11
#define AVX_MODE_DX10_CLAMP_HWREG 513      // hwreg(HW_REG_MODE, offset 8, size 1)
#define AVX_F16_SAT_BEGIN() do { __builtin_amdgcn_s_setreg(AVX_MODE_DX10_CLAMP_HWREG, 0); __builtin_amdgcn_s_setreg(AVX_MODE_FP16_OVFL_HWREG, 1); asm volatile("" ::: "memory"); } while (0)
#define AVX_F16_SAT_END() do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_setreg(AVX_MODE_FP16_OVFL_HWREG, 0); __builtin_amdgcn_s_setreg(AVX_MODE_DX10_CLAMP_HWREG, 1); } while (0)
// (SAT_END also puts MODE.DX10_CLAMP back to the value the kernel was compiled for: whatever clamp folds the compiler made in the code that
// follows -- the K loop of a persistent kernel -- see the mode they assume.)
// The code that RELIES on DX10_CLAMP being clear (gelu_clamp_t below: a `clamp` output modifier that must pass a NaN through) is tied to the
// mode write by DATA: its multiplier arrives in a scalar register that a volatile asm defines BEHIND the mode write (volatile asm and the
// s_setreg builtin both have side effects: the compiler keeps their order), so no scheduler can lift the multiply above the s_setreg --
// a register-only, non-volatile asm with no such operand may move freely.  One s_mov_b32 per epilogue.
#define AVX_CLAMP_TOKEN(name) float name; asm volatile("s_mov_b32 %0, 0x3e33a62d" : "=s"(name))      /* 1 / 5.7 (AVX_GELUH_INVA) */

// Range alarm of the f16 outputs (Half<_Float16>::from saturates silently): a kernel keeps the running max of |value| over
// everything a lane rounds to f16 (v_max3_f32 with |.| modifiers: half a VALU slot per element) and commits once at its end.
template <typename T> static __device__ __forceinline__ void ovf_see(float& mx, float a, float b) {
    if constexpr (sizeof(T) == 2 && __is_same(T, _Float16)) mx = __builtin_fmaxf(__builtin_fmaxf(mx, __builtin_fabsf(a)), __builtin_fabsf(b));
}
template <typename T> static __device__ __forceinline__ void ovf_see4(float& mx, f32x4 v) {
    ovf_see<T>(mx, v[0], v[1]);
    ovf_see<T>(mx, v[2], v[3]);
}
// a wave's lanes whose running max left the f16 range, as a (wave-uniform) mask to OR into a scalar that lives across a kernel's main
// loop in SGPRs (a VGPR kept alive through a 256-register MFMA loop would spill)
template <typename T> static __device__ __forceinline__ unsigned long long ovf_mask(float mx) {
    if constexpr (__is_same(T, _Float16)) return __ballot(mx > 65504.0f);
    return 0ull;
}
template <typename T> static __device__ __forceinline__ void ovf_commit(unsigned int* ctr, unsigned long long mask) {
    if constexpr (__is_same(T, _Float16)) {
        if (ctr != nullptr && mask != 0ull && (int)(threadIdx.x & 63) == __ffsll((long long)mask) - 1) atomicAdd(ctr, (unsigned int)__popcll(mask));
    }
}
template <typename T> static __device__ __forceinline__ void ovf_commit(unsigned int* ctr, float mx) { ovf_commit<T>(ctr, ovf_mask<T>(mx)); }

static __device__ __forceinline__ f32x4 mfma16(f16x8 a, f16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
static __device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
static __device__ __forceinline__ f32x16 mfma32(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
static __device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// erf-form GELU (reference: torch.nn.functional.gelu default, modules.py:191-200).  erf is evaluated
// branch-free with Abramowitz-Stegun 7.1.26 (|error| <= 6e-7 in fp32, i.e. 3 orders below the f16
// rounding of the stored activation): ~14 VALU ops instead of libm erff's ~45 with divergent branches,
// which matters because this runs 3072 times per token in the fc1 epilogue.
static __device__ __forceinline__ float fast_erf(float x) {
    const float ax = __builtin_fabsf(x);
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, ax, 1.0f));
    float p = 1.061405429f;
    p = __builtin_fmaf(p, t, -1.453152027f);
    p = __builtin_fmaf(p, t, 1.421413741f);
    p = __builtin_fmaf(p, t, -0.284496736f);
    p = __builtin_fmaf(p, t, 0.254829592f);
    p = p * t;
    const float r = __builtin_fmaf(-p, __expf(-(ax * ax)), 1.0f);
    return __builtin_copysignf(r, x);
}
// Exact-erf GELU with ONE transcendental per element:
//   gelu(x) = x Phi(x) = 0.5 x + |x| (0.5 - q(|x|)),   q(a) = Phi(-a) = 0.5 erfc(a / sqrt2) = 2^P(a)
// P is a degree-6 polynomial fitted to log2 q on [0, 5.7] with the error weighted by a * q (what reaches the output);
// beyond 5.7 the argument is clamped (a * q < 4e-8 there).  |gelu - exact| <= 3.5e-7 over [-10, 10] in fp32 (the A-S 7.1.26
// form this replaces: 2.1e-7, with a v_rcp_f32 and a v_exp_f32 per element; v_exp/v_rcp cost two VALU slots each).
// Explicit FMAs because the library is built with -ffp-contract=off.
#define AVX_GELU_A 5.7f
#define AVX_GELU_C0 -0.999993085861206f
#define AVX_GELU_C1 -1.1512017250061035f
#define AVX_GELU_C2 -0.4587709605693817f
#define AVX_GELU_C3 -0.05341210588812828f
#define AVX_GELU_C4 0.00808071717619896f
#define AVX_GELU_C5 -0.0007692198269069195f
#define AVX_GELU_C6 3.309283420094289e-05f
static __device__ __forceinline__ float gelu_erf(float x) {
    const float ax = __builtin_fabsf(x);
    const float a = __builtin_fminf(ax, AVX_GELU_A);
    float p = AVX_GELU_C6;
    p = __builtin_fmaf(p, a, AVX_GELU_C5);
    p = __builtin_fmaf(p, a, AVX_GELU_C4);
    p = __builtin_fmaf(p, a, AVX_GELU_C3);
    p = __builtin_fmaf(p, a, AVX_GELU_C2);
    p = __builtin_fmaf(p, a, AVX_GELU_C1);
    p = __builtin_fmaf(p, a, AVX_GELU_C0);
    const float u = 0.5f - __builtin_amdgcn_exp2f(p);
    return __builtin_fmaf(ax, u, 0.5f * x);
}

// Two elements at a time on the packed-fp32 pipe (v_pk_fma_f32: two fp32 lanes per VALU slot); used by the GEMM / pos-conv
// epilogues (no MFMA runs beside them).
typedef float f32x2 __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ f32x2 gelu_erf2(f32x2 x) {
    f32x2 ax, a;
    ax[0] = __builtin_fabsf(x[0]); ax[1] = __builtin_fabsf(x[1]);
    a[0] = __builtin_fminf(ax[0], AVX_GELU_A); a[1] = __builtin_fminf(ax[1], AVX_GELU_A);
    f32x2 p = (f32x2)(AVX_GELU_C6);
    p = __builtin_elementwise_fma(p, a, (f32x2)(AVX_GELU_C5));
    p = __builtin_elementwise_fma(p, a, (f32x2)(AVX_GELU_C4));
    p = __builtin_elementwise_fma(p, a, (f32x2)(AVX_GELU_C3));
    p = __builtin_elementwise_fma(p, a, (f32x2)(AVX_GELU_C2));
    p = __builtin_elementwise_fma(p, a, (f32x2)(AVX_GELU_C1));
    p = __builtin_elementwise_fma(p, a, (f32x2)(AVX_GELU_C0));
    f32x2 q;
    q[0] = __builtin_amdgcn_exp2f(p[0]); q[1] = __builtin_amdgcn_exp2f(p[1]);
    const f32x2 u = (f32x2)(0.5f) - q;
    return __builtin_elementwise_fma(ax, u, x * (f32x2)(0.5f));
}
// The GELU of an epilogue whose output is ROUNDED TO THE OPERAND TYPE (fc1 of the streaming GEMM): degree 4 instead of 6 -- two packed FMAs
// per pair less, of ~15 vector slots.  |gelu - exact| <= 6.3e-6 over [-12, 12] in fp32 (fit: weighted minimax of log2 q with weight a q ln 2,
// like the degree-6 one): a tenth of an f16 ulp at |gelu| = 0.1, i.e. + 0.5 % on the rounding noise (sqrt(1 + 12 delta^2)) of a stored
// activation, against 3.5e-7 which is finer than any half type can hold.  fp32 outputs keep the degree-6 form.
#define AVX_GELUH_C0 -1.0004795789718628f
#define AVX_GELUH_C1 -1.1473724842071533f
#define AVX_GELUH_C2 -0.46801453828811646f
#define AVX_GELUH_C3 -0.044079434126615524f
#define AVX_GELUH_C4 0.0038662925362586975f
// ... and in the form  gelu(x) = max(x, 0) - a q(a),  a = min(|x|, 5.7):  the same function (x >= 0: x (1 - q) = x Phi(x); x < 0: -|x| Phi(-|x|)),
// one |x|-clamp (the |.| is an operand modifier), one max and one packed FMA around the exponent instead of |x|, the clamp, a packed multiply,
// a packed subtract and a packed FMA: 2.5 instead of 3.5 vector slots per element.  Beyond the clamp a q is 3.4e-8 instead of |x| q.
#ifndef AVX_GELUH_RELU
#define AVX_GELUH_RELU 1
#endif
// NaN in, NaN out, at no extra instruction (round 5).  min(|x|, 5.7) and max(x, 0) return their non-NaN operand, so the form above turned a NaN
// pre-activation into -3e-8: the fc2 hook taps of a clip with one NaN sample -- what extract_embeddings returns -- were finite (the reference's fp32
// F.gelu gives NaN, backbone.py:368).  The clamp of |x| is now the CLAMP OUTPUT MODIFIER of a multiply, t = clamp01(|x| / 5.7), which passes a NaN
// through when MODE.DX10_CLAMP is clear (AVX_NAN_CLAMP_OFF, set by the epilogue macros below); the polynomial is the same one written in t
// (coefficient i times 5.7^i, log2(5.7) added to the constant so that 2^P'(t) = 5.7 q): |new - old| <= 7.3e-9 over [-12, 12], |gelu - exact| <= 6.1e-6
// as before.  gelu = max(x, 0) - t * 2^P'(t): a NaN t makes the result NaN whatever max() returned.
#define AVX_GELUH_T0 1.5104823112487793f
#define AVX_GELUH_T1 -6.540023326873779f
#define AVX_GELUH_T2 -15.205792427062988f
#define AVX_GELUH_T3 -8.163202285766602f
#define AVX_GELUH_T4 4.081258773803711f
#define AVX_GELUH_INVA 0.17543859779834747f
// `inva`: an AVX_CLAMP_TOKEN of the calling kernel (1 / 5.7 in a scalar register defined behind the write that clears MODE.DX10_CLAMP)
static __device__ __forceinline__ float gelu_clamp_t(float x, float inva) {
    float t;
    asm("v_mul_f32_e64 %0, |%1|, %2 clamp" : "=v"(t) : "v"(x), "s"(inva));
    return t;
}
static __device__ __forceinline__ f32x2 gelu_erf2_h(f32x2 x, float inva) {
#if AVX_GELUH_RELU
    const f32x2 t = {gelu_clamp_t(x[0], inva), gelu_clamp_t(x[1], inva)};
    f32x2 p = __builtin_elementwise_fma((f32x2)(AVX_GELUH_T4), t, (f32x2)(AVX_GELUH_T3));
    p = __builtin_elementwise_fma(p, t, (f32x2)(AVX_GELUH_T2));
    p = __builtin_elementwise_fma(p, t, (f32x2)(AVX_GELUH_T1));
    p = __builtin_elementwise_fma(p, t, (f32x2)(AVX_GELUH_T0));
    f32x2 q, r;
    q[0] = __builtin_amdgcn_exp2f(p[0]); q[1] = __builtin_amdgcn_exp2f(p[1]);
    r[0] = __builtin_fmaxf(x[0], 0.f); r[1] = __builtin_fmaxf(x[1], 0.f);
    return __builtin_elementwise_fma(-t, q, r);
#else
    f32x2 a;
    a[0] = __builtin_fminf(__builtin_fabsf(x[0]), AVX_GELU_A); a[1] = __builtin_fminf(__builtin_fabsf(x[1]), AVX_GELU_A);
    f32x2 p = __builtin_elementwise_fma((f32x2)(AVX_GELUH_C4), a, (f32x2)(AVX_GELUH_C3));
    p = __builtin_elementwise_fma(p, a, (f32x2)(AVX_GELUH_C2));
    p = __builtin_elementwise_fma(p, a, (f32x2)(AVX_GELUH_C1));
    p = __builtin_elementwise_fma(p, a, (f32x2)(AVX_GELUH_C0));
    f32x2 q;
    q[0] = __builtin_amdgcn_exp2f(p[0]); q[1] = __builtin_amdgcn_exp2f(p[1]);
    f32x2 ax;
    ax[0] = __builtin_fabsf(x[0]); ax[1] = __builtin_fabsf(x[1]);
    const f32x2 u = (f32x2)(0.5f) - q;
    return __builtin_elementwise_fma(ax, u, x * (f32x2)(0.5f));
#endif
}
static __device__ __forceinline__ f32x4 gelu_erf4_h(f32x4 v, float inva) {
    const f32x2 a = gelu_erf2_h((f32x2){v[0], v[1]}, inva), b = gelu_erf2_h((f32x2){v[2], v[3]}, inva);
    return (f32x4){a[0], a[1], b[0], b[1]};
}
// The same GELU over N pairs at once, written step by step ACROSS the pairs: N independent chains side by side in program order (the
// scheduler keeps a dependent chain of packed FMAs together when it is handed one pair at a time, and each link then waits for the last).
template <int N>
static __device__ __forceinline__ void gelu_erf2xN(f32x2 (&x)[N]) {
    f32x2 ax[N], a[N], p[N];
#pragma unroll
    for (int q = 0; q < N; ++q) {
        ax[q][0] = __builtin_fabsf(x[q][0]); ax[q][1] = __builtin_fabsf(x[q][1]);
        a[q][0] = __builtin_fminf(ax[q][0], AVX_GELU_A); a[q][1] = __builtin_fminf(ax[q][1], AVX_GELU_A);
    }
#pragma unroll
    for (int q = 0; q < N; ++q) p[q] = __builtin_elementwise_fma((f32x2)(AVX_GELU_C6), a[q], (f32x2)(AVX_GELU_C5));
#pragma unroll
    for (int q = 0; q < N; ++q) p[q] = __builtin_elementwise_fma(p[q], a[q], (f32x2)(AVX_GELU_C4));
#pragma unroll
    for (int q = 0; q < N; ++q) p[q] = __builtin_elementwise_fma(p[q], a[q], (f32x2)(AVX_GELU_C3));
#pragma unroll
    for (int q = 0; q < N; ++q) p[q] = __builtin_elementwise_fma(p[q], a[q], (f32x2)(AVX_GELU_C2));
#pragma unroll
    for (int q = 0; q < N; ++q) p[q] = __builtin_elementwise_fma(p[q], a[q], (f32x2)(AVX_GELU_C1));
#pragma unroll
    for (int q = 0; q < N; ++q) p[q] = __builtin_elementwise_fma(p[q], a[q], (f32x2)(AVX_GELU_C0));
#pragma unroll
    for (int q = 0; q < N; ++q) { p[q][0] = __builtin_amdgcn_exp2f(p[q][0]); p[q][1] = __builtin_amdgcn_exp2f(p[q][1]); }
#pragma unroll
    for (int q = 0; q < N; ++q) x[q] = __builtin_elementwise_fma(ax[q], (f32x2)(0.5f) - p[q], x[q] * (f32x2)(0.5f));
}
// SiLU x * sigmoid(x) = x / (1 + 2^(-x log2 e)) (EfficientNet's activation), two elements at a time
static __device__ __forceinline__ f32x2 silu2(f32x2 x) {
    const f32x2 t = x * (f32x2)(-1.4426950408889634f);
    f32x2 e;
    e[0] = __builtin_amdgcn_exp2f(t[0]); e[1] = __builtin_amdgcn_exp2f(t[1]);
    const f32x2 d = e + (f32x2)(1.0f);          // one v_pk_add_f32 (written per element hipcc kept two v_add_f32)
    f32x2 r;
    r[0] = __builtin_amdgcn_rcpf(d[0]); r[1] = __builtin_amdgcn_rcpf(d[1]);
    return x * r;
}
// activation selector of the GEMM epilogues: 1 = exact-erf GELU, 2 = SiLU
static __device__ __forceinline__ f32x4 act4(f32x4 v, int act);
static __device__ __forceinline__ f32x4 gelu_erf4(f32x4 v) {
    const f32x2 a = gelu_erf2((f32x2){v[0], v[1]}), b = gelu_erf2((f32x2){v[2], v[3]});
    return (f32x4){a[0], a[1], b[0], b[1]};
}
static __device__ __forceinline__ f32x4 silu4(f32x4 v) {
    const f32x2 a = silu2((f32x2){v[0], v[1]}), b = silu2((f32x2){v[2], v[3]});
    return (f32x4){a[0], a[1], b[0], b[1]};
}
static __device__ __forceinline__ f32x4 act4(f32x4 v, int act) { return act == 2 ? silu4(v) : gelu_erf4(v); }
// every activation code of GemmArgs::gelu (the generic epilogues; the streaming kernel's fast epilogue takes 1 and 2 only):
// 3 = ReLU, 4 = tanh-form GELU (the reference's gelu_accurate, modules.py:177-188), 5 = tanh
static __device__ __forceinline__ float tanh_fast(float x) {      // 1 - 2 / (1 + e^{2x}); saturates cleanly at +-1
    const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + e);
}
// 6 = exact-erf GELU of a product whose ONLY output is in the operand type: the degree-4 fit (gelu_erf4_h).  avx::gemm turns 1 into 6 for such
// products, so that every kernel and epilogue form rounds the same value (the fast epilogue of the streaming kernel is one of them).
static __device__ __forceinline__ f32x4 act4_any(f32x4 v, int act, float inva) {
    if (act == 6) return gelu_erf4_h(v, inva);
    if (act <= 2) return act4(v, act);
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float x = v[e];
        if (act == 3) r[e] = __builtin_fmaxf(x, 0.f);
        else if (act == 4) r[e] = 0.5f * x * (1.0f + tanh_fast(0.7978845608028654f * (x + 0.044715f * x * x * x)));
        else r[e] = tanh_fast(x);
    }
    return r;
}

// x[0] + x[1] as ONE plain v_add_f32 the compiler cannot merge with a neighbour into a packed add with swapped halves
// (v_pk_add_f32 ... op_sel:[0,1] is wrong beside MFMA work on gfx950, avex_amd/isa_lint.py).
static __device__ __forceinline__ float hsum2(f32x2 x) {
    float r;
    asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(x[0]), "v"(x[1]));
    return r;
}

static __device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
static __device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// XCD-aware, bijective block remap: blocks that share blockIdx % 8 share an XCD (and its L2);
// give each XCD a contiguous range of logical tiles (cdna guide T1).
static __device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// ---------------------------------------------------------------------------------------------
// Host-side error plumbing
// ---------------------------------------------------------------------------------------------
void avexhip_set_error(const char* fmt, ...);

#define AVX_HIP_CHECK(expr)                                                                   \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            avexhip_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                              __LINE__);                                                      \
            return AVEXHIP_ERR_HIP;                                                           \
        }                                                                                     \
    } while (0)

#define AVX_REQUIRE(cond, ...)                 \
    do {                                       \
        if (!(cond)) {                         \
            avexhip_set_error(__VA_ARGS__);    \
            return AVEXHIP_ERR_INVALID;        \
        }                                      \
    } while (0)

#define AVX_LAUNCH_CHECK()                                                                   \
    do {                                                                                     \
        hipError_t _e = hipGetLastError();                                                   \
        if (_e != hipSuccess) {                                                              \
            avexhip_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e),     \
                              __FILE__, __LINE__);                                           \
            return AVEXHIP_ERR_HIP;                                                          \
        }                                                                                    \
    } while (0)

// Internal launchers shared between translation units (all return AVEXHIP_* codes).
namespace avx {

// Per-device launcher state (thread-safe): the dynamic-LDS opt-in of a kernel is a property of the function ON ONE DEVICE, and the
// CU count differs between devices; a process that drives several GPUs, or several threads, must not share one `static bool`.
//   AVX_ENSURE_LDS(kernel, bytes)  hipFuncSetAttribute(..MaxDynamicSharedMemorySize..) once per (current device, kernel)
//   avx::device_cu_count(&n)       multiProcessorCount of the current device, cached per device
int ensure_max_dynamic_lds(const void* func, int bytes);
int device_cu_count(int* n_cu);
#define AVX_ENSURE_LDS(kernel, bytes) do { const int rc_ = avx::ensure_max_dynamic_lds((const void*)(kernel), (int)(bytes)); if (rc_ != AVEXHIP_OK) return rc_; } while (0)

struct GemmArgs {
    const void* A; int64_t lda;
    const void* W; int64_t ldw;
    int M, N, K;
    const float* bias;
    const float* resid; int64_t ldr; float alpha;
    const void* resid_half; int64_t ldrh;   // residual in the operand type (used when resid == NULL)
    int gelu;
    float* out_f32; int64_t ldo;
    void* out_half; int64_t ldh;
    float* out_raw; int64_t ldraw;
    const uint8_t* row_zero;  // optional [M] mask: rows with 1 store zeros to every output
    float half_scale;         // 0 or 1: off.  Otherwise out_half receives value * half_scale (a power of two: the third rung of the f16 range ladder stores
                              // fc1's hidden activations scaled down and folds the inverse into fc2's weights); generic epilogues only, fp32 outputs unscaled
    int variant;
    // ---- LayerNorm folded into the GEMMs around it (256-tile streaming kernel) --------------------------------------
    // A tensor y that is only ever consumed through LayerNorm is kept RAW in the operand type.  The GEMM that produces y writes
    // per-row partial statistics [M][N/64][2] = (sum, sum of squares) of each 64-column segment (stats_out); avx::ln_rowstats
    // reduces them, in a fixed order, to one (rstd, -mu * rstd) pair per row; the consumers take those pairs:
    //  * consumer of LN(y) as its A operand:  A = y, W = W * diag(gamma) (folded by the caller), ln_s[n] = sum_k W'[n][k],
    //    bias = b + W beta;  the epilogue forms  rstd[m] * acc + ((-mu rstd)[m] * ln_s[n] + bias[n])  before GELU / rounding;
    //  * consumer of LN(y) as its residual:  out = alpha * ((y * rstd - mu rstd) * gamma + beta) + acc + bias;
    //  * producer: stats_out receives the partial statistics of the rows it writes (from the fp32 values before rounding).
    const float* ln_rows;     // consumer-as-A: [M rounded up to 256][2] (rstd, -mu * rstd) of the A rows, or NULL
    const float* ln_s;        // [N]
    const void* lnr_y;        // consumer-as-residual: raw residual rows (half), or NULL
    int ldy;
    const float* lnr_rows;    // [M][2] (rstd, -mu * rstd) of the residual rows
    const float* lnr_gamma;   // [N]
    const float* lnr_beta;    // [N]
    int lnr_prefolded;        // lnr_gamma holds alpha * gamma and lnr_beta holds alpha * beta + bias (what the kernel takes; avx::lnr_fold makes them).
                              // avx::gemm folds per launch when the flag is clear; callers that launch the same fold repeatedly keep the vectors
    float* stats_out;         // [M][N/64][2] or NULL
    // The finished row statistics instead of (or beside) the partial ones: rows_out[m] = (rstd, -mu rstd) of output row m with rows_eps inside
    // the root -- what avx::ln_rowstats makes of stats_out, same bits.  The full-row kernel (gemm_row.hip, N = 768) writes them from its
    // epilogue; when another kernel runs the product, avx::gemm writes the partials to stats_out (required then, as scratch) and launches
    // ln_rowstats itself.  Readable / writable up to M rounded up to even.
    float* rows_out; float rows_eps;
    // Mean-pooled hook tap without the tap: the rows are clips of pool_T (>= 64) consecutive rows; each 64-row block writes the column
    // sums of acc + bias (what out_raw would hold) over its rows, split at the one clip boundary it can contain:
    // pool_part[block][slot][N], slot 0 = the clip of the block's first row, slot 1 = the next clip.  avx::pool_reduce adds a clip's
    // blocks in order and divides by pool_T.  256-tile kernel, generic epilogue.
    float* pool_part; int pool_T;
    int pool_mode;            // 0: column sums (mean after pool_reduce); 1: column maxima; 2: each clip's FIRST row, written straight to pool_part = [clips][N]
    // 0, or the number of leading output columns that exist in memory (a multiple of 4, < N): the product is computed for N (a multiple of
    // the tile width, W and bias padded by the caller) but rows of every output / residual are only n_store wide.  128-tile kernels.
    int n_store;
    // skinny kernel (variant 7): A[m][k] *= a_scale[(m / a_scale_rows) * a_scale_ld + k] (fp32 product rounded to the operand type) as the rows
    // are loaded: a per-(clip, channel) rescale of the input without a pass of its own (EfficientNet's squeeze-excitation)
    const float* a_scale; int a_scale_rows; int a_scale_ld;
    // 128-tile LDS-DMA kernel: scratch for split-K (fp32 partial products [S][M][N]), or NULL.  With it a product of <= 64 tiles and
    // K >= 1024 is split S <= 8 ways along K and finished by splitk_epilogue_kernel (partials added in order).
    float* splitk_ws; size_t splitk_bytes;
    // A LayerNorm of the output rows y (after bias / residual) in the same pass: the workspace path above with an epilogue kernel that owns
    // whole rows (one wave per row, N <= 1024, N % 256 == 0, no activation): post_ln_out_* = LN(y) * post_ln_w + post_ln_b; y itself still
    // goes to out_f32 / out_half when those are set.  post_ln_round: y is rounded to the operand type before the statistics -- what a
    // LayerNorm kernel reading a half residual stream sees.  For the few-row products of one to eight clips, where a kernel less per
    // LayerNorm is 13 us less (avx::gemm_post_ln_ok says whether a product qualifies).
    const float* post_ln_w; const float* post_ln_b; float post_ln_eps; int post_ln_round;
    float* post_ln_out_f32; int64_t post_ln_ldo; void* post_ln_out_half; int64_t post_ln_ldh;
    // sticky range alarm: the number of (lane, launch) pairs that rounded at least one |value| > 65504 to an f16 output is added
    // here (one atomic per wave at most, at the end of the kernel); NULL = not counted.  bf16 outputs cannot overflow.
    unsigned int* ovf;
    int tile_order;           // 256-tile kernel: 0 = grouped walk where K < 2048 (default), 1 = row-major, >= 2 = grouped walk with that many row panels per group
    int nt;                   // set by the launcher: bit 0 non-temporal output stores (256-tile kernels)
};
int gemm(const GemmArgs& a, int dtype, hipStream_t s);
// the full-row residual kernel (gemm_row.hip): N = 768, one workgroup per 128 rows x all columns; avx::gemm dispatches to it (variant 8, or auto)
bool gemm_row_ok(const GemmArgs& a);
int gemm_row(const GemmArgs& a, int dtype, hipStream_t s);
// fp32 NHWC rows [B * HW, ld] -> NCHW [B, C, HW], optionally undoing a folded BatchNorm: (x - shift[c]) / scale[c] (effnet.hip)
int nhwc_to_nchw(const float* in, int64_t ld, int B, int HW, int C, const float* scale, const float* shift, float* out, hipStream_t s);
// MBConv front in one kernel: 1x1 expansion (kin = 32 | 64 input channels) + BN + SiLU + depthwise k x k + BN + SiLU + squeeze sums (effnet.hip)
int64_t mbconv_front_tiles(int H, int W, int k, int stride, int kin);
// the handle's squeeze path: depthwise kernels leave rows of partial sums (part), se_from_parts adds them in order and runs both layers
int dwconv_parts(const void* in, int B, int H, int W, int Cp, int k, int stride, const float* w, const float* bias, void* out, float* part, size_t part_bytes,
                 int64_t* rows, int dtype, hipStream_t s);
int64_t dwconv_lds_tiles(int H, int W, int k, int stride);
int dwconv_lds_parts(const void* in, int B, int H, int W, int Cp, int k, int stride, const float* w, const float* bias, void* out, float* part,
                     size_t part_bytes, int64_t* rows, int dtype, hipStream_t s);
int se_from_parts(const float* part, int64_t rows, int B, int64_t hw, int C, int Cp, int Cs, const float* w1, const float* b1, const float* w2t,
                  const float* b2, float* scale, void* x, int dtype, hipStream_t s);
int mbconv_front(const void* in, int B, int H, int W, int ld_in, int kin, const void* w_exp, int ldw, const float* b_exp, int k, int stride,
                 const float* w_dw, const float* b_dw, int cp_exp, void* out, float* pool, float* part, size_t part_bytes, unsigned int* ovf,
                 int dtype, hipStream_t s);
// GemmArgs::pool_part [ceil(M / 64)][2][N] -> out[b][n] = mean over clip b's T rows (b < B, M = B * T), blocks added in order
//                                                 (mode 1: the maximum over the clip's rows instead)
int pool_reduce(const float* part, int B, int T, int N, float* out, int64_t ldo, hipStream_t s, int mode = 0);
// fp32 [B, T, C] -> [B, C]: mode 1 mean, 2 max over the T rows, 3 the first row (the aggregations of extract_embeddings, beats_model.py:403-417)
int agg_pool(const float* in, int B, int T, int C, int mode, float* out, hipStream_t s);
// partial statistics [M][nseg][2] (GemmArgs::stats_out) -> [M][2] (rstd, -mu * rstd), summed in segment order (deterministic)
int ln_rowstats(const float* stats, int M, int nseg, float eps, float* rows, hipStream_t s);
// [H, 2T - 1] Toeplitz rows of the relative position bias from the resident [buckets, H] table and the host-made bucket LUT (elementwise.hip)
int bias_toeplitz(const float* rel_table, const int* lut, int maxd, int T, int H, float* out, hipStream_t s);
// ga = alpha * gamma, bb = bias + alpha * beta: the column vectors a residual-side fold takes (GemmArgs::lnr_prefolded)
int lnr_fold(const float* gamma, const float* beta, const float* bias, float alpha, int N, float* ga, float* bb, hipStream_t s);
// exactly one of in / in_half is non-null
int layernorm(const float* in, const void* in_half, int64_t ld_in, const float* w, const float* b, float eps, int M,
              int C, float* out_f32, int64_t ldo, void* out_half, int64_t ldh, int dtype, hipStream_t s);
// ovf: optional device counter of the lanes that rounded a value beyond +-65504 to an f16 destination (see GemmArgs::ovf)
int cast_to_half(const float* in, void* out, int64_t n, int dtype, hipStream_t s, unsigned int* ovf = nullptr);
int row_sum_half(const void* w, int N, int K, float* out, int dtype, hipStream_t s);
int cast_to_f32(const void* in, float* out, int64_t n, int dtype, hipStream_t s);
int mean_pool(const float* in, int B, int T, int C, const uint8_t* frame_pad, float* out, hipStream_t s);
// [B * Tp, C] patch rows (half) -> [B * (Tp + 1), C]: class token row + (patch + position) rows, LayerNorm'ed
int token_embed_ln(const void* patches, const float* pos, const float* cls, const float* w, const float* b, float eps, int B, int Tp, int C,
                   void* out_half, float* out_f32, int dtype, hipStream_t s);
// rows with pad[m] != 0 set to zero in the fp32 and / or the operand-type copy (either may be NULL)
// out[m][f] = in[m][f] * swish(in[m][F + f]) for a [M, 2F] half matrix: the second half of the reference's GLU_Linear(E, F, "swish")
// (modules.py:155-171); values that leave the f16 range count into *ovf (may be NULL)
int glu_swish(const void* in, int64_t M, int F, void* out, unsigned int* ovf, int dtype, hipStream_t s);
int zero_rows(float* x32, int64_t ld32, void* x_half, int64_t ldh, int M, int C, const uint8_t* pad, hipStream_t s);
// final LayerNorm + mean over tokens in one pass (half rows in, [B, C] fp32 out); C % 8 == 0, C <= 768
int layernorm_pool(const void* in_half, int64_t ld_in, const float* w, const float* b, float eps, int B, int T, int C, float* out, int dtype,
                   hipStream_t s);
// q_log2e != 0: the Q columns of qkv already carry log2(e) (folded into W_q / b_q in fp32 by the handles); the kernel then scales by the
// exact 1/8 only and feeds the gate with weights divided by log2(e)
// avx::gemm's choice for a plain product (variant 0, no folded LayerNorm, no pooled tap): the 256-tile streaming kernel (true) or the 128-tile one
bool gemm_streams(int M, int N);
// whether a product (M, N, K set; splitk_ws lent) can take GemmArgs::post_ln_*
bool gemm_post_ln_ok(const GemmArgs& a);
int attention(const void* qkv, int B, int T, int H, const float* bias_tab, const float* grep_w,
              const float* grep_b, const float* grep_a, const uint8_t* key_pad, void* out, int dtype,
              hipStream_t s, int q_log2e = 0);
// variant 3 of the same (attention16.hip: 16x16x32 MFMAs, up to 512 tokens); avx::attention dispatches to it
int attention16(const void* qkv, int B, int T, int H, const float* bias_tab, const float* grep_w, const float* grep_b, const float* grep_a,
                const uint8_t* key_pad, void* out, int dtype, int q_log2e, int n_wg, hipStream_t s);
int attention_hd(const void* qkv, int B, int T, int H, int head_dim, const uint8_t* key_pad, void* out, int dtype, hipStream_t s, int q_log2e = 0);
int posconv_pack(const float* g, const float* v, int E, int groups, int K, void* w_packed, int dtype,
                 hipStream_t s);
// residual = x_f32 if non-null else x_half; writes out_f32 and/or out_half
int posconv(const void* x_half, const float* x_f32, const void* w_packed, const float* bias, int B, int T,
            int E, int groups, int K, float* out_f32, void* out_half, int dtype, hipStream_t s);

struct FbankDev {
    int win, hop, n_mels;
    float input_scale, preemph, log_floor, norm_mean, norm_div;
    int remove_dc;
    const float* window;      // [win]
    const float2* twiddle;    // [512] (cos, -sin)(2 pi k / 512)
    const int* mel_start;     // [n_mels] first FFT bin with non-zero weight
    const int* mel_len;       // [n_mels]
    const int* mel_off;       // [n_mels] offset into mel_w
    const float* mel_w;       // packed non-zero weights
};
// out_f32: [B, frames, n_mels] or NULL; out_patch: half patch-major [B, frames/P, n_mels/P, P*P] or NULL
int fbank(const FbankDev& fb, const float* wav, int B, int64_t T, int64_t stride, int frames,
          float* out_f32, void* out_patch, int patch, int dtype, hipStream_t s, const float* clip_offset = nullptr,
          int out_frames = 0);
// fbank [B, frames, n_mels] fp32 -> half patch-major (used by forward_fbank)
int patchify(const float* fbank, int B, int frames, int n_mels, int patch, void* out_patch, int dtype,
             hipStream_t s);

}  // namespace avx
