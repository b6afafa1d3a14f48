// MFMA GEMM for torch.nn.Linear-shaped products:  out[m,n] = epi( sum_k A[m,k] * W[n,k] ).
//
// Replaces every Linear on the BEATs path (reference call sites: beats.py:350-359 patch-embed as a
// GEMM + post_extract_proj; backbone.py:531-533 q/k/v_proj, :572 out_proj, :368 fc1, :370 fc2) with
// the bias / DeepNorm-residual / exact-erf GELU / hook-tap work fused into the epilogue.
//
// Tiling (gfx950): 128x128 output tile per 256-thread workgroup (4 waves as 2x2, 64x64 per wave),
// BK = 64, v_mfma_f32_16x16x32_{f16,bf16}, fp32 accumulate.  The MFMA "A" operand is the WEIGHT tile
// and the "B" operand the ACTIVATION tile, i.e. the hardware computes D[n][m]; with the 16x16 C/D
// layout (col = lane&15, row = 4*(lane>>4)+reg) every lane then owns 4 CONSECUTIVE n of one row m,
// so the epilogue moves 16-byte (fp32) / 8-byte (half) vectors and bias/residual are vector loads.
// Both operands are K-contiguous, so both are staged the same way: HBM -> LDS by LDS-DMA
// (global_load_lds_dwordx4, 1 KiB per wave-instruction = 8 rows x 128 B) into a double buffer, one
// barrier per K-step, next tile's DMA in flight under the current tile's MFMAs.  LDS rows are 128 B;
// the 16-byte chunk c of row r lives in slot c ^ ((r>>1)&7) (applied on the DMA *source* address and
// on the ds_read_b128 address, never on the DMA destination) which makes every fragment read
// bank-conflict free for the 16x16x32 lane groups.
#include <stdlib.h>

#include "common.h"
#include "gemm_epi.h"

#ifndef GEMM_PEEL
#define GEMM_PEEL 1       // 1: K-tile 0 of every output tile is a copy of the loop body whose first MFMA per accumulator has C = 0 (no zeroing pass); 0: A/B
#endif
#ifndef GEMM_SADDR
#define GEMM_SADDR 0      // 1 (A/B builds): the streaming kernel's LDS-DMA addresses as scalar base + 32-bit lane offset instead of 64-bit pointers per lane.
                          // MEASURED (profiles/r04w_epilogue_instructions.txt): the six DMAs of a K-tile's main path lose their v_lshl_add_u64, eight registers
                          // come free -- and the step is 0.2 % SLOWER (every shape within +-0.4 %): the scalar unit now forms each base (s_mul / s_addc chains
                          // between the s_mov m0 writes).  Not the default.
#endif
#ifndef GEMM_LNA_PK
#define GEMM_LNA_PK 1     // 1: EPI 1's folded LayerNorm as packed FMAs on column pairs (0: four scalar fmaf, A/B)
#endif
#ifndef GEMM_GELU_H
#define GEMM_GELU_H 1     // 1: the streaming kernel's bias + GELU epilogue (half output) evaluates the degree-4 fit; 0: the degree-6 one (A/B)
#endif
#ifndef GEMM_HW_SAT
#define GEMM_HW_SAT 1      // 1: the kernels of this file set MODE.FP16_OVFL around their epilogues (NOT while their MFMAs run: common.h) and their f16 outputs saturate through it instead of a v_med3_f32 per element
#endif
#if GEMM_HW_SAT
#define HFROM(x) Half<T>::from_hw(x)
#else
#define HFROM(x) Half<T>::from(x)      // the software clamp (v_med3_f32 per element): the A side of profiles/r04u_fp16_ovfl.txt
#endif

namespace { template <int N> struct a_ic { static constexpr int value = N; }; }
namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * BK * 2;  // one operand tile: 128 rows x 64 halves = 16 KiB

typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

template <typename T, bool GLDS>
__global__ __launch_bounds__(256) void gemm_nt_kernel(avx::GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename Half<T>::v8 v8;
    typedef typename Half<T>::v4 v4;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int tiles_n = p.N / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int wr = wid >> 1, wc = wid & 1;  // wave owns n in [n0+64wr, +64), m in [m0+64wc, +64)

    const T* __restrict__ A = (const T*)p.A;
    const T* __restrict__ W = (const T*)p.W;
    // split-K (gridDim.y > 1, LDS-DMA form only): this workgroup takes K / gridDim.y of the contraction and leaves its fp32 partial tile in
    // p.splitk_ws[blockIdx.y][M][N]; splitk_epilogue_kernel adds the partials in order and applies the epilogue.  For the few-row, long-K
    // product of a single clip (fc2: 24 workgroups x 48 K-steps on 256 CUs).
    const int kofs = GLDS ? (int)blockIdx.y * (p.K / (int)gridDim.y) : 0;
    const int nk = (GLDS ? p.K / (int)gridDim.y : p.K) / BK;

    // ---- staging -------------------------------------------------------------------------
    // LDS-DMA: wave w fills rows [32w, 32w+32) of each operand tile with 4 instructions of 8 rows.
    auto stage_dma = [&](int st, int k0) __attribute__((always_inline)) {
        char* wbase = smem + st * (2 * TILE_BYTES);
        char* abase = wbase + TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rloc = wid * 32 + i * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ ((rloc >> 1) & 7);
            const int wrow = n0 + rloc;  // N % 128 == 0: always in range
            int arow = m0 + rloc;
            arow = arow < p.M ? arow : p.M - 1;
            const T* wsrc = W + (int64_t)wrow * p.ldw + k0 + chunk * 8;
            const T* asrc = A + (int64_t)arow * p.lda + k0 + chunk * 8;
            const int dst = (wid * 32 + i * 8) * 128;  // wave-uniform; hardware adds lane*16
            __builtin_amdgcn_global_load_lds((gptr_t*)wsrc, (lptr_t*)(wbase + dst), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t*)asrc, (lptr_t*)(abase + dst), 16, 0, 0);
        }
    };
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int st) __attribute__((always_inline)) {
        const char* wbase = smem + st * (2 * TILE_BYTES);
        const char* abase = wbase + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            v8 wf[4], af[4];
            const int chunk = (lane >> 4) + 4 * ks;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = wr * 64 + i * 16 + (lane & 15);
                wf[i] = *(const v8*)(wbase + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = wc * 64 + j * 16 + (lane & 15);
                af[j] = *(const v8*)(abase + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(wf[i], af[j], acc[i][j]);
        }
    };

    // ---- main loop: one barrier per K-step, next tile in flight during compute --------------
    if constexpr (GLDS) {
        stage_dma(0, kofs);
        for (int kt = 0; kt < nk; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my DMA pieces of tile kt have landed
            __syncthreads();  // everyone's pieces landed; everyone finished reading buffer (kt+1)&1
            if (kt + 1 < nk) stage_dma((kt + 1) & 1, kofs + (kt + 1) * BK);
            compute(kt & 1);
        }
    } else {
        // register staging (variant 1): same LDS image, written with swizzled ds_write_b128
        uint4 rw0, rw1, rw2, rw3, ra0, ra1, ra2, ra3;
#define AVX_LD(i, k0)                                                                          \
        {                                                                                      \
            const int c = tid + 256 * i;                                                       \
            const int rloc = c >> 3, chunk = c & 7;                                            \
            int arow = m0 + rloc;                                                              \
            arow = arow < p.M ? arow : p.M - 1;                                                \
            rw##i = *(const uint4*)(W + (int64_t)(n0 + rloc) * p.ldw + (k0) + chunk * 8);      \
            ra##i = *(const uint4*)(A + (int64_t)arow * p.lda + (k0) + chunk * 8);             \
        }
        // GemmArgs::a_scale (this form only, besides the skinny kernel): the A rows pass through registers here, so the per-(clip, input
        // channel) rescale of EfficientNet's squeeze-excitation is applied on the way (fp32 product rounded to the operand type: the
        // arithmetic of scale_channels_kernel) instead of in a pass of its own over the expanded tensor
#define AVX_ST(i, st, k0)                                                                      \
        {                                                                                      \
            const int c = tid + 256 * i;                                                       \
            const int rloc = c >> 3, chunk = c & 7;                                            \
            const int off = rloc * 128 + ((chunk ^ ((rloc >> 1) & 7)) << 4);                   \
            if (p.a_scale) {                                                                   \
                int arow = m0 + rloc;                                                          \
                arow = arow < p.M ? arow : p.M - 1;                                            \
                const float* sp = p.a_scale + (int64_t)(arow / p.a_scale_rows) * p.a_scale_ld + (k0) + chunk * 8; \
                const f32x4 s0 = *(const f32x4*)sp, s1 = *(const f32x4*)(sp + 4);              \
                v8 x = *(v8*)&ra##i;                                                           \
                _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                \
                    x[e] = HFROM((float)x[e] * s0[e]);                                 \
                    x[4 + e] = HFROM((float)x[4 + e] * s1[e]);                         \
                }                                                                              \
                ra##i = *(uint4*)&x;                                                           \
            }                                                                                  \
            *(uint4*)(smem + (st) * (2 * TILE_BYTES) + off) = rw##i;                           \
            *(uint4*)(smem + (st) * (2 * TILE_BYTES) + TILE_BYTES + off) = ra##i;              \
        }
        AVX_LD(0, 0) AVX_LD(1, 0) AVX_LD(2, 0) AVX_LD(3, 0)
        AVX_ST(0, 0, 0) AVX_ST(1, 0, 0) AVX_ST(2, 0, 0) AVX_ST(3, 0, 0)
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const bool more = kt + 1 < nk;
            const int k1 = (kt + 1) * BK;
            if (more) { AVX_LD(0, k1) AVX_LD(1, k1) AVX_LD(2, k1) AVX_LD(3, k1) }
            compute(kt & 1);
            if (more) { AVX_ST(0, (kt + 1) & 1, k1) AVX_ST(1, (kt + 1) & 1, k1) AVX_ST(2, (kt + 1) & 1, k1) AVX_ST(3, (kt + 1) & 1, k1) }
            __syncthreads();
        }
#undef AVX_LD
#undef AVX_ST
    }

    // ---- epilogue ---------------------------------------------------------------------------
    if (GLDS && (gridDim.y > 1 || p.post_ln_w)) {      // split-K (or a row-owning epilogue kernel behind it): the raw partial tile
        float* part = p.splitk_ws + (int64_t)blockIdx.y * p.M * p.N;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + wc * 64 + j * 16 + (lane & 15);
            if (m >= p.M) continue;
#pragma unroll
            for (int i = 0; i < 4; ++i) *(f32x4*)(part + (int64_t)m * p.N + n0 + wr * 64 + i * 16 + (lane >> 4) * 4) = acc[i][j];
        }
        return;
    }
    if (GEMM_HW_SAT) AVX_F16_SAT_BEGIN();      // every MFMA of this workgroup has been issued: the f16 conversions below saturate in hardware (common.h)
    AVX_CLAMP_TOKEN(inva);
    const float alpha = p.alpha;
    float ovf_mx = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + wc * 64 + j * 16 + (lane & 15);
        if (m >= p.M) continue;
        const bool zero_row = p.row_zero != nullptr && p.row_zero[m] != 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + wr * 64 + i * 16 + (lane >> 4) * 4;
            if (p.n_store > 0 && n >= p.n_store) continue;      // columns computed for the tile shape only (rows of the outputs are n_store wide)
            f32x4 v = acc[i][j];
            if (p.bias) v += *(const f32x4*)(p.bias + n);
            if (zero_row) v = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (p.out_raw) *(f32x4*)(p.out_raw + (int64_t)m * p.ldraw + n) = v;
            if (p.resid) {
                const f32x4 r = *(const f32x4*)(p.resid + (int64_t)m * p.ldr + n);
                v = r * alpha + v;
            } else if (p.resid_half) {
                const v4 rh = *(const v4*)((const T*)p.resid_half + (int64_t)m * p.ldrh + n);
                const f32x4 r = {(float)rh[0], (float)rh[1], (float)rh[2], (float)rh[3]};
                v = r * alpha + v;
            }
            if (p.gelu) v = act4_any(v, p.gelu, inva);
            if (p.out_f32) *(f32x4*)(p.out_f32 + (int64_t)m * p.ldo + n) = v;
            if (p.out_half) {
                if (p.half_scale != 0.f) v = v * p.half_scale;      // (GemmArgs::half_scale; wave-uniform)
                ovf_see4<T>(ovf_mx, v);
                v4 h;
                h[0] = HFROM(v[0]); h[1] = HFROM(v[1]);
                h[2] = HFROM(v[2]); h[3] = HFROM(v[3]);
                *(v4*)((T*)p.out_half + (int64_t)m * p.ldh + n) = h;
            }
        }
    }
    ovf_commit<T>(p.ovf, ovf_mx);
}


// split-K partials [S][M][N] -> the epilogue of gemm_nt_kernel on their sum (added in split order: reproducible).  One thread = 4 columns.
template <typename T>
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(avx::GemmArgs p, int S) {
    if (GEMM_HW_SAT) AVX_F16_SATURATE_ON();
    AVX_CLAMP_TOKEN(inva);
    typedef typename Half<T>::v4 v4;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int nq = p.N >> 2;
    float ovf_mx = 0.f;
    if (idx < (int64_t)p.M * nq) {
        const int m = (int)(idx / nq), n = (int)(idx - (int64_t)m * nq) * 4;
        f32x4 v = *(const f32x4*)(p.splitk_ws + (int64_t)m * p.N + n);
        for (int sp = 1; sp < S; ++sp) v += *(const f32x4*)(p.splitk_ws + ((int64_t)sp * p.M + m) * p.N + n);
        if (!(p.n_store > 0 && n >= p.n_store)) {
            if (p.bias) v += *(const f32x4*)(p.bias + n);
            if (p.row_zero != nullptr && p.row_zero[m] != 0) v = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (p.out_raw) *(f32x4*)(p.out_raw + (int64_t)m * p.ldraw + n) = v;
            if (p.resid) {
                const f32x4 r = *(const f32x4*)(p.resid + (int64_t)m * p.ldr + n);
                v = r * p.alpha + v;
            } else if (p.resid_half) {
                const v4 rh = *(const v4*)((const T*)p.resid_half + (int64_t)m * p.ldrh + n);
                const f32x4 r = {(float)rh[0], (float)rh[1], (float)rh[2], (float)rh[3]};
                v = r * p.alpha + v;
            }
            if (p.gelu) v = act4_any(v, p.gelu, inva);
            if (p.out_f32) *(f32x4*)(p.out_f32 + (int64_t)m * p.ldo + n) = v;
            if (p.out_half) {
                if (p.half_scale != 0.f) v = v * p.half_scale;
                ovf_see4<T>(ovf_mx, v);
                v4 h;
                h[0] = HFROM(v[0]); h[1] = HFROM(v[1]); h[2] = HFROM(v[2]); h[3] = HFROM(v[3]);
                *(v4*)((T*)p.out_half + (int64_t)m * p.ldh + n) = h;
            }
        }
    }
    ovf_commit<T>(p.ovf, ovf_mx);
}

// ... and with a LayerNorm of the finished rows (GemmArgs::post_ln_*): one wave per row, lane l holds columns 256 c + 4 l .. + 3 of chunk c.
// Same per-row arithmetic as layernorm_half_kernel (two-pass statistics in registers, (v - mean) * rstd * w + b).
template <typename T>
__global__ __launch_bounds__(256) void splitk_ln_epilogue_kernel(avx::GemmArgs p, int S) {
    if (GEMM_HW_SAT) AVX_F16_SATURATE_ON();
    typedef typename Half<T>::v4 v4;
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= p.M) return;                          // wave-uniform
    const int nch = p.N >> 8;
    f32x4 v[4];
    float ovf_mx = 0.f, s = 0.f;
    const bool zero_row = p.row_zero != nullptr && p.row_zero[m] != 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        v[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (c >= nch) continue;
        const int n = c * 256 + lane * 4;
        f32x4 a = *(const f32x4*)(p.splitk_ws + (int64_t)m * p.N + n);
        for (int sp = 1; sp < S; ++sp) a += *(const f32x4*)(p.splitk_ws + ((int64_t)sp * p.M + m) * p.N + n);
        if (p.bias) a += *(const f32x4*)(p.bias + n);
        if (zero_row) a = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (p.out_raw) *(f32x4*)(p.out_raw + (int64_t)m * p.ldraw + n) = a;
        if (p.resid) {
            const f32x4 r = *(const f32x4*)(p.resid + (int64_t)m * p.ldr + n);
            a = r * p.alpha + a;
        } else if (p.resid_half) {
            const v4 rh = *(const v4*)((const T*)p.resid_half + (int64_t)m * p.ldrh + n);
            const f32x4 r = {(float)rh[0], (float)rh[1], (float)rh[2], (float)rh[3]};
            a = r * p.alpha + a;
        }
        if (p.out_f32) *(f32x4*)(p.out_f32 + (int64_t)m * p.ldo + n) = a;
        if (p.out_half || p.post_ln_round) {
            ovf_see4<T>(ovf_mx, a);
            v4 h;
            h[0] = HFROM(a[0]); h[1] = HFROM(a[1]); h[2] = HFROM(a[2]); h[3] = HFROM(a[3]);
            if (p.out_half) *(v4*)((T*)p.out_half + (int64_t)m * p.ldh + n) = h;
            if (p.post_ln_round) a = (f32x4){(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
        }
        v[c] = a;
        s += (a[0] + a[1]) + (a[2] + a[3]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / (float)p.N;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (c >= nch) continue;
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[c][e] -= mean; q += v[c][e] * v[c][e]; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    const float rstd = 1.0f / sqrtf(q / (float)p.N + p.post_ln_eps);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (c >= nch) continue;
        const int n = c * 256 + lane * 4;
        const f32x4 w = *(const f32x4*)(p.post_ln_w + n), b = *(const f32x4*)(p.post_ln_b + n);
        f32x4 y;
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = v[c][e] * rstd * w[e] + b[e];
        if (p.post_ln_out_f32) *(f32x4*)(p.post_ln_out_f32 + (int64_t)m * p.post_ln_ldo + n) = y;
        if (p.post_ln_out_half) {
            v4 h;
            h[0] = HFROM(y[0]); h[1] = HFROM(y[1]); h[2] = HFROM(y[2]); h[3] = HFROM(y[3]);
            *(v4*)((T*)p.post_ln_out_half + (int64_t)m * p.post_ln_ldh + n) = h;
        }
    }
    ovf_commit<T>(p.ovf, ovf_mx);
}

// ---------------------------------------------------------------------------------------------
// The 256-tile pipeline (gemm256p_kernel below): 256x256x64 tile, 8 waves (2 along n x 4 along m, 128x64 outputs per wave), 128 KiB LDS
// (2 stages), staged by LDS-DMA in HALF-tiles that follow the order in which the waves finish with
// them, so three half-tiles are always in flight under the MFMAs (counted vmcnt, raw s_barrier).
//
// A K-tile is consumed in two phases of 32 MFMAs per wave:
//     A: (W0 x X0,X1)      B: (W1 x X0,X1)
// W0/W1 = the wave's first/second 64 weight rows, X0/X1 = its first/second 32 activation rows; a
// half-tile is those rows of ALL waves (128 rows x 64 k = 16 KiB = 2 DMA instructions per wave).
// Phase = [L: ds_read the fragments that change (A: 16 reads, B: 8), issue half-tile DMAs, retire my
// LDS reads] barrier [M: 32 MFMAs = 512 cycles] barrier.  Waves 4-7 run one barrier behind waves 0-3,
// so on every SIMD one wave is in its MFMA segment while its partner is in its (shorter) load segment.
// Half-tile release/refill schedule (region: last read in phase -> refilled with tile t+2 in phase):
//     W0, X0, X1: A(t) -> B(t)          W1: B(t) -> A(t+1)
// All LDS reads are retired (lgkmcnt(0)) BEFORE the barrier that ends an L segment, so a refill
// issued one phase later can never overtake a read (WAR), also across the one-barrier stagger.
// Tile t+1 is complete when its last half-tile W1(t+1) (issued in A(t)) has landed: the three
// half-tiles issued in B(t) = 6 DMA instructions are younger, hence vmcnt(6) in B(t), before the
// barrier that every wave must pass before any wave reads tile t+1 (RAW).
// ---------------------------------------------------------------------------------------------
// Clock stamps exist only in the diagnostic build (AVEX_AMD_DIAG=1 python -m avex_amd.build -> lib/libavexhip_diag.so, -DAVEX_DIAG):
// in the product library AVX_STAMPS_ON is the constant false and every stamp statement below is compiled out.
#ifdef AVEX_DIAG
__device__ unsigned long long g_gemm_stamps[4 * 8192];
__device__ int g_gemm_stamps_on = 0;
__device__ unsigned long long g_gemm_clk[2 * 8192];
__device__ unsigned long long g_gemm_kclk[256 * 64];
#define AVX_STAMPS_ON (g_gemm_stamps_on != 0)
#define AVX_STAMP(...) do { __VA_ARGS__; } while (0)
#else
#define AVX_STAMPS_ON false
#define AVX_STAMP(...) do { } while (0)
#endif
//   // s_memtime at the top of every K-tile of each workgroup's third tile, variant 5   // s_memtime (shader clock) at loop start / end, variant 5

constexpr int T2 = 256;
// Tile walk of the 256-tile kernels: consecutive logical tile ids (each XCD owns a contiguous range of them, and its 32 CUs
// run 32 consecutive ids at a time) go DOWN groups of GROUP_M row panels before moving one column to the right, so an XCD
// keeps a group's A panels (GROUP_M x 393 KB at K = 768; GROUP_M = 8) in its 4 MiB L2 while the weight column panels stream past once per
// group.  With the row-major walk every row panel re-fetched the whole weight matrix (4.7 MB for fc1, more than the L2), and
// the PMC passes showed 1.59 GB fetched per fc1 launch against 0.2 GB of activations (profiles/r01g_gemm_tile_order.txt).
static __device__ __forceinline__ void tile_coords(int tile, int tiles_m, int tiles_n, int GROUP_M, int& tm, int& tn) {
    const int per_group = GROUP_M * tiles_n;
    const int gid = tile / per_group;
    const int first_m = gid * GROUP_M;
    const int gsz = (tiles_m - first_m) < GROUP_M ? (tiles_m - first_m) : GROUP_M;
    const int r = tile - gid * per_group;
    tn = r / gsz;
    tm = first_m + (r - tn * gsz);
}

constexpr int STAGE2 = 2 * T2 * BK * 2;   // 65536: W tile (32 KiB) + X tile (32 KiB)

#define AVX_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define AVX_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

#define AVX_READ_W(h, st)                                                                      \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                            \
        const char* r = smem + (st) * STAGE2 + wfrag + (64 * (h) + 16 * i) * 128;              \
        wf[i][0] = *(const v8*)(r + foff0);                                                    \
        wf[i][1] = *(const v8*)(r + foff1);                                                    \
    }
#define AVX_READ_X(st)                                                                         \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                            \
        const char* r = smem + (st) * STAGE2 + xfrag + (16 * j) * 128;                         \
        xf[j][0] = *(const v8*)(r + foff0);                                                    \
        xf[j][1] = *(const v8*)(r + foff1);                                                    \
    }
// AVX_SNAKE: walk the 4 x 4 MFMA grid boustrophedon so that consecutive MFMAs share one operand register (the B fragment stays when
// the A fragment changes): operand toggling is worth a few percent of MFMA power (profiles/r01h_mfma_power.txt)
// GEMM_NT: the 256-tile kernels' output stores carry the non-temporal hint.  Outputs are hundreds of MB per launch and are read next by
// another kernel: letting them allocate in the 4 MiB L2 evicts the A / weight panels the next K-tiles need (profiles/r01h_gemm_nt.txt)
#ifndef GEMM_NT
#define GEMM_NT 1
#endif
template <typename V>
static __device__ __forceinline__ void st_out(V* ptr, const V& v, int nt) {
    static_assert(sizeof(V) == 16, "16-byte stores only");
    typedef int i32x4_st __attribute__((ext_vector_type(4)));
    // the hinted store is inline asm: written as a builtin next to the plain store, the two branches are merged into ONE plain store
    // (the !nontemporal metadata is dropped) and the hint silently disappears
    if (GEMM_NT && (nt & 1)) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(ptr), "v"(__builtin_bit_cast(i32x4_st, v)) : "memory");   // s_nop 1: the hazard recogniser does not see a store in an asm block; on gfx940+/gfx950 a VMEM store of more than 64 bits followed by a VALU write of its data registers needs TWO wait states
    else *ptr = v;
}
// (buf_rsrc, buf_st16, buf_ld16 -- raw-buffer access of the fast epilogues -- live in gemm_epi.h)
// AVEX_AMD_GEMM_NT: 0 no hints, 1 (default) non-temporal output stores, 5 = only for outputs wider than 768 columns (diagnostics)
static int gemm_nt_mode(const avx::GemmArgs& a) {
    static const int mode = getenv("AVEX_AMD_GEMM_NT") ? atoi(getenv("AVEX_AMD_GEMM_NT")) : 1;
    if ((mode & 4) && a.N <= 768) return 0;
    return mode & 1;
}
#ifndef AVX_SNAKE
#define AVX_SNAKE 1
#endif
// (KT0: the first K-tile of an output tile -- its first MFMA per accumulator takes a literal zero as C, so the 128 accumulator registers are
// never zeroed by v_mov_b32)
#define AVX_HALF(hw)                                                                           \
    __builtin_amdgcn_s_setprio(1);                                                             \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                           \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                              \
    _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) {                                         \
        const int j = AVX_SNAKE && (i & 1) ? 3 - jj : jj;                                      \
        acc[4 * (hw) + i][j] = mfma16(wf[i][ks], xf[j][ks], (KT0 && ks == 0) ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[4 * (hw) + i][j]);             \
    }                                                                                          \
    __builtin_amdgcn_s_setprio(0);
#define AVX_BAR()                                   \
    __builtin_amdgcn_sched_barrier(0);              \
    __builtin_amdgcn_s_barrier();                   \
    __builtin_amdgcn_sched_barrier(0);


// (seg8_sum, seg8_sum2, stats8 -- the row-statistics helpers both residual epilogues use -- live in gemm_epi.h: gemm_row.hip shares them)

// ---------------------------------------------------------------------------------------------
// The 256-tile kernel: the half-tile pipeline above as ONE CONTINUOUS K STREAM over a persistent workgroup's tiles (one
// workgroup per CU).  The last iterations of a tile issue the DMA for the next tile's first K-tiles as if they were K-tiles
// nk, nk + 1 of the same product (the source pointers switch in the middle of iteration nk - 2), the epilogue's transpose slabs
// live in the 32 KiB of LDS above the two stages, per-tile row / column vectors (bias, folded-LayerNorm vectors) arrive by
// LDS-DMA one tile ahead, the two wave groups are re-aligned around the epilogue and the next tile's first counted wait skips
// the epilogue's stores.  (History: the tile-per-workgroup form of this pipeline, gemm256_kernel, was the default until round 1's
// last day and the home of the folded-LayerNorm epilogues until round 3; profiles/r01h_gemm_stream.txt has the comparison.)
//
// Epilogues (template EPI, flags LN):
//   EPI 1  half output = act(acc + bias)                      LN bit 0: the A rows are raw y, LayerNorm folded in (GemmArgs::ln_rows)
//   EPI 2  half output = resid * alpha + acc + bias           LN bit 0: resid = LayerNorm(lnr_y) on the fly; bit 1: stats_out
//   EPI 0  everything avexhip_gemm can ask for (fp32 / raw outputs, fp32 residual, masked rows, missing bias ...), same arithmetic
//          and operation order as the two fast forms where they overlap (bit-identical rows), conservative waits.
// ---------------------------------------------------------------------------------------------
constexpr int LDS5 = 2 * STAGE2 + 32768;
constexpr int L5_BIAS = 8 * 2304;            // EPI 1: [2][256] floats, bias of the current / next tile
constexpr int L5_LNS = L5_BIAS + 2048;       // EPI 1 + LN: [2][256] floats, ln_s of the current / next tile's columns
constexpr int L5_ROWS = L5_LNS + 2048;       // EPI 1 + LN: [2][256] float2, (rstd, -mu rstd) of the current / next tile's rows
static_assert(L5_ROWS + 4096 <= 32768, "EPI 1 scratch above the stages");
#ifndef GEMM_NOSTORE
#define GEMM_NOSTORE 0   // diagnostic builds: 1 drops the half-only epilogue's global stores (is the tile start waiting for them?)
#endif
#ifndef GEMM_XCD_WALK
#define GEMM_XCD_WALK 0
#endif
#ifndef GEMM_NOEPI
#define GEMM_NOEPI 0
#endif
#ifndef GEMM_COL_WALK
#define GEMM_COL_WALK 0     // 1: AVEX_AMD_GEMM_TILE_ORDER=-n selects the column-group walk (A/B builds; the scalar code of a third walk in
                            // set_tile is kept out of the default kernel: it sits inside the K loop's second-to-last iteration)
#endif
#ifndef GEMM_GELU_CHAINS
#define GEMM_GELU_CHAINS 0  // EPI 1: 1 = the activation of a 16-row chunk as 8 chains side by side (gelu_erf2xN); fewer stall cycles, same
                            // instructions: fc1 -0.3 %, QKV +1.1 % (profiles/r04h_epilogue_ab2.txt) -- under the power cap stall cycles are not the currency
#endif
#ifndef GEMM_EPI2_EARLY
#define GEMM_EPI2_EARLY 0
#endif
#ifndef GEMM_W_POLICY
#define GEMM_W_POLICY 0
#endif
#ifndef GEMM_A_POLICY
#define GEMM_A_POLICY 0
#endif
#ifndef GEMM_EPI1_SWAP
#define GEMM_EPI1_SWAP 0   // EPI 1: 0 = transpose through a private LDS slab (default), 1 = in registers with v_permlane16_swap (A/B builds; measured SLOWER, see below)
#endif

template <typename T, int EPI, int LN, int ACT>
__global__ __launch_bounds__(512) void gemm256p_kernel(avx::GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename Half<T>::v8 v8;
    typedef typename Half<T>::v4 v4;
    static_assert(EPI >= 0 && EPI <= 2 && LN >= 0 && LN <= 3 && (EPI != 1 || LN <= 1), "epilogue selector");      // (EPI 0: LN is the pool mode)
    // ACT (EPI 1 only): 0 none, 1 exact-erf GELU, 2 SiLU, as a template argument.  As the run-time value p.gelu it put a uniform branch around
    // every 4-element group of the epilogue: one basic block per group, so the six dependent packed FMAs of a group's polynomial ran
    // as a bare latency chain (profiles/r04a_epilogue_isa.txt: fc1's epilogue 7.7 us for waves 0-3 and 11 us for waves 4-7 per tile).
    static_assert(ACT >= 0 && ACT <= 2 && (EPI == 1 || ACT == 0), "activation selector");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3;      // wm also selects the stagger group (waves 4-7 lag one barrier)
    const int tiles_n = p.N / T2;
    const int tiles_m = (p.M + T2 - 1) / T2;
    const int ntiles = tiles_m * tiles_n;
    const T* __restrict__ A = (const T*)p.A;
    const T* __restrict__ W = (const T*)p.W;
    const int nk = p.K / BK;   // >= 2 (launcher)
    __builtin_assume(nk >= 2);      // (without it the residual epilogue's kernels zero the 128 accumulators twice per tile: once more on the path around an empty K loop)
    // blocks sharing blockIdx % 8 share an XCD: in round `it` they take 32 consecutive tiles
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = (gridDim.x + 7) >> 3;
    int m0 = 0, n0 = 0;

    // operand addresses of the LDS-DMA: a UNIFORM base per tile (the tile's first W row / A row: scalar registers) + a 32-bit byte offset per
    // lane, so that the instruction takes the base from SGPRs and the K-tile's advance is scalar arithmetic.  As per-lane 64-bit pointers
    // (GEMM_SADDR 0) every DMA cost a v_lshl_add_u64: eight 64-bit vector adds per K-tile and wave, and sixteen registers instead of eight.
#if GEMM_SADDR
    unsigned wsrc[2][2], xsrc[2][2];
#else
    const T* wsrc[2][2];
    const T* xsrc[2][2];
#endif
    int wdst[2][2], xdst[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            wdst[h][q] = (128 * q + 64 * h + 8 * wid) * 128;
            xdst[h][q] = T2 * BK * 2 + (128 * q + 64 * (wid >> 2) + 32 * h + 8 * (wid & 3)) * 128;
        }
    // grouped walk where a group of 8 A panels fits the L2 (K < 2048: QKV, out_proj, fc1); measured against the row-major walk: fc1 +3 %,
    // out_proj +3 %, QKV +-0, whole step +1.4 %; HBM fetch per fc1 launch 1.57 -> 0.97 GB (profiles/r01g_gemm_tile_order.txt).
    // At K = 3072 a group would be 12.6 MB: row-major there.
    const bool grouped = p.tile_order >= 2 || (p.tile_order == 0 && p.K < 2048);
    const int group_m = p.tile_order >= 2 ? p.tile_order : 8;
    // tile_order < 0: COLUMN-group walk, -tile_order column tiles per group.  Tile ids run (column group, row panel, column inside the
    // group) and every XCD owns a contiguous eighth of them (xw below): an XCD keeps one group of weight panels (e.g. 6 x 393 KB) for
    // hundreds of tiles while the A row panels stream past ONCE per group -- the row-group walk above re-fetches a group's A panels for
    // every round of column tiles once 32 concurrent tiles have pushed 4.7 MB through a 4 MiB L2 (roofline.traffic_by_shape).
    const bool colwalk = GEMM_COL_WALK && p.tile_order < 0;
    const int group_n = colwalk ? (-p.tile_order < tiles_n ? -p.tile_order : tiles_n) : 1;
    auto set_tile = [&](int tile) __attribute__((always_inline)) {
        int tm, tn;
        if (colwalk) {
            const int per_group = tiles_m * group_n;
            const int gid = tile / per_group;
            const int first_n = gid * group_n;
            const int gsz = (tiles_n - first_n) < group_n ? (tiles_n - first_n) : group_n;
            const int r = tile - gid * per_group;
            tm = r / gsz;
            tn = first_n + (r - tm * gsz);
        } else if (grouped) tile_coords(tile, tiles_m, tiles_n, group_m, tm, tn);
        else { tm = tile / tiles_n; tn = tile - tm * tiles_n; }
        m0 = tm * T2; n0 = tn * T2;
#if GEMM_SADDR
        m0 = __builtin_amdgcn_readfirstlane(m0); n0 = __builtin_amdgcn_readfirstlane(n0);      // scalar registers, said explicitly: the DMA bases are formed from them
        const int last = p.M - 1 - m0;                       // rows past M read row M - 1 (their outputs are never stored)
#endif
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int wr = 128 * q + 64 * h + 8 * wid + (lane >> 3);
                const int xr = 128 * q + 64 * (wid >> 2) + 32 * h + 8 * (wid & 3) + (lane >> 3);
#if GEMM_SADDR
                wsrc[h][q] = (unsigned)(((int64_t)wr * p.ldw + (((lane & 7) ^ ((wr >> 1) & 7)) << 3)) * 2);
                const int ar = xr < last ? xr : last;
                xsrc[h][q] = (unsigned)(((int64_t)ar * p.lda + (((lane & 7) ^ ((xr >> 1) & 7)) << 3)) * 2);
#else
                wsrc[h][q] = W + (int64_t)(n0 + wr) * p.ldw + (((lane & 7) ^ ((wr >> 1) & 7)) << 3);
                int arow = m0 + xr;
                arow = arow < p.M ? arow : p.M - 1;
                xsrc[h][q] = A + (int64_t)arow * p.lda + (((lane & 7) ^ ((xr >> 1) & 7)) << 3);
#endif
            }
    };
    // GEMM_W_POLICY / GEMM_A_POLICY (A/B builds): cache-policy bits of the two operand streams' LDS-DMA (0 default, 2 = nt: stream, evict first).
    // Measured (profiles/r04g_epilogue_ab.txt, r04h_epilogue_ab2.txt): W nt 6 - 24 % slower on every shape; A nt 13 - 15 % slower on QKV / fc1
    // and 8 % FASTER on out_proj stand-alone -- but chosen per product inside the step (N <= 768 only) it still LOSES 0.8 %: what it
    // pushes out of the caches are the next kernels' inputs.  (As a run-time branch around each DMA pair it also cost the loop 1.5 %.)
    auto dma_w = [&](int h, int kt, int stg) __attribute__((always_inline)) {
        char* base = smem + stg * STAGE2;
#if GEMM_SADDR
        const char* wk = (const char*)(W + (int64_t)n0 * p.ldw) + (int64_t)kt * (BK * 2);      // from the (scalar) tile coordinates every time: as a loop-carried pointer the base ends up in vector registers
        // (the W offsets never change: left visible, their zero-extension is hoisted out of the loop as a 64-bit register pair and the DMA is
        // back to a 64-bit vector address + v_lshl_add_u64; the empty asm keeps the extension at the instruction, where it is free)
        unsigned w0 = wsrc[h][0], w1 = wsrc[h][1];
        asm volatile("" : "+v"(w0), "+v"(w1));
        __builtin_amdgcn_global_load_lds((gptr_t*)(wk + w0), (lptr_t*)(base + wdst[h][0]), 16, 0, GEMM_W_POLICY);
        __builtin_amdgcn_global_load_lds((gptr_t*)(wk + w1), (lptr_t*)(base + wdst[h][1]), 16, 0, GEMM_W_POLICY);
#else
        __builtin_amdgcn_global_load_lds((gptr_t*)(wsrc[h][0] + kt * BK), (lptr_t*)(base + wdst[h][0]), 16, 0, GEMM_W_POLICY);
        __builtin_amdgcn_global_load_lds((gptr_t*)(wsrc[h][1] + kt * BK), (lptr_t*)(base + wdst[h][1]), 16, 0, GEMM_W_POLICY);
#endif
    };
    auto dma_x = [&](int h, int kt, int stg) __attribute__((always_inline)) {
        char* base = smem + stg * STAGE2;
#if GEMM_SADDR
        const char* ak = (const char*)(A + (int64_t)m0 * p.lda) + (int64_t)kt * (BK * 2);
        unsigned x0 = xsrc[h][0], x1 = xsrc[h][1];
        asm volatile("" : "+v"(x0), "+v"(x1));
        __builtin_amdgcn_global_load_lds((gptr_t*)(ak + x0), (lptr_t*)(base + xdst[h][0]), 16, 0, GEMM_A_POLICY);
        __builtin_amdgcn_global_load_lds((gptr_t*)(ak + x1), (lptr_t*)(base + xdst[h][1]), 16, 0, GEMM_A_POLICY);
#else
        __builtin_amdgcn_global_load_lds((gptr_t*)(xsrc[h][0] + kt * BK), (lptr_t*)(base + xdst[h][0]), 16, 0, GEMM_A_POLICY);
        __builtin_amdgcn_global_load_lds((gptr_t*)(xsrc[h][1] + kt * BK), (lptr_t*)(base + xdst[h][1]), 16, 0, GEMM_A_POLICY);
#endif
    };
    constexpr bool fast_half = EPI == 1, fast_resid = EPI == 2;
    constexpr bool LNA = EPI == 1 && (LN & 1), LNR = EPI == 2 && (LN & 1), STATS = EPI == 2 && (LN & 2);
    float* ldsbias = (float*)(smem + 2 * STAGE2 + L5_BIAS);
    float* ldslns = (float*)(smem + 2 * STAGE2 + L5_LNS);
    float* ldsrows = (float*)(smem + 2 * STAGE2 + L5_ROWS);
    // EPI 1: the tile's bias row (wave 0), with the fold its ln_s row (wave 1) and the (rstd, -mu rstd) pairs of its 256 rows (waves 2, 3)
    // go to LDS by LDS-DMA, at most one wave instruction (1 KiB) per wave, one tile ahead: a global load in the epilogue could only be
    // waited for together with the next tile's DMAs that are in flight by then
    auto dma_aux = [&](int n0b, int m0b, int par) __attribute__((always_inline)) {
        if (fast_half) {
            if (wid == 0) __builtin_amdgcn_global_load_lds((gptr_t*)(p.bias + n0b + lane * 4), (lptr_t*)(ldsbias + par * 256), 16, 0, 0);
            if (LNA && wid == 1) __builtin_amdgcn_global_load_lds((gptr_t*)(p.ln_s + n0b + lane * 4), (lptr_t*)(ldslns + par * 256), 16, 0, 0);
            if (LNA && (wid == 2 || wid == 3)) {
                int r = m0b + 128 * (wid - 2) + 2 * lane;                 // two rows (16 bytes) per lane
                const int last = (p.M - 1) & ~1;                           // the pair that holds row M - 1 (ln_rows is readable up to M rounded up to even); rows past M are never stored
                r = r < last ? r : last;
                __builtin_amdgcn_global_load_lds((gptr_t*)(p.ln_rows + 2 * (int64_t)r), (lptr_t*)(ldsrows + par * 512 + (wid - 2) * 256), 16, 0, 0);
            }
        }
    };
    auto tile_prologue = [&]() __attribute__((always_inline)) {   // 14 DMA instructions
        dma_w(0, 0, 0); dma_x(0, 0, 0); dma_x(1, 0, 0); dma_w(1, 0, 0);
        dma_w(0, 1, 1); dma_x(0, 1, 1); dma_x(1, 1, 1);
    };
    const int sw = (lane >> 1) & 7;
    const int foff0 = (lane & 15) * 128 + ((((lane >> 4)) ^ sw) << 4);   // k-substep 0 (swizzle depends on the lane only: rows start at multiples of 16)
    const int foff1 = foff0 ^ 64;                                        // k-substep 1: chunk + 4
    const int wfrag = (wm * 128) * 128;
    const int xfrag = T2 * BK * 2 + (wn * 64) * 128;

    f32x4 acc[8][4];
    v8 wf[4][2], xf[4][2];
    unsigned long long ovf_lanes = 0ull;      // range alarm: lanes that clipped a value in any tile so far (scalar registers)

    // GEMM_XCD_WALK 1 (A/B builds): each XCD owns a CONTIGUOUS eighth of the logical tile ids and walks it 32 ids per round, so that a
    // group's A panels could stay in that XCD's L2 from one round to the next.  MEASURED: no change in time (+-0.1 % on the step) and
    // none in L2-miss bytes (QKV 806 -> 860 MB, fc1 879 -> 997 MB per launch, scripts/pmc_gemm_shapes.sh): 32 concurrent 256 x 256 tiles
    // touch at least 11.3 operand panels of 393 KB (K = 768) per round, 4.4 MB against 4 MB of L2, so under LRU nothing survives from
    // one round to the next whichever XCD runs it; the fetch counter sits at rounds x that footprint (the floor of this tiling), and
    // what it counts is L2 <-> fabric traffic, of which the 256 MB Infinity Cache absorbs the A re-reads (A is 195 MB).
    const bool xw = GEMM_XCD_WALK || colwalk;
    const int t_lo = xw ? (int)(((int64_t)ntiles * xcd) >> 3) : 0;
    const int t_hi = xw ? (int)(((int64_t)ntiles * (xcd + 1)) >> 3) : ntiles;
    int tile = xw ? t_lo + slot : (0 * 8 + xcd) * per_xcd + slot;
    if (tile >= t_hi) return;
    set_tile(tile);
    dma_aux(n0, m0, 0);
    tile_prologue();
    const bool stamp_on = AVX_STAMPS_ON && tid == 0;
    AVX_VMCNT(6);
    AVX_BAR();
    if (wm == 1) { AVX_BAR(); }      // stagger: waves 4-7 run one barrier behind, for the whole tile walk
    int g0 = 0;                      // global K-tile index of the tile's first K-tile: its stage parity
    // EPI 1 (and EPI 2 in GEMM_EPI2_EARLY builds): the epilogue's trailing stores are COUNTED PAST by the next tile's first wait instead of
    // waited for (see the B phase of K-tile 0 below).  Their number is exact since the stores became raw-buffer stores that always issue
    // (rows past M are dropped by the bounds check, not skipped by an exec mask): 16 output stores per wave, EPI 2 with statistics 16 more,
    // interleaved -- whichever form, at least the last 16 vector-memory operations before the new tile's DMAs are epilogue operations
    // that need not have retired.  For EPI 2 it measured level (out_proj) to 1 % slower (fc2): profiles/r04h_epilogue_ab2.txt; not the default.
    constexpr bool early_w1 = (EPI == 1 || (GEMM_EPI2_EARLY && EPI == 2)) && !GEMM_NOSTORE && !GEMM_NOEPI;

    // One continuous K stream over this workgroup's tiles: the DMA for the next tile's first K-tiles is issued by the
    // LAST iterations of the current tile exactly as if they were K-tiles nk, nk + 1 of the same product (the source
    // pointers switch to the next tile in the middle of iteration nk - 2, after the last use of the current ones), so the
    // pipeline never drains: no per-tile prologue, no workgroup turnaround, and the epilogue (private LDS slabs above the
    // stages, no barrier) runs with two K-tiles of the next tile already in flight.
    for (int it = 0;; ++it) {
#if !GEMM_PEEL
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#endif
        const bool stamp = stamp_on && tile < 8192;
        const int em0 = m0, en0 = n0;
        const int next_tile = xw ? t_lo + (it + 1) * per_xcd + slot : ((it + 1) * 8 + xcd) * per_xcd + slot;
        const bool has_next = next_tile < t_hi;
        AVX_STAMP(if (stamp) { g_gemm_stamps[4 * tile + 0] = g_gemm_stamps[4 * tile + 1] = __builtin_amdgcn_s_memrealtime(); g_gemm_clk[2 * tile] = __builtin_amdgcn_s_memtime(); });
        auto k_tile = [&](int kt, auto kt0_tag) __attribute__((always_inline)) {
            constexpr bool KT0 = decltype(kt0_tag)::value;
            const int st = (g0 + kt) & 1;
            AVX_STAMP(if (stamp && it == 2 && kt < 63 && blockIdx.x < 256) g_gemm_kclk[blockIdx.x * 64 + kt] = __builtin_amdgcn_s_memtime(););
            // A: W0 x (X0, X1); refill W1 of K-tile kt+1 (other stage, last read in B of K-tile kt-1)
            AVX_READ_X(st);
            AVX_READ_W(0, st);
            if (kt + 1 < nk) { if (!(early_w1 && kt == 0 && it > 0)) dma_w(1, kt + 1, st ^ 1); }   // (issued before the epilogue, see below)
            else if (has_next) dma_w(1, 0, st ^ 1);                  // pointers already switched (below, one iteration ago)
            if (kt == nk - 2 && has_next) set_tile(next_tile);       // last use of this tile's pointers was the line above
            // next tile's row / column vectors: in the LAST K-tile, i.e. at least four barriers into this tile, when the lagging wave group
            // has left the previous tile's epilogue (which read the slot this overwrites); they are older than the six DMAs of the
            // B phase below, so each issuing wave's vmcnt(6) there and the two barriers that follow publish them before the epilogue
            if (kt == nk - 1 && has_next) dma_aux(n0, m0, (it + 1) & 1);
            AVX_LGKM0();
            AVX_BAR();
            AVX_HALF(0);
            AVX_BAR();
            // B: W1 x (X0, X1); refill W0, X0, X1 with K-tile kt+2; make K-tile kt+1 visible
            AVX_READ_W(1, st);
            if (kt + 2 < nk) {
                dma_w(0, kt + 2, st); dma_x(0, kt + 2, st); dma_x(1, kt + 2, st);
                // first K-tile after an epilogue: the epilogue's last 16 vector-memory operations (its trailing stores) sit between K-tile 1's
                // DMAs (older) and these six; count past them instead of waiting for them to retire
                if (early_w1 && kt == 0 && it > 0) { AVX_VMCNT(22); }
                else { AVX_VMCNT(6); }
            } else if (has_next) {
                dma_w(0, kt + 2 - nk, st); dma_x(0, kt + 2 - nk, st); dma_x(1, kt + 2 - nk, st);
                AVX_VMCNT(6);
            } else {
                AVX_VMCNT(0);
            }
            AVX_LGKM0();
            AVX_BAR();
            AVX_HALF(1);
            AVX_BAR();
        
        };
#if GEMM_PEEL
        k_tile(0, a_ic<1>{});                              // K-tile 0: C = 0
        for (int kt = 1; kt < nk; ++kt) k_tile(kt, a_ic<0>{});
#else
        for (int kt = 0; kt < nk; ++kt) k_tile(kt, a_ic<0>{});
#endif
        AVX_STAMP(if (stamp && it == 2 && nk < 64 && blockIdx.x < 256) g_gemm_kclk[blockIdx.x * 64 + nk] = __builtin_amdgcn_s_memtime(););
        // Re-align the two wave groups for the epilogue: left staggered, the lagging group cannot pass its last loop barrier before the
        // leading group reaches the next tile's first one, i.e. the two epilogues would run one after the other.
        if (wm == 0) { AVX_BAR(); }
        // every wave is past the loop now: the W1 half of the stage K-tile 1 of the next tile goes to is free.  Issued here, ahead of the
        // epilogue's stores, the first K-tile's counted wait does not have to wait for those stores.
        if (early_w1 && has_next) dma_w(1, 1, ((g0 + nk) & 1) ^ 1);
        g0 += nk;
        AVX_STAMP(if (stamp) { g_gemm_stamps[4 * tile + 2] = __builtin_amdgcn_s_memrealtime(); g_gemm_clk[2 * tile + 1] = __builtin_amdgcn_s_memtime(); });

        // ---- epilogue ---------------------------------------------------------------------------
        // Each wave transposes its 128(n) x 64(m) accumulators through a private LDS slab in 16-row chunks, so that global traffic is
        // row-contiguous: a lane owns 8 consecutive n of one row (16-byte f16 / 2 x 16-byte fp32 vectors), 8 lanes cover a 128-byte
        // line, a wave instruction writes 8 full lines.  (The direct-from-accumulator form, 8-byte stores, ran the store path at ~2 TB/s.)
        // the epilogue's lane constants (slab addresses, row / column offsets) are derived from an opaque copy of the lane id: computed from
        // `lane` itself they are loop-invariant, get hoisted above the tile loop, and then live -- spilled -- through the K loop
        int le = lane;
        asm volatile("" : "+v"(le));
        const int er = le >> 3, ec = le & 7, lc = le & 15, lg = le >> 4;
        float ovf_mx = 0.f;
        if (GEMM_HW_SAT) AVX_F16_SAT_BEGIN();      // MODE.FP16_OVFL for the epilogue's conversions only: set, the MFMAs drop NaN operands (common.h)
        AVX_CLAMP_TOKEN(inva);

        if constexpr (GEMM_NOEPI && (EPI == 1 || EPI == 2)) {
            // diagnostic build: NO epilogue at all (the accumulators are only kept alive) -- the upper bound of what any scheme that hides
            // the epilogue under the next tile's MFMAs could reach, with the epilogue's energy taken away as well (profiles/r04n_noepi.txt)
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[i][j]));
        } else if constexpr (fast_half) {
            // bias / GELU (/ the folded LayerNorm of the A rows) in the accumulator layout; the slab holds the converted halves
            // (ds_write_b64 of 4 halves, ds_read_b128 of 8); no uniform branches inside, so the scheduler interleaves the
            // independent GELU chains of a 64-column slice
            const float* lb = ldsbias + (it & 1) * 256 + wm * 128;
            const float* lsv = ldslns + (it & 1) * 256 + wm * 128;
            float2 rst[4];
            if (LNA) {
#pragma unroll
                for (int j = 0; j < 4; ++j) rst[j] = ((const float2*)(ldsrows + (it & 1) * 512))[wn * 64 + 16 * j + lc];      // (rstd, -mu rstd) of tile row 64 wn + 16 j + lc
            }
#if GEMM_EPI1_SWAP
            // (A/B form, NOT the default: profiles/r04g_epilogue_ab.txt -- QKV 6.5 % slower than the slab form below, fc1 1.2 % faster; the
            //  residual epilogue built the same way was 14 % / 4 % slower on out_proj / fc2.  What it loses is the store shape: 16 rows x 64
            //  contiguous bytes per instruction instead of 8 rows x 128, i.e. half lines.)
            // Register transpose instead of an LDS round trip.  After the arithmetic a lane holds, for each of the four 16-column MFMA tiles
            // i of a 16-row chunk, 4 consecutive columns (16 i + 4 lg ..) of row lc as two registers of halves.  One v_permlane16_swap per
            // register of a tile PAIR (2 p, 2 p + 1) exchanges the odd rows of 16 lanes of the first tile with the even rows of the second:
            // a lane then owns 8 CONSECUTIVE columns of row lc -- lanes lg = 0, 2 the two halves of tile 2 p, lanes lg = 1, 3 those of tile
            // 2 p + 1 -- i.e. one 16-byte store, 64 contiguous bytes per row and instruction (the fabric's request size).  Per chunk:
            // 4 swaps in place of 4 ds_write_b64 + 2 ds_read_b128 + two lgkmcnt(0) waits (scripts/micro/permlane16.hip checks the rows).
            const int ldh = (int)p.ldh;
            const int vrows = p.M - (em0 + wn * 64);
            const __amdgpu_buffer_rsrc_t obuf = buf_rsrc((const T*)p.out_half + (int64_t)(em0 + wn * 64) * p.ldh + en0 + wm * 128,
                                                         vrows > 0 ? (unsigned)(vrows < 64 ? vrows : 64) * (unsigned)ldh * 2u : 0u);
            const int ovoff = (lc * ldh + 16 * (lg & 1) + 8 * (lg >> 1)) * 2;
#pragma unroll
            for (int ih = 0; ih < 2; ++ih) {
                f32x4 bv[4], sv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) bv[i] = *(const f32x4*)(lb + 64 * ih + 16 * i + 4 * lg);
                if (LNA) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) sv[i] = *(const f32x4*)(lsv + 64 * ih + 16 * i + 4 * lg);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x2 v[8];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            // LayerNorm of the A rows folded in: rstd * acc + ((-mu rstd) * s[n] + bias'[n])
                            if (LNA) v[2 * i + (e >> 1)][e & 1] = __builtin_fmaf(rst[j].x, acc[4 * ih + i][j][e], __builtin_fmaf(rst[j].y, sv[i][e], bv[i][e]));
                            else v[2 * i + (e >> 1)][e & 1] = acc[4 * ih + i][j][e] + bv[i][e];
                        }
                    }
                    if constexpr (ACT == 1) gelu_erf2xN<8>(v);
                    else if constexpr (ACT == 2) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] = silu2(v[q]);
                    }
                    unsigned hw[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        ovf_see<T>(ovf_mx, v[q][0], v[q][1]);
                        typename Half<T>::v4 h2 = {HFROM(v[q][0]), HFROM(v[q][1]), HFROM(0.f), HFROM(0.f)};
                        hw[q] = __builtin_bit_cast(uint2, h2).x;
                    }
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) {
                        const auto s0 = __builtin_amdgcn_permlane16_swap(hw[4 * pr], hw[4 * pr + 2], false, false);          // columns 0, 1 of the lane's four
                        const auto s1 = __builtin_amdgcn_permlane16_swap(hw[4 * pr + 1], hw[4 * pr + 3], false, false);      // columns 2, 3
                        const i32x4_buf o = {(int)s0[0], (int)s1[0], (int)s0[1], (int)s1[1]};
                        if (!GEMM_NOSTORE) buf_st16<GEMM_NT ? 2 : 0>(o, obuf, ovoff + (16 * j * ldh + 64 * ih + 32 * pr) * 2);
                    }
                }
            }
#else
            constexpr int HP_LD = 72;     // halves per slab row (64 n + 8 pad = 144 B)
            T* slab = (T*)(smem + 2 * STAGE2 + wid * (16 * HP_LD * 2));
            // this wave's 64 rows x 128 columns of the output as a raw buffer that ends with the last valid row (see buf_rsrc)
            const int ldh = (int)p.ldh;
            const int vrows = p.M - (em0 + wn * 64);
            const __amdgpu_buffer_rsrc_t obuf = buf_rsrc((const T*)p.out_half + (int64_t)(em0 + wn * 64) * p.ldh + en0 + wm * 128,
                                                         vrows > 0 ? (unsigned)(vrows < 64 ? vrows : 64) * (unsigned)ldh * 2u : 0u);
            const int ovoff = (er * ldh + 8 * ec) * 2;
#pragma unroll
            for (int ih = 0; ih < 2; ++ih) {
                f32x4 bv[4], sv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) bv[i] = *(const f32x4*)(lb + 64 * ih + 16 * i + 4 * lg);
                if (LNA) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) sv[i] = *(const f32x4*)(lsv + 64 * ih + 16 * i + 4 * lg);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#if GEMM_GELU_CHAINS
                    // the chunk's 16 values per lane as 8 pairs, so that the activation runs as 8 chains side by side (gelu_erf2xN)
                    f32x2 v[8];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            // LayerNorm of the A rows folded in: rstd * acc + ((-mu rstd) * s[n] + bias'[n])
                            if (LNA) v[2 * i + (e >> 1)][e & 1] = __builtin_fmaf(rst[j].x, acc[4 * ih + i][j][e], __builtin_fmaf(rst[j].y, sv[i][e], bv[i][e]));
                            else v[2 * i + (e >> 1)][e & 1] = acc[4 * ih + i][j][e] + bv[i][e];
                        }
                    }
                    if constexpr (ACT == 1) gelu_erf2xN<8>(v);
                    else if constexpr (ACT == 2) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] = silu2(v[q]);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        ovf_see<T>(ovf_mx, v[2 * i][0], v[2 * i][1]);
                        ovf_see<T>(ovf_mx, v[2 * i + 1][0], v[2 * i + 1][1]);
                        v4 h;
                        h[0] = HFROM(v[2 * i][0]); h[1] = HFROM(v[2 * i][1]); h[2] = HFROM(v[2 * i + 1][0]); h[3] = HFROM(v[2 * i + 1][1]);
                        *(v4*)(slab + lc * HP_LD + 16 * i + 4 * lg) = h;
                    }
#else
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        f32x4 v;
                        if (LNA) {
                            // LayerNorm of the A rows folded in: rstd * acc + ((-mu rstd) * s[n] + bias'[n]).  Written on column PAIRS: as four
                            // scalar fmaf the compiler kept them scalar (256 v_fma_f32 per 128 outputs of a lane: half of QKV's epilogue arithmetic);
                            // v_pk_fma_f32 with the row's two scalars broadcast does a pair per slot.  Same operations: same bits.
#if GEMM_LNA_PK
                            const f32x2 rx = {rst[j].x, rst[j].x}, ry = {rst[j].y, rst[j].y};
                            const f32x4 a4 = acc[4 * ih + i][j];
                            const f32x2 lo = __builtin_elementwise_fma(rx, (f32x2){a4[0], a4[1]}, __builtin_elementwise_fma(ry, (f32x2){sv[i][0], sv[i][1]}, (f32x2){bv[i][0], bv[i][1]}));
                            const f32x2 hi = __builtin_elementwise_fma(rx, (f32x2){a4[2], a4[3]}, __builtin_elementwise_fma(ry, (f32x2){sv[i][2], sv[i][3]}, (f32x2){bv[i][2], bv[i][3]}));
                            v = (f32x4){lo[0], lo[1], hi[0], hi[1]};
#else
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                v[e] = __builtin_fmaf(rst[j].x, acc[4 * ih + i][j][e], __builtin_fmaf(rst[j].y, sv[i][e], bv[i][e]));
#endif
                        } else {
                            v = acc[4 * ih + i][j] + bv[i];
                        }
                        if constexpr (ACT == 1) v = GEMM_GELU_H ? gelu_erf4_h(v, inva) : gelu_erf4(v);      // half output: the degree-4 fit (common.h)
                        else if constexpr (ACT == 2) v = silu4(v);
                        ovf_see4<T>(ovf_mx, v);
                        v4 h;
                        h[0] = HFROM(v[0]); h[1] = HFROM(v[1]); h[2] = HFROM(v[2]); h[3] = HFROM(v[3]);
                        *(v4*)(slab + lc * HP_LD + 16 * i + 4 * lg) = h;
                    }
#endif
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int ps = 0; ps < 2; ++ps) {
                        const int ml = 8 * ps + er;
                        const v8 h = *(const v8*)(slab + ml * HP_LD + 8 * ec);
                        if (!GEMM_NOSTORE) buf_st16<GEMM_NT ? 2 : 0>(h, obuf, ovoff + (16 * j + 8 * ps) * ldh * 2 + 128 * ih);
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // slab reads done before the next chunk overwrites it
                }
            }
#endif
        } else if constexpr (fast_resid) {
            // out = half( resid * alpha + acc + bias ), the sum formed in fp32 AFTER the transpose (fp32 slab), so the residual is read and the
            // result written as row-contiguous 16-byte vectors and nothing is rounded twice.  LNR: resid = LayerNorm(lnr_y) applied on the
            // fly, alpha * LN(y) + bias = (y * rstd - mu rstd) * (alpha gamma) + (alpha beta + bias); the launcher hands over alpha gamma
            // and alpha beta + bias ready-made (GemmArgs::lnr_prefolded: lnr_gamma, lnr_beta).
            // The epilogue walks 8 chunks of 16 rows x 64 columns (c = 4 ih + j).  A chunk's residual vectors (and row statistics) are
            // requested three chunks ahead: all of a half up front cost 64 registers this kernel does not have once the fold's vectors are
            // there (it spilled 60), none ahead exposed the memory latency eight times per tile (8-10 us against 3.3 us).
            // Output, residual and statistics rows are raw buffers that end with the wave's last valid row (buf_rsrc): no row clamp, no
            // exec masks, scalar offsets.
            const T* __restrict__ resid = (const T*)(LNR ? p.lnr_y : p.resid_half);
            const int ldres = LNR ? p.ldy : (int)p.ldrh;
            const int ldh = (int)p.ldh;
            const int row0 = em0 + wn * 64, col0 = en0 + wm * 128;
            int vrows = p.M - row0;
            vrows = vrows < 0 ? 0 : (vrows < 64 ? vrows : 64);
            const __amdgpu_buffer_rsrc_t obuf = buf_rsrc((const T*)p.out_half + (int64_t)row0 * p.ldh + col0, (unsigned)vrows * (unsigned)ldh * 2u);
            const __amdgpu_buffer_rsrc_t rbuf = buf_rsrc(resid + (int64_t)row0 * ldres + col0, (unsigned)vrows * (unsigned)ldres * 2u);
            const int ovoff = (er * ldh + 8 * ec) * 2, rvoff = (er * ldres + 8 * ec) * 2;
            constexpr int AHEAD = 3;
            v8 rh[8][2];
            auto request = [&](int c) __attribute__((always_inline)) {
#pragma unroll
                for (int ps = 0; ps < 2; ++ps) rh[c][ps] = buf_ld16<v8>(rbuf, rvoff + (16 * (c & 3) + 8 * ps) * ldres * 2 + 128 * (c >> 2));
            };
            // LNR: lane L keeps the (rstd, -mu rstd) pair of row L of the wave's 64 rows (one coalesced 512-byte read, two registers);
            // the lane that stores row r of a chunk fetches the pair from lane r with ds_bpermute (no LDS memory involved)
            float2 rsl = make_float2(0.f, 0.f);
            if (LNR) {
                int m = row0 + le;
                m = m < p.M ? m : p.M - 1;
                rsl = ((const float2*)p.lnr_rows)[m];
            }
            f32x4 bb[2][2], ga[2][2];
            auto columns = [&](int ih) __attribute__((always_inline)) {
                const int nb = col0 + 64 * ih + 8 * ec;
                const float* bsrc = LNR ? p.lnr_beta : p.bias;      // LNR: alpha * beta + bias, ready-made
                bb[ih][0] = *(const f32x4*)(bsrc + nb);
                bb[ih][1] = *(const f32x4*)(bsrc + nb + 4);
                if (LNR) { ga[ih][0] = *(const f32x4*)(p.lnr_gamma + nb); ga[ih][1] = *(const f32x4*)(p.lnr_gamma + nb + 4); }
            };
            columns(0);
#pragma unroll
            for (int c = 0; c < AHEAD; ++c) request(c);
            asm volatile("" ::: "memory");
            float* slab = (float*)(smem + 2 * STAGE2 + wid * 4096);   // 16 rows x 64 floats, 16-byte chunk c of row r at slot c ^ r
            const float alpha = p.alpha;
            // partial statistics [M][N / 64][2]: this wave's rows x its two 64-column segments; only the lane with ec == 0 of a row
            // segment stores (the other lanes' offsets are out of the buffer's range)
            const int nseg_out = p.N >> 6;
            __amdgpu_buffer_rsrc_t sbuf = obuf;
            int svoff = 0;
            if constexpr (STATS) {
                sbuf = buf_rsrc(p.stats_out + ((int64_t)row0 * nseg_out + (col0 >> 6)) * 2, (unsigned)vrows * (unsigned)nseg_out * 8u);
                svoff = ec == 0 ? er * nseg_out * 8 : 0x7ff00000;
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int ih = c >> 2, j = c & 3;
                if (c + AHEAD < 8) request(c + AHEAD);
                if (c == 2) columns(1);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    *(f32x4*)((char*)slab + lc * 256 + (((4 * i + lg) ^ lc) << 4)) = acc[4 * ih + i][j];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int ps = 0; ps < 2; ++ps) {
                    const int ml = 8 * ps + er;
                    const f32x4 v0 = *(const f32x4*)((const char*)slab + ml * 256 + (((2 * ec) ^ ml) << 4));
                    const f32x4 v1 = *(const f32x4*)((const char*)slab + ml * 256 + (((2 * ec + 1) ^ ml) << 4));
                    f32x4 o0, o1;
                    if (LNR) {
                        float2 st;
                        st.x = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(4 * (16 * j + ml), __builtin_bit_cast(int, rsl.x)));
                        st.y = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(4 * (16 * j + ml), __builtin_bit_cast(int, rsl.y)));
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            o0[e] = __builtin_fmaf(__builtin_fmaf((float)rh[c][ps][e], st.x, st.y), ga[ih][0][e], bb[ih][0][e]) + v0[e];
                            o1[e] = __builtin_fmaf(__builtin_fmaf((float)rh[c][ps][4 + e], st.x, st.y), ga[ih][1][e], bb[ih][1][e]) + v1[e];
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            o0[e] = __builtin_fmaf((float)rh[c][ps][e], alpha, v0[e] + bb[ih][0][e]);
                            o1[e] = __builtin_fmaf((float)rh[c][ps][4 + e], alpha, v1[e] + bb[ih][1][e]);
                        }
                    }
                    ovf_see4<T>(ovf_mx, o0); ovf_see4<T>(ovf_mx, o1);
                    v8 h;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { h[e] = HFROM(o0[e]); h[4 + e] = HFROM(o1[e]); }
                    buf_st16<GEMM_NT ? 2 : 0>(h, obuf, ovoff + (16 * j + 8 * ps) * ldh * 2 + 128 * ih);
                    if (STATS) {
                        // partial LayerNorm statistics of the row segment (64 columns = the 8 lanes that share er), from the fp32 values
                        // (the rounding of the stored row moves the sums by ~2^-11 / sqrt(64) relative: far below LayerNorm's own error)
                        float s1, s2;
                        stats8(o0, o1, s1, s2);
                        seg8_sum2(s1, s2);
                        __builtin_amdgcn_raw_buffer_store_b64((i32x2_buf){__builtin_bit_cast(int, s1), __builtin_bit_cast(int, s2)}, sbuf,
                                                              svoff + (16 * j + 8 * ps) * nseg_out * 8 + 8 * ih, 0, 0);
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        } else {
            // EPI 0: every option of GemmArgs, in the fast forms' operation order where they overlap (so a row comes out bit-identical
            // whichever epilogue wrote it); plain loads and the compiler's own waits (this is the path of hook taps, fp32 residual streams
            // and the first / last products of a forward, not of the layer loop)
            float* slab = (float*)(smem + 2 * STAGE2 + wid * 4096);
            const float alpha = p.alpha;
            const int nseg_out = p.N >> 6;
#pragma unroll
            for (int ih = 0; ih < 2; ++ih) {
                const int nb = en0 + wm * 128 + 64 * ih + 8 * ec;   // this lane's 8 consecutive columns
                f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = {0.f, 0.f, 0.f, 0.f};
                if (p.bias) { b0 = *(const f32x4*)(p.bias + nb); b1 = *(const f32x4*)(p.bias + nb + 4); }
                f32x4 ls0 = b0, ls1 = b0, lg0 = b0, lg1 = b0, lb0 = b0, lb1 = b0;     // folded-LayerNorm column vectors (see GemmArgs)
                // pool_part: column sums of the raw tap over this wave's 64 rows, split at the one clip boundary a 64-row block can hold
                // (EPI 0 reads its third template argument as the pool mode -- 0 none, 1 column sums, 2 column maxima, 3 each clip's first row --
                //  so that the ordinary generic epilogue carries none of this: as run-time branches it cost 68 spilled registers)
                constexpr int POOL = EPI == 0 ? LN : 0;
                const float pinit = POOL == 2 ? -__builtin_inff() : 0.f;
                f32x4 pa0 = {pinit, pinit, pinit, pinit}, pa1 = pa0, pb0 = pa0, pb1 = pa0;
                const int pool_rb = (em0 + wn * 64) >> 6;
                const int pool_bnd = POOL ? (pool_rb * 64 / p.pool_T + 1) * p.pool_T : 0;      // first row of the block's second clip
                if (p.ln_rows) { ls0 = *(const f32x4*)(p.ln_s + nb); ls1 = *(const f32x4*)(p.ln_s + nb + 4); }
                if (p.lnr_y) {      // (the launcher has folded alpha into gamma and alpha * beta into the bias: GemmArgs::lnr_prefolded)
                    lg0 = *(const f32x4*)(p.lnr_gamma + nb); lg1 = *(const f32x4*)(p.lnr_gamma + nb + 4);
                    lb0 = *(const f32x4*)(p.lnr_beta + nb); lb1 = *(const f32x4*)(p.lnr_beta + nb + 4);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        *(f32x4*)((char*)slab + lc * 256 + (((4 * i + lg) ^ lc) << 4)) = acc[4 * ih + i][j];
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int ps = 0; ps < 2; ++ps) {
                        const int ml = 8 * ps + er;
                        const int m = em0 + wn * 64 + 16 * j + ml;
                        f32x4 v0 = *(const f32x4*)((const char*)slab + ml * 256 + (((2 * ec) ^ ml) << 4));
                        f32x4 v1 = *(const f32x4*)((const char*)slab + ml * 256 + (((2 * ec + 1) ^ ml) << 4));
                        if (m < p.M) {
                            const f32x4 a0 = v0, a1 = v1;      // accumulators before the bias (the folded-LayerNorm residual adds them last)
                            if (p.ln_rows) {
                                const float2 st = ((const float2*)p.ln_rows)[m];
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    v0[e] = __builtin_fmaf(st.x, v0[e], __builtin_fmaf(st.y, ls0[e], b0[e]));
                                    v1[e] = __builtin_fmaf(st.x, v1[e], __builtin_fmaf(st.y, ls1[e], b1[e]));
                                }
                            } else {
                                v0 += b0; v1 += b1;
                            }
                            if (p.row_zero != nullptr && p.row_zero[m] != 0) {
                                v0 = (f32x4){0.f, 0.f, 0.f, 0.f}; v1 = v0;
                            }
                            if (p.out_raw) {
                                st_out((f32x4*)(p.out_raw + (int64_t)m * p.ldraw + nb), v0, p.nt);
                                st_out((f32x4*)(p.out_raw + (int64_t)m * p.ldraw + nb + 4), v1, p.nt);
                            }
                            if constexpr (POOL == 3) {      // pool_part IS the [clips][N] output: the row that starts a clip is copied
                                if (m == pool_bnd || m == pool_bnd - p.pool_T) {
                                    float* dst = p.pool_part + (int64_t)(m / p.pool_T) * p.N + nb;
                                    *(f32x4*)dst = v0; *(f32x4*)(dst + 4) = v1;
                                }
                            } else if constexpr (POOL == 2) {
                                if (m < pool_bnd) { pa0 = __builtin_elementwise_max(pa0, v0); pa1 = __builtin_elementwise_max(pa1, v1); }
                                else { pb0 = __builtin_elementwise_max(pb0, v0); pb1 = __builtin_elementwise_max(pb1, v1); }
                            } else if constexpr (POOL == 1) {
                                if (m < pool_bnd) { pa0 += v0; pa1 += v1; } else { pb0 += v0; pb1 += v1; }
                            }
                            if (p.resid) {
                                const f32x4 r0 = *(const f32x4*)(p.resid + (int64_t)m * p.ldr + nb);
                                const f32x4 r1 = *(const f32x4*)(p.resid + (int64_t)m * p.ldr + nb + 4);
#pragma unroll
                                for (int e = 0; e < 4; ++e) { v0[e] = __builtin_fmaf(r0[e], alpha, v0[e]); v1[e] = __builtin_fmaf(r1[e], alpha, v1[e]); }
                            } else if (p.resid_half) {
                                const v8 rh = *(const v8*)((const T*)p.resid_half + (int64_t)m * p.ldrh + nb);
#pragma unroll
                                for (int e = 0; e < 4; ++e) { v0[e] = __builtin_fmaf((float)rh[e], alpha, v0[e]); v1[e] = __builtin_fmaf((float)rh[4 + e], alpha, v1[e]); }
                            } else if (p.lnr_y) {
                                const v8 rh = *(const v8*)((const T*)p.lnr_y + (int64_t)m * p.ldy + nb);
                                const float2 st = ((const float2*)p.lnr_rows)[m];
#pragma unroll
                                for (int e = 0; e < 4; ++e) {     // the same operations in the same order as EPI 2 with LN bit 0: bit-identical rows
                                    v0[e] = __builtin_fmaf(__builtin_fmaf((float)rh[e], st.x, st.y), lg0[e], lb0[e]) + a0[e];
                                    v1[e] = __builtin_fmaf(__builtin_fmaf((float)rh[4 + e], st.x, st.y), lg1[e], lb1[e]) + a1[e];
                                }
                            }
                            if (p.gelu) { v0 = act4_any(v0, p.gelu, inva); v1 = act4_any(v1, p.gelu, inva); }
                            if (p.out_f32) {
                                st_out((f32x4*)(p.out_f32 + (int64_t)m * p.ldo + nb), v0, p.nt);
                                st_out((f32x4*)(p.out_f32 + (int64_t)m * p.ldo + nb + 4), v1, p.nt);
                            }
                            if (p.out_half) {
                                if (p.half_scale != 0.f) { v0 = v0 * p.half_scale; v1 = v1 * p.half_scale; }      // (after the fp32 output; stats_out is refused with it)
                                ovf_see4<T>(ovf_mx, v0); ovf_see4<T>(ovf_mx, v1);
                                v8 h;
#pragma unroll
                                for (int e = 0; e < 4; ++e) { h[e] = HFROM(v0[e]); h[4 + e] = HFROM(v1[e]); }
                                st_out((v8*)((T*)p.out_half + (int64_t)m * p.ldh + nb), h, p.nt);
                            }
                            if (p.stats_out) {     // the 8 lanes of a row segment share m: all of them are here
                                float s1, s2;
                                stats8(v0, v1, s1, s2);      // the function EPI 2 uses: same bits
                                seg8_sum2(s1, s2);
                                if (ec == 0)
                                    *(float2*)(p.stats_out + ((int64_t)m * nseg_out + ((en0 + wm * 128 + 64 * ih) >> 6)) * 2) = make_float2(s1, s2);
                            }
                        }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                if constexpr (POOL == 1 || POOL == 2) {
                    // the 8 lanes with the same (lane & 7) hold the same 8 columns of different rows: combine them in a fixed order
                    // (xor 8, 16, 32), then lanes 0..7 write the block's two slots
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
#pragma unroll
                        for (int sh = 8; sh <= 32; sh <<= 1) {
                            if constexpr (POOL == 2) {
                                pa0[e] = __builtin_fmaxf(pa0[e], __shfl_xor(pa0[e], sh, 64)); pa1[e] = __builtin_fmaxf(pa1[e], __shfl_xor(pa1[e], sh, 64));
                                pb0[e] = __builtin_fmaxf(pb0[e], __shfl_xor(pb0[e], sh, 64)); pb1[e] = __builtin_fmaxf(pb1[e], __shfl_xor(pb1[e], sh, 64));
                            } else {
                                pa0[e] += __shfl_xor(pa0[e], sh, 64); pa1[e] += __shfl_xor(pa1[e], sh, 64);
                                pb0[e] += __shfl_xor(pb0[e], sh, 64); pb1[e] += __shfl_xor(pb1[e], sh, 64);
                            }
                        }
                    }
                    if (er == 0 && pool_rb * 64 < p.M) {
                        float* dst = p.pool_part + ((int64_t)pool_rb * 2) * p.N + nb;
                        *(f32x4*)dst = pa0; *(f32x4*)(dst + 4) = pa1;
                        *(f32x4*)(dst + p.N) = pb0; *(f32x4*)(dst + p.N + 4) = pb1;
                    }
                }
            }
        }
        ovf_lanes |= ovf_mask<T>(ovf_mx);
        if (GEMM_HW_SAT) AVX_F16_SAT_END();
        AVX_STAMP(if (stamp) g_gemm_stamps[4 * tile + 3] = __builtin_amdgcn_s_memrealtime(););
        if (!has_next) break;
        {   // the next tile's 8 source pointers (16 registers) are computed a second time here instead of being carried through the
            // epilogue, which needs every register it can get (the barrier on the tile id keeps the two computations apart)
            int t2 = next_tile;
            asm volatile("" : "+s"(t2));
            set_tile(t2);
        }
        if (wm == 1) { AVX_BAR(); }      // stagger again for the next tile's loop
        tile = next_tile;
    }
    ovf_commit<T>(p.ovf, ovf_lanes);
}

template <typename T, int EPI, int LN, int ACT = 0>
static int launch256(const avx::GemmArgs& a5, int grid, hipStream_t s) {
    AVX_ENSURE_LDS((gemm256p_kernel<T, EPI, LN, ACT>), LDS5);
    hipLaunchKernelGGL((gemm256p_kernel<T, EPI, LN, ACT>), dim3(grid), dim3(512), LDS5, s, a5);
    AVX_LAUNCH_CHECK();
    return AVEXHIP_OK;
}


// ---------------------------------------------------------------------------------------------
// Skinny streaming product (variant 7): a long thin activation matrix against a small weight matrix -- EfficientNet's 1 x 1 convolutions
// at the early stages (8.2 M rows, K = 64, N = 128): HBM-bound.  The tile-per-workgroup kernels re-stage a 16 KB weight block through the
// LDS for every 128 rows (1.9 TB/s measured).  Here the WHOLE W [N, K] (<= 64 KiB) is put in LDS once per workgroup and the rows of A
// stream from global memory straight into MFMA operand registers: lane l of a 16-row tile reads the 16 bytes A[row l % 16][32 ks + 8 (l / 16) ..]
// -- exactly the B-operand layout of v_mfma_f32_16x16x32 -- so there is no LDS staging of A, no barrier and no transpose in the loop; with W
// as the A operand a lane ends up with four consecutive output columns of one row (8-byte stores).  One wave = 32 rows x N per trip.
// NT = N / 16, KS = K / 32 are compile-time (register arrays).  Epilogue: bias, activation, half residual, n_store; half output only.
// ---------------------------------------------------------------------------------------------
template <typename T, int NT, int KS, bool SCALE, bool RAW>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(const avx::GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename Half<T>::v8 v8;
    typedef typename Half<T>::v4 v4;
    constexpr int N = NT * 16, K = KS * 32, KC = K / 8;      // KC 16-byte chunks per W row
    // chunk index XOR (row & SW): 16 consecutive rows at one chunk spread over the banks; SW + 1 = the largest power of two (<= 8) dividing KC,
    // so that the swizzled index stays inside the row (K = 96: KC = 12, SW = 3)
    constexpr int SW = KC % 8 == 0 ? 7 : (KC % 4 == 0 ? 3 : 1);
    constexpr int G = NT % 4 == 0 ? 4 : 2;                   // MFMA tiles whose accumulators a lane stores as one run of 4 G consecutive columns
    static_assert(NT % G == 0 && KC % 2 == 0, "skinny kernel: N must be a multiple of 32, K of 32");
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const T* W = (const T*)p.W;
    const T* A = (const T*)p.A;
    // W rows are PERMUTED on their way into the LDS so that a lane's accumulators of G = 4 MFMA tiles are 16 CONSECUTIVE output columns:
    // LDS row (tile 4 t + q, row 4 g + r) holds W row n = 64 t + 16 g + 4 q + r.  A lane (g = lane / 16) then owns columns
    // 64 t + 16 g .. + 15 of its output row -- two 16-byte stores -- and the four lanes of a row cover one whole 128-byte line.
    // (N = 96, 160: G = 2, runs of 8 columns, one 16-byte store.)
    for (int c = tid; c < N * KC; c += 256) {
        const int n = c / KC, ch = c - n * KC;
        const int t = n / (16 * G), within = n - t * (16 * G);
        const int row = ((t * G + ((within % (4 * G)) >> 2)) << 4) + ((within / (4 * G)) << 2) + (n & 3);
        *(uint4*)(smem + ((row * KC + (ch ^ (row & SW))) << 4)) = *(const uint4*)(W + (int64_t)n * p.ldw + ch * 8);
    }
    __syncthreads();
    const int lr = lane & 15, lq = lane >> 4;
    const int nst = p.n_store > 0 ? p.n_store : p.N;
    const float alpha = p.alpha;
    float ovf_mx = 0.f;
    const int64_t nblk = ((int64_t)p.M + 31) >> 5;
    const int64_t bstep = (int64_t)gridDim.x * 4;
    auto load_rows = [&](int64_t blk, v8 (&dst)[KS][2]) __attribute__((always_inline)) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            int64_t row = (blk << 5) + rt * 16 + lr;
            row = row < p.M ? row : p.M - 1;
            const T* ap = A + row * p.lda + lq * 8;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) dst[ks][rt] = *(const v8*)(ap + ks * 32);
        }
    };
    constexpr bool PF = KS <= 4;                             // K <= 128: the next trip's rows are requested before this trip's MFMAs (2 x 16 KS registers)
    v8 af[KS][2], an[KS][2];
    int64_t blk0 = (int64_t)blockIdx.x * 4 + wid;
    if (PF && blk0 < nblk) load_rows(blk0, an);
    for (int64_t blk = blk0; blk < nblk; blk += bstep) {
        const int64_t r0 = blk << 5;
        // small W (<= 16 fragments = 64 registers): the compiler keeps the fragments in registers across the trips -- weights resident, no LDS
        // read in the loop.  Larger: re-read from LDS every trip (hoisted they would be up to 256 registers and spill).
        if constexpr (NT * KS > 16) asm volatile("" ::: "memory");
        if constexpr (PF) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) { af[ks][0] = an[ks][0]; af[ks][1] = an[ks][1]; }
            load_rows(blk + bstep < nblk ? blk + bstep : blk, an);
        } else {
            load_rows(blk, af);
        }
        if constexpr (SCALE) {
            if (GEMM_HW_SAT) AVX_F16_SAT_BEGIN();      // (the conversions of the scaled rows; cleared again in front of the MFMAs below)
            // A rows scaled per (clip, input channel) on their way into the product: EfficientNet's squeeze-excitation rescale without its
            // own pass over the expanded tensor.  Same arithmetic as scale_channels_kernel (fp32 product, rounded to the operand type).
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                int64_t row = r0 + rt * 16 + lr;
                row = row < p.M ? row : p.M - 1;
                const float* sp = p.a_scale + (row / p.a_scale_rows) * p.a_scale_ld + lq * 8;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const f32x4 s0 = *(const f32x4*)(sp + ks * 32), s1 = *(const f32x4*)(sp + ks * 32 + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        af[ks][rt][e] = HFROM((float)af[ks][rt][e] * s0[e]);
                        af[ks][rt][4 + e] = HFROM((float)af[ks][rt][4 + e] * s1[e]);
                    }
                }
            }
        }
        // 128 output columns at a time (N = 256: two passes over the same A fragments): 64 accumulator registers instead of 128
        constexpr int NTP = (NT > 8 && NT % 8 == 0) ? 8 : NT;
#pragma unroll
        for (int np = 0; np < NT; np += NTP) {
            if (GEMM_HW_SAT && (SCALE || np > 0 || blk != blk0)) AVX_F16_SAT_END();      // MFMAs run with MODE.FP16_OVFL clear (common.h)
            f32x4 acc[NTP][2];
#pragma unroll
            for (int nt = 0; nt < NTP; ++nt) { acc[nt][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[nt][1] = acc[nt][0]; }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                for (int nt = 0; nt < NTP; ++nt) {
                    const int n = (np + nt) * 16 + lr;
                    const v8 wf = *(const v8*)(smem + ((n * KC + ((ks * 4 + lq) ^ (n & SW))) << 4));
                    acc[nt][0] = mfma16(wf, af[ks][0], acc[nt][0]);
                    acc[nt][1] = mfma16(wf, af[ks][1], acc[nt][1]);
                }
            }
            if (GEMM_HW_SAT) AVX_F16_SAT_BEGIN();
            AVX_CLAMP_TOKEN(inva);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const int64_t m = r0 + rt * 16 + lr;
                if (m >= p.M) continue;
#pragma unroll
                for (int t = 0; t < NTP / G; ++t) {
                    const int n = (np / G + t) * (16 * G) + lq * (4 * G);      // this lane's 4 G consecutive columns
                    if (n >= nst) continue;
                    v8 rh[2];
                    if (p.resid_half) {
                        rh[0] = *(const v8*)((const T*)p.resid_half + m * p.ldrh + n);
                        if (G == 4) rh[1] = *(const v8*)((const T*)p.resid_half + m * p.ldrh + n + 8);
                    }
                    v8 h[2];
#pragma unroll
                    for (int q = 0; q < G; ++q) {
                        f32x4 v = acc[G * t + q][rt];
                        if (p.bias) v += *(const f32x4*)(p.bias + n + 4 * q);
                        if constexpr (RAW) *(f32x4*)(p.out_raw + m * p.ldraw + n + 4 * q) = v;      // fp32 hook tap (before the residual); a template flag: as a run-time branch it cost the N = 128 instantiation its second wave per SIMD
                        if (p.resid_half) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = (float)rh[q >> 1][4 * (q & 1) + e] * alpha + v[e];
                        }
                        if (p.gelu) v = act4_any(v, p.gelu, inva);
                        ovf_see4<T>(ovf_mx, v);
#pragma unroll
                        for (int e = 0; e < 4; ++e) h[q >> 1][4 * (q & 1) + e] = HFROM(v[e]);
                    }
                    *(v8*)((T*)p.out_half + m * p.ldh + n) = h[0];
                    if (G == 4) *(v8*)((T*)p.out_half + m * p.ldh + n + 8) = h[1];
                }
            }
        }
    }
    ovf_commit<T>(p.ovf, ovf_mx);
}

template <typename T, int NT, int KS, bool SCALE, bool RAW>
static int launch_skinny(const avx::GemmArgs& a, hipStream_t s) {
    int n_cu = 256;
    { const int rc_ = avx::device_cu_count(&n_cu); if (rc_ != AVEXHIP_OK) return rc_; }
    const size_t lds = (size_t)NT * 16 * KS * 32 * 2;
    AVX_ENSURE_LDS((gemm_skinny_kernel<T, NT, KS, SCALE, RAW>), 64 * 1024);
    const int64_t nblk = ((int64_t)a.M + 127) / 128;
    int per_cu = (int)(128 * 1024 / (lds > 16384 ? lds : 16384));      // workgroups per CU the LDS (and ~100 registers per lane) allows
    per_cu = per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu);
    int64_t grid = (int64_t)n_cu * per_cu;
    grid = grid < nblk ? grid : nblk;
    hipLaunchKernelGGL((gemm_skinny_kernel<T, NT, KS, SCALE, RAW>), dim3((unsigned)grid), dim3(256), lds, s, a);
    AVX_LAUNCH_CHECK();
    return AVEXHIP_OK;
}

// does the skinny kernel take this product?  (half output only, no fp32 / raw outputs, no folded LayerNorm, no row mask)
static bool skinny_ok(const avx::GemmArgs& a) {
    if (!(a.K == 32 || a.K == 64 || a.K == 96 || a.K == 128 || a.K == 160 || a.K == 256) ||
        !(a.N == 32 || a.N == 64 || a.N == 96 || a.N == 128 || a.N == 160 || a.N == 256) || a.N * a.K > 32768) return false;
    if (!a.out_half || a.out_f32 || a.resid || a.row_zero || a.ln_rows || a.lnr_y || a.stats_out || a.pool_part) return false;
    if (a.half_scale != 0.f && a.half_scale != 1.f) return false;
    if (a.lda % 8 || a.ldw % 8 || a.ldh % 8 || (a.resid_half && a.ldrh % 8) || (a.out_raw && a.ldraw % 4) || (a.n_store > 0 && a.n_store % 16)) return false;
    return true;
}

template <typename T>
static int launch_skinny_any(const avx::GemmArgs& a, hipStream_t s) {
    // the raw fp32 tap exists for the projection widths (N = 64, 128, 256) in one form: with the A-row scale (a NULL a_scale is refused)
#define AVX_SK(NTV, KSV) if (a.N == NTV * 16 && a.K == KSV * 32) { \
        if (a.out_raw) { \
            if constexpr (NTV == 2 || NTV == 4 || NTV == 8 || NTV == 16) { if (a.a_scale) return launch_skinny<T, NTV, KSV, true, true>(a, s); } \
            avexhip_set_error("gemm: the skinny kernel writes a raw tap only for N = 32 / 64 / 128 / 256 with a_scale (N=%d)", a.N); return AVEXHIP_ERR_INVALID; \
        } \
        return a.a_scale ? launch_skinny<T, NTV, KSV, true, false>(a, s) : launch_skinny<T, NTV, KSV, false, false>(a, s); }
    AVX_SK(4, 1); AVX_SK(8, 1); AVX_SK(4, 2); AVX_SK(4, 4); AVX_SK(4, 8); AVX_SK(8, 2); AVX_SK(8, 4); AVX_SK(8, 8); AVX_SK(16, 2); AVX_SK(16, 4);
    AVX_SK(6, 2); AVX_SK(4, 3); AVX_SK(8, 3); AVX_SK(10, 2); AVX_SK(4, 5); AVX_SK(8, 5);      // EfficientNet's 96- and 144 (-> 160)-channel expansions
    AVX_SK(2, 1); AVX_SK(2, 2); AVX_SK(2, 3); AVX_SK(2, 4); AVX_SK(2, 5); AVX_SK(2, 8); AVX_SK(6, 1); AVX_SK(10, 1); AVX_SK(16, 1);      // ... and its 16- and 24-channel block outputs kept at 32 channels in memory
#undef AVX_SK
    avexhip_set_error("gemm: no skinny instantiation for N=%d K=%d", a.N, a.K);
    return AVEXHIP_ERR_INVALID;
}

template <typename T>
int launch(const avx::GemmArgs& a, hipStream_t s) {
    // variant 7 / auto for long thin products: the skinny streaming kernel (W resident in LDS, A rows straight into MFMA operands)
    AVX_REQUIRE(!a.a_scale || ((a.variant == 7 || a.variant == 1) && a.a_scale_rows > 0 && a.a_scale_ld >= a.K && a.a_scale_ld % 4 == 0),
                "gemm: a_scale is built for the skinny kernel (variant 7) and the register-staged 128-tile kernel (variant 1)");
    if (a.variant == 7) {
        AVX_REQUIRE(skinny_ok(a), "gemm: variant 7 (skinny) takes K in {32, 64, 96, 128, 160, 256}, N in {32, 64, 96, 128, 160, 256} with N K <= 32768, a half output (N=%d K=%d)", a.N, a.K);
        return launch_skinny_any<T>(a, s);
    }
    if (a.variant == 0 && a.M >= 32768 && a.K % 64 == 0 && (a.N == 64 || a.N % 128 == 0) && !a.out_raw && skinny_ok(a) &&
        (a.N % BN != 0 || (!(getenv("AVEX_AMD_GEMM_SKINNY") && atoi(getenv("AVEX_AMD_GEMM_SKINNY")) == 0) && !getenv("AVEX_AMD_GEMM_VARIANT"))))
        return launch_skinny_any<T>(a, s);
    // variant 8: the full-row residual kernel (gemm_row.hip), bit-identical to the streaming kernel.  NOT chosen automatically: measured on the
    // attention output projection (126 976 rows, K = 768) it takes 228 us against 206 for the streaming kernel + ln_rowstats, on fc2's shape 25 %
    // longer (profiles/r06b_gemm_row.txt: its 48 KiB of weights per 32-deep k-step arrive at the CU's L2 -> LDS rate, not at the MFMA rate).
    // AVEX_AMD_GEMM_ROW=1 selects it for N = 768 products from 32 768 rows (A/B inside one process), AVEX_AMD_GEMM_ROW_KMAX bounds K.
    if (a.variant == 8) {
        AVX_REQUIRE(avx::gemm_row_ok(a), "gemm: variant 8 (full-row kernel) takes N = 768, K %% 32 == 0, K >= 128, a half output with bias and a half or LayerNorm-folded residual only (N=%d K=%d)", a.N, a.K);
        return avx::gemm_row(a, __is_same(T, _Float16) ? AVEXHIP_F16 : AVEXHIP_BF16, s);
    }
    if (a.variant == 0 && a.N == 768 && a.M >= 32768 && avx::gemm_row_ok(a) && !getenv("AVEX_AMD_GEMM_VARIANT") && !getenv("AVEX_AMD_GEMM_GENERIC")) {
        const char* er = getenv("AVEX_AMD_GEMM_ROW");      // read per launch: A/B runs switch it inside one process
        const char* ek = getenv("AVEX_AMD_GEMM_ROW_KMAX");
        if (er && atoi(er) != 0 && a.K <= (ek ? atoi(ek) : 1024)) return avx::gemm_row(a, __is_same(T, _Float16) ? AVEXHIP_F16 : AVEXHIP_BF16, s);
    }
    if (a.rows_out) {
        // no kernel below finishes the row statistics: partials to stats_out, then ln_rowstats
        AVX_REQUIRE(a.stats_out, "gemm: rows_out without the full-row kernel needs stats_out as scratch");
        avx::GemmArgs b = a;
        b.rows_out = nullptr;
        const int rc = launch<T>(b, s);
        if (rc != AVEXHIP_OK) return rc;
        return avx::ln_rowstats(a.stats_out, a.M, a.N / 64, a.rows_eps, a.rows_out, s);
    }
    // variant: 0 = auto, 1 = 128-tile register staging, 3 = 128-tile LDS-DMA, 5 (or 2, its tile-per-workgroup ancestor's number) =
    // the 256-tile streaming kernel
    int variant = a.variant;
    const bool ln_fold = a.ln_rows || a.lnr_y || a.stats_out;      // (rows_out was turned into stats_out + ln_rowstats above)
    if (ln_fold) {
        // folded LayerNorm exists in the 256-tile kernel only
        AVX_REQUIRE(a.N % T2 == 0 && a.K >= 2 * BK && (!a.out_half || a.ldh % 8 == 0), "gemm: folded LayerNorm needs N %% 256 == 0 and K >= 128 (N=%d K=%d)", a.N, a.K);
        AVX_REQUIRE(!a.ln_rows || (a.ln_s && a.bias && a.M >= 2), "gemm: ln_rows needs ln_s and bias");
        AVX_REQUIRE(!a.lnr_y || (a.lnr_rows && a.lnr_gamma && a.lnr_beta && a.bias && a.lnr_prefolded && a.ldy % 8 == 0 && !a.resid && !a.resid_half),
                    "gemm: lnr_y needs lnr_rows / gamma / beta / bias and no other residual");
        AVX_REQUIRE(a.variant == 0 || a.variant == 2 || a.variant == 5, "gemm: folded LayerNorm is built for the 256-tile kernel only");
        variant = 5;
    }
    if (a.n_store > 0 && a.n_store < a.N) {      // narrow outputs: the 128-tile kernels only
        AVX_REQUIRE(a.n_store % 4 == 0 && !a.ln_rows && !a.lnr_y && !a.stats_out && !a.pool_part, "gemm: n_store=%d needs a multiple of 4 and no folded LayerNorm / pooled tap", a.n_store);
        if (variant == 0 || variant == 2 || variant == 5) variant = 3;
    }
    if (a.pool_part) {
        AVX_REQUIRE(a.pool_T >= 64 && a.pool_mode >= 0 && a.pool_mode <= 2, "gemm: pool_part needs clips of at least 64 rows (got %d) and pool_mode 0..2 (got %d)", a.pool_T, a.pool_mode);
        AVX_REQUIRE((a.variant == 0 || a.variant == 2 || a.variant == 5) && a.N % T2 == 0 && a.K >= 2 * BK, "gemm: pool_part is built for the 256-tile kernel only");
        variant = 5;
    }
    if (variant == 0) { static const char* fv = getenv("AVEX_AMD_GEMM_VARIANT"); if (fv) variant = atoi(fv); }
    if (variant == 0) {
        // the 256-tile streaming kernel wants enough tiles to occupy the chip: from about half a tile per CU it wins (K = 768 -> N = 2304 at
        // 3 968 rows, 144 tiles: 26 us against 41), below that the 128-tile kernel does -- four times the tiles, split-K for long contractions
        // (K = 3072 -> N = 768 at 3 968 rows, 48 tiles: 44 us against 66; scripts/gemm_midsize.py, profiles/r03r_midsize.txt)
        variant = avx::gemm_streams(a.M, a.N) ? 5 : 3;
    }
    if (a.post_ln_w) variant = 3;      // (the caller checked gemm_post_ln_ok)
    if (variant == 2) variant = 5;
    if (variant == 5 && (a.K < 2 * BK || a.N % T2 != 0 || (a.out_half && a.ldh % 8) || (a.resid_half && a.ldrh % 8))) variant = 3;
    if (variant == 5) {
        int n_cu = 256;
        { const int rc_ = avx::device_cu_count(&n_cu); if (rc_ != AVEXHIP_OK) return rc_; }
        const int tiles = ((a.M + T2 - 1) / T2) * (a.N / T2);
        avx::GemmArgs a5 = a;
        const char* eo = getenv("AVEX_AMD_GEMM_TILE_ORDER");      // read per launch: A/B runs switch it inside one process
        a5.tile_order = eo ? atoi(eo) : 0;
        a5.nt = gemm_nt_mode(a);
        int grid = tiles < n_cu ? ((tiles + 7) / 8) * 8 : (n_cu / 8) * 8;
        if (grid < 8) grid = 8;
        if (const char* fg = getenv("AVEX_AMD_GEMM_GRID")) { const int g = atoi(fg); if (g >= 8) grid = (g / 8) * 8; }   // tests: force many tiles per workgroup
        const char* fgen = getenv("AVEX_AMD_GEMM_GENERIC");
        const bool force_generic = fgen && atoi(fgen) != 0;     // tests: cross-check of the fast epilogues
        const bool scaled = a.half_scale != 0.f && a.half_scale != 1.f;      // the fast epilogues do not know GemmArgs::half_scale
        const bool plain_out = a.out_half && a.bias && !a.out_f32 && !a.out_raw && !a.pool_part && !a.resid && !a.row_zero && !force_generic && !scaled;
        const bool fast_half = plain_out && !a.resid_half && !a.lnr_y && !a.stats_out && (a.gelu <= 2 || a.gelu == 6);      // the fast epilogue knows GELU and SiLU only
        const bool fast_resid = plain_out && (a.resid_half || a.lnr_y) && !a.gelu && !a.ln_rows;
        if (fast_half) {
            if (a.gelu == 1 || a.gelu == 6) return a.ln_rows ? launch256<T, 1, 1, 1>(a5, grid, s) : launch256<T, 1, 0, 1>(a5, grid, s);      // (half output only: both mean the degree-4 fit here)
            if (a.gelu == 2) return a.ln_rows ? launch256<T, 1, 1, 2>(a5, grid, s) : launch256<T, 1, 0, 2>(a5, grid, s);
            return a.ln_rows ? launch256<T, 1, 1, 0>(a5, grid, s) : launch256<T, 1, 0, 0>(a5, grid, s);
        }
        if (fast_resid) {
            if (a.lnr_y) return a.stats_out ? launch256<T, 2, 3>(a5, grid, s) : launch256<T, 2, 1>(a5, grid, s);
            return a.stats_out ? launch256<T, 2, 2>(a5, grid, s) : launch256<T, 2, 0>(a5, grid, s);
        }
        if (a.pool_part) {
            if (a.pool_mode == 0) return launch256<T, 0, 1>(a5, grid, s);
            if (a.pool_mode == 1) return launch256<T, 0, 2>(a5, grid, s);
            return launch256<T, 0, 3>(a5, grid, s);
        }
        return launch256<T, 0, 0>(a5, grid, s);
    }
    const int tiles = ((a.M + BM - 1) / BM) * (a.N / BN);
    static const char* pad_env = getenv("AVEX_AMD_DEBUG_LDS_PAD");
    const size_t lds = 2 * 2 * TILE_BYTES + (pad_env ? atoi(pad_env) : 0);
    AVX_ENSURE_LDS((gemm_nt_kernel<T, false>), 96 * 1024);
    AVX_ENSURE_LDS((gemm_nt_kernel<T, true>), 96 * 1024);
    if (variant == 1) {
        hipLaunchKernelGGL((gemm_nt_kernel<T, false>), dim3(tiles), dim3(256), lds, s, a);
    } else {
        // split-K when the caller lent a workspace and the product is few tiles of a long contraction (one clip's fc2: 24 tiles, K = 3072)
        int S = 1;
        if (a.post_ln_w) AVX_REQUIRE(avx::gemm_post_ln_ok(a), "gemm: post_ln_* needs the 128-tile kernel's workspace path (N %% 256 == 0, N <= 1024, no activation, splitk_ws >= M N floats)");
        if (a.splitk_ws && (a.K >= 1024 || a.post_ln_w)) {
            int n_cu = 256;
            { const int rc_ = avx::device_cu_count(&n_cu); if (rc_ != AVEXHIP_OK) return rc_; }
            S = 8;      // as many splits as keep the launch within two workgroups per CU (and leave every split at least two K-steps)
            while (S > 1 && (tiles * S > 2 * n_cu || a.K % (S * BK) != 0 || a.K / S < 2 * BK || (size_t)S * a.M * a.N * sizeof(float) > a.splitk_bytes)) S >>= 1;
        }
        hipLaunchKernelGGL((gemm_nt_kernel<T, true>), dim3(tiles, S), dim3(256), lds, s, a);
        if (a.post_ln_w) {
            AVX_LAUNCH_CHECK();
            hipLaunchKernelGGL((splitk_ln_epilogue_kernel<T>), dim3((unsigned)((a.M + 3) / 4)), dim3(256), 0, s, a, S);
        } else if (S > 1) {
            AVX_LAUNCH_CHECK();
            const int64_t nthr = (int64_t)a.M * (a.N / 4);
            hipLaunchKernelGGL((splitk_epilogue_kernel<T>), dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, s, a, S);
        }
    }
    AVX_LAUNCH_CHECK();
    return AVEXHIP_OK;
}

}  // namespace

namespace avx {

bool gemm_streams(int M, int N) {
    static const int min_tiles = getenv("AVEX_AMD_GEMM_256_MIN_TILES") ? atoi(getenv("AVEX_AMD_GEMM_256_MIN_TILES")) : 128;
    const int t256 = ((M + T2 - 1) / T2) * (N / T2);
    return N % T2 == 0 && M >= 1024 && t256 >= min_tiles;
}

bool gemm_post_ln_ok(const GemmArgs& a) {
    const char* e = getenv("AVEX_AMD_POST_LN");      // read per call (24 per forward): tests switch it within a process
    const bool off = e && atoi(e) == 0;
    return !off && a.splitk_ws && a.N % 256 == 0 && a.N <= 1024 && a.K % BK == 0 && !a.gelu && !a.ln_rows && !a.lnr_y && !a.stats_out && !a.pool_part &&
           !(a.n_store > 0 && a.n_store < a.N) && (size_t)a.M * a.N * sizeof(float) <= a.splitk_bytes && !gemm_streams(a.M, a.N) &&
           (a.variant == 0 || a.variant == 3);
}

int gemm(const GemmArgs& a, int dtype, hipStream_t s) {
    AVX_REQUIRE(a.A && a.W, "gemm: A and W must be non-null");
    AVX_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "gemm: empty problem M=%d N=%d K=%d", a.M, a.N, a.K);
    AVX_REQUIRE(a.N % BN == 0 || ((a.variant == 7 || (a.variant == 0 && a.N == 64 && a.M >= 32768 && a.K % 64 == 0)) && skinny_ok(a)),
                "gemm: N=%d must be a multiple of %d (64 columns: the skinny streaming kernel only, >= 32768 rows)", a.N, BN);
    AVX_REQUIRE(a.K % BK == 0 || (a.K % 32 == 0 && a.variant == 7), "gemm: K=%d must be a multiple of %d (of 32 with the skinny kernel, variant 7)", a.K, BK);
    AVX_REQUIRE(a.lda % 8 == 0 && a.ldw % 8 == 0, "gemm: lda/ldw must be multiples of 8 elements");
    AVX_REQUIRE(a.half_scale == 0.f || a.half_scale == 1.f || (a.out_half && !a.stats_out && !a.post_ln_w && !a.pool_part && a.variant != 7 && a.half_scale > 0.f),
                "gemm: half_scale goes with a plain half output (no row statistics, folded post-LayerNorm, pooled tap or skinny kernel)");
    AVX_REQUIRE(a.out_f32 || a.out_half || a.out_raw || (a.post_ln_w && (a.post_ln_out_f32 || a.post_ln_out_half)), "gemm: no output buffer");
    AVX_REQUIRE((!a.out_f32 || a.ldo % 4 == 0) && (!a.out_half || a.ldh % 4 == 0) &&
                    (!a.out_raw || a.ldraw % 4 == 0) && (!a.resid || a.ldr % 4 == 0) &&
                    (!a.resid_half || a.ldrh % 4 == 0),
                "gemm: output/residual leading dims must be multiples of 4 elements");
    if (dtype != AVEXHIP_F16 && dtype != AVEXHIP_BF16) {
        avexhip_set_error("gemm: unknown dtype %d", dtype);
        return AVEXHIP_ERR_INVALID;
    }
    if (a.gelu == 1 && GEMM_GELU_H && a.out_half && !a.out_f32) {
        // GELU whose only consumer reads the operand type: the degree-4 fit, in whichever kernel and epilogue form runs (activation code 6)
        GemmArgs b = a;
        b.gelu = 6;
        return gemm(b, dtype, s);
    }
    if (a.lnr_y && !a.lnr_prefolded) {
        // the kernel takes alpha * gamma and bias + alpha * beta: callers that launch the same fold repeatedly keep those vectors
        // (lnr_prefolded); for the others they are made here, in stream-ordered scratch
        AVX_REQUIRE(a.lnr_gamma && a.lnr_beta && a.bias, "gemm: lnr_y needs lnr_gamma, lnr_beta and bias");
        float* tmp = nullptr;
        AVX_HIP_CHECK(hipMallocAsync((void**)&tmp, sizeof(float) * 2 * (size_t)a.N, s));
        GemmArgs b = a;
        int rc = lnr_fold(a.lnr_gamma, a.lnr_beta, a.bias, a.alpha, a.N, tmp, tmp + a.N, s);
        if (rc == AVEXHIP_OK) {
            b.lnr_gamma = tmp; b.lnr_beta = tmp + a.N; b.lnr_prefolded = 1;
            rc = dtype == AVEXHIP_F16 ? launch<_Float16>(b, s) : launch<__bf16>(b, s);
        }
        (void)hipFreeAsync(tmp, s);
        return rc;
    }
    return dtype == AVEXHIP_F16 ? launch<_Float16>(a, s) : launch<__bf16>(a, s);
}

}  // namespace avx

#ifdef AVEX_DIAG
// debug: enable/read the per-block stamps of gemm256_kernel (start, prologue done, loop done, epilogue stores retired)
extern "C" int avexhip_debug_gemm_stamps(int enable, unsigned long long* host_out, int n_blocks) {
    int on = enable;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_stamps_on), &on, sizeof(int)) != hipSuccess) return -2;
    if (host_out && n_blocks > 0) {
        if (n_blocks > 8192) n_blocks = 8192;
        if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_gemm_stamps), sizeof(unsigned long long) * 4 * n_blocks) != hipSuccess) return -2;
    }
    return 0;
}

// debug: shader-clock readings (s_memtime) at the K loop's start and end, per tile, of the persistent 256-tile GEMM
extern "C" int avexhip_debug_gemm_clocks(unsigned long long* host_out, int n_tiles) {
    if (!host_out || n_tiles <= 0) return -1;
    if (n_tiles > 8192) n_tiles = 8192;
    if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_gemm_clk), sizeof(unsigned long long) * 2 * n_tiles) != hipSuccess) return -2;
    return 0;
}

// debug: shader-clock readings at the top of each K-tile (and after the last) of every workgroup's third tile, variant 5: [256][64]
extern "C" int avexhip_debug_gemm_kclocks(unsigned long long* host_out, int n_blocks) {
    if (!host_out || n_blocks <= 0) return -1;
    if (n_blocks > 256) n_blocks = 256;
    if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_gemm_kclk), sizeof(unsigned long long) * 64 * n_blocks) != hipSuccess) return -2;
    return 0;
}
#endif  // AVEX_DIAG
