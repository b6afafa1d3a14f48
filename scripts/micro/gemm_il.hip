// The K = 768 GEMM with its epilogue BETWEEN the MFMAs of the next pass's K loop (round 5; DESIGN.md section 7 item 1, the review's item 2).
// The successor of gemm_dbuf.hip, written with what attention16.hip taught about keeping hipcc to a hand-made schedule from HIP source:
//   * every K-tile position is straight-line code (twelve K-tiles of a pass unrolled, the pass a template over its accumulator set): no
//     switch, no phi copies, accumulation in place;
//   * the epilogue of the pass before is a list of two-instruction STEPS handed out between the MFMAs by a static schedule; every step's
//     inputs go through an empty asm so that instruction selection cannot float it past the scheduling fences;
//   * LDS-DMA as inline assembly on a scalar base + lane offset (no address arithmetic for the compiler to hoist and spill);
//   * the transpose for 16-byte stores is two lane swaps (v_permlane32_swap, v_permlane16_swap) in registers: the two-pass loop is bound
//     by LDS bandwidth (64 x 64 outputs per wave: 512 B of fragments per MFMA), so the slab of gemm_dbuf.hip is gone;
//   * stores leave from the MFMA segment; the load segment's counted wait counts past them.
// UNLIKE gemm_dbuf.hip THIS ONE IS CHECKED: out = f16(gelu(A W^T + bias)) against a host reference (first argument `check`).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off scripts/micro/gemm_il.hip -o scripts/micro/bin/gemm_il
//   scripts/micro/bin/gemm_il check        (M = 2048: every output against the host)
//   scripts/micro/bin/gemm_il [iters]      (M = 126976, N = 3072, K = 768: fc1 of the 256-clip step)
#include "../../avex_amd/csrc/common.h"
#include <math.h>
#include <stdlib.h>
#include <type_traits>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

#ifndef IL_KO
#define IL_KO 0          // knock-outs: 1 no stores, 2 no epilogue arithmetic, 4 no lane swaps, 8 no MODE toggles, 16 no fences
#endif
#ifndef IL_ST_AUX
#define IL_ST_AUX 2      // cache policy of the output stores: 2 = nt (the product kernel's), 0 = default
#endif

constexpr int BK = 64, NK = 12, K = NK * BK;
constexpr int STAGE = 49152;                   // W half-tile 128 rows (16 KiB) + X tile 256 rows (32 KiB)
constexpr int BIAS_OFF = 3 * STAGE;            // the whole bias vector (N <= 3072 floats) behind the three stages
constexpr int LDS_BYTES = BIAS_OFF + 12288;    // 159 744

#define VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define FENCE() do { if (!(IL_KO & 16)) __builtin_amdgcn_sched_barrier(0); } while (0)
#define BAR() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)

template <int I> using IC = std::integral_constant<int, I>;
template <int B, int E, typename F> static __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) { f(IC<B>{}); static_for<B + 1, E>(f); }
}
static __device__ __forceinline__ void il_dma16(const char* sbase, unsigned voff, unsigned lds) {
    asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds) : "memory");
}
// max(x, 0) without the canonicalising v_max x, x the compiler puts in front of fmaxf on a value that came out of an asm statement
static __device__ __forceinline__ float relu_nc(float x) {
    float r;
    asm("v_max_f32_e32 %0, 0, %1" : "=v"(r) : "v"(x));
    return r;
}
static __device__ __forceinline__ void pin(f32x2& x) { asm volatile("" : "+v"(x)); }
static __device__ __forceinline__ void pin(f32x4& x) { asm volatile("" : "+v"(x)); }
static __device__ __forceinline__ void pin(unsigned& x) { asm volatile("" : "+v"(x)); }

// steps of the epilogue: 14 per group of four values (one 16 x 16 MFMA tile), 5 behind the four groups of a 16-token chunk
constexpr int NS = 14, NT = 5, CH_STEPS = 4 * NS + NT, PASS_STEPS = 4 * CH_STEPS;      // 61 per chunk, 244 per pass
constexpr int NSLOT = NK * 32;                                                          // 384 MFMAs per pass
constexpr int step_lo(int slot) { return (int)(((long long)slot * PASS_STEPS) / NSLOT); }
// the stores of chunk c are step c * 61 + 60: the MFMA slot that carries it, and from it the K-tile whose MFMA segment issues them
#ifndef IL_DEFER
#define IL_DEFER 0       // 1: the stores of chunks 0 - 2 leave at the START of the next K-tile's MFMA segment (slot 96 (c + 1) + 2) instead of at the end of
#endif                   //    their own (the counted waits are in order: a store must be acknowledged by the second wait behind it).  MEASURED: 617 against 612 us, no gain
constexpr int store_slot(int c) {
    if (IL_DEFER && c < 3) return 96 * (c + 1) + 2;
    const int s = c * CH_STEPS + CH_STEPS - 1;
    int q = 0;
    while (!(step_lo(q) <= s && s < step_lo(q + 1))) ++q;
    return q;
}
constexpr bool is_deferred_store(int s) { return IL_DEFER && s % CH_STEPS == CH_STEPS - 1 && s / CH_STEPS < 3; }
constexpr bool stores_in_ktile(int kt) {
    for (int c = 0; c < 4; ++c)
        if (store_slot(c) / 32 == kt) return true;
    return false;
}

struct PassCoord { int m0, n0; };

__global__ __launch_bounds__(512) void il_kernel(const _Float16* __restrict__ A, const _Float16* __restrict__ W, _Float16* __restrict__ out,
                                                 const float* __restrict__ bias, int M, int N, unsigned long long* __restrict__ clk, int ring_rows) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef f16x8 v8;
    const int tid = threadIdx.x;
    int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3;
    const int tiles_n = N / 256, tiles_m = M / 256, ntiles = tiles_m * tiles_n;
    float* ldsbias = (float*)(smem + BIAS_OFF);
    for (int i = tid; i < N; i += 512) ldsbias[i] = bias[i];      // (a product kernel gets the tile's bias row by LDS-DMA a tile ahead)

    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    int nt_mine = 0;
    for (int it = 0; (it * 8 + xcd) * per_xcd + slot < ntiles; ++it) ++nt_mine;
    const int npass = 2 * nt_mine;
    if (npass == 0) return;
    auto coords = [&](int p) __attribute__((always_inline)) -> PassCoord {
        const int t = ((p >> 1) * 8 + xcd) * per_xcd + slot;
        const int per_group = 8 * tiles_n;
        const int gid = t / per_group;
        const int first_m = gid * 8;
        const int gsz = (tiles_m - first_m) < 8 ? (tiles_m - first_m) : 8;
        const int r = t - gid * per_group;
        const int tn = r / gsz, tm = first_m + (r - tn * gsz);
        return {tm * 256, tn * 256 + 128 * (p & 1)};
    };
    // lane parts of the addresses (the lane index goes through an empty asm first: its arithmetic stays here instead of being re-derived,
    // hoisted and spilled all over the unrolled passes)
    asm volatile("" : "+v"(lane));
    const int lc = lane & 15, lg = lane >> 4;
    unsigned woff[2], xoff[4];       // this wave's 2 W pieces and 4 X pieces (8 rows x 128 B each) of every K-tile
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int r = 8 * (wid + 8 * q) + (lane >> 3);
        woff[q] = (unsigned)(r * K + (((lane & 7) ^ ((r >> 1) & 7)) << 3)) * 2u;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = 8 * (wid + 8 * q) + (lane >> 3);
        xoff[q] = (unsigned)(r * K + (((lane & 7) ^ ((r >> 1) & 7)) << 3)) * 2u;
    }
    const int sw = (lane >> 1) & 7;
    const int foff0 = lc * 128 + ((lg ^ sw) << 4), foff1 = foff0 ^ 64;
    const int wfrag = (64 * wm) * 128, xfrag = 16384 + (64 * wn) * 128;
    const unsigned st_voff = (unsigned)(lc * N * 2 + 16 * lg);               // store: token row lc of the chunk, bytes [16 lg, 16 lg + 16) of each 64-byte half
    const float* bias_lane = ldsbias + 64 * wm + 4 * lg;                      // + n0 + 16 i

    auto dma = [&](const char* wb, const char* xb, int kt, int stg) __attribute__((always_inline)) {      // 6 x 1 KiB
        const unsigned base = (unsigned)(stg * STAGE);
        const char* wk = wb + kt * (BK * 2);
        const char* xk = xb + kt * (BK * 2);
#pragma unroll
        for (int q = 0; q < 2; ++q) il_dma16(wk, woff[q], base + (wid + 8 * q) * 1024);
#pragma unroll
        for (int q = 0; q < 4; ++q) il_dma16(xk, xoff[q], base + 16384 + (wid + 8 * q) * 1024);
    };

    f32x4 acc[2][4][4];
    v8 wf[4][2], xf[4][2];
    // epilogue state: one group's temporaries, one chunk's converted halves
    f32x4 bq;
    f32x2 x01, x23, t01, t23, p01, p23, q01, q23;
    unsigned hA[4], hB[4];

    const char *wb_cur, *xb_cur, *wb_nxt = nullptr, *xb_nxt = nullptr;
    {
        const PassCoord c0 = coords(0);
        wb_cur = (const char*)(W + (int64_t)c0.n0 * K);
        xb_cur = (const char*)(A + (int64_t)c0.m0 * K);
    }
    __builtin_amdgcn_s_setreg(AVX_MODE_DX10_CLAMP_HWREG, 0);      // the GELU's clamp modifier passes a NaN
    __syncthreads();                    // the bias vector is in LDS
    dma(wb_cur, xb_cur, 0, 0);
    dma(wb_cur, xb_cur, 1, 1);
    VMCNT(6);
    BAR();
    if (wm == 1) BAR();                 // stagger: waves 4-7 one barrier behind
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();

    // ---- one epilogue step of accumulator set OTHER (the pass that ended at output (em0, en0)) ----
    auto estep = [&](auto OTHER, auto S, int en0, const __amdgpu_buffer_rsrc_t rs) __attribute__((always_inline)) {
        constexpr int other = decltype(OTHER)::value, s = decltype(S)::value;
        constexpr int j = s / CH_STEPS, r = s % CH_STEPS;          // chunk (16 tokens), step in the chunk
        if constexpr (r < 4 * NS) {
            constexpr int i = r / NS, u = r % NS;                   // group (16 features), step in the group
            if (IL_KO & 2) {
                if constexpr (u == NS - 1) {
                    f32x4 a = acc[other][i][j];
                    pin(a);
                    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                    hA[i] = __builtin_bit_cast(unsigned, (h2){(_Float16)a[0], (_Float16)a[1]});
                    hB[i] = __builtin_bit_cast(unsigned, (h2){(_Float16)a[2], (_Float16)a[3]});
                }
                return;
            }
            if constexpr (u == 0) {
                bq = *(const f32x4*)(bias_lane + en0 + 16 * i);
            } else if constexpr (u == 1) {
                f32x4 a = acc[other][i][j];
                pin(a);
                acc[other][i][j] = a;
                x01 = (f32x2){a[0], a[1]} + (f32x2){bq[0], bq[1]};
                x23 = (f32x2){a[2], a[3]} + (f32x2){bq[2], bq[3]};
            } else if constexpr (u == 2) {
                pin(x01);
                t01 = (f32x2){gelu_clamp_t(x01[0], AVX_GELUH_INVA), gelu_clamp_t(x01[1], AVX_GELUH_INVA)};
            } else if constexpr (u == 3) {
                pin(x23);
                t23 = (f32x2){gelu_clamp_t(x23[0], AVX_GELUH_INVA), gelu_clamp_t(x23[1], AVX_GELUH_INVA)};
            } else if constexpr (u == 4) {
                pin(t01); pin(t23);
                p01 = __builtin_elementwise_fma((f32x2)(AVX_GELUH_T4), t01, (f32x2)(AVX_GELUH_T3));
                p23 = __builtin_elementwise_fma((f32x2)(AVX_GELUH_T4), t23, (f32x2)(AVX_GELUH_T3));
            } else if constexpr (u == 5) {
                pin(p01); pin(p23);
                p01 = __builtin_elementwise_fma(p01, t01, (f32x2)(AVX_GELUH_T2));
                p23 = __builtin_elementwise_fma(p23, t23, (f32x2)(AVX_GELUH_T2));
            } else if constexpr (u == 6) {
                pin(p01); pin(p23);
                p01 = __builtin_elementwise_fma(p01, t01, (f32x2)(AVX_GELUH_T1));
                p23 = __builtin_elementwise_fma(p23, t23, (f32x2)(AVX_GELUH_T1));
            } else if constexpr (u == 7) {
                pin(p01); pin(p23);
                p01 = __builtin_elementwise_fma(p01, t01, (f32x2)(AVX_GELUH_T0));
                p23 = __builtin_elementwise_fma(p23, t23, (f32x2)(AVX_GELUH_T0));
            } else if constexpr (u == 8) {
                pin(p01);
                q01 = (f32x2){__builtin_amdgcn_exp2f(p01[0]), __builtin_amdgcn_exp2f(p01[1])};
            } else if constexpr (u == 9) {
                pin(p23);
                q23 = (f32x2){__builtin_amdgcn_exp2f(p23[0]), __builtin_amdgcn_exp2f(p23[1])};
            } else if constexpr (u == 10) {
                pin(x01);
                x01 = (f32x2){relu_nc(x01[0]), relu_nc(x01[1])};
            } else if constexpr (u == 11) {
                pin(x23);
                x23 = (f32x2){relu_nc(x23[0]), relu_nc(x23[1])};
            } else if constexpr (u == 12) {
                pin(q01); pin(q23);
                x01 = __builtin_elementwise_fma(-t01, q01, x01);
                x23 = __builtin_elementwise_fma(-t23, q23, x23);
            } else {
                pin(x01); pin(x23);
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                if (!(IL_KO & 8)) __builtin_amdgcn_s_setreg(AVX_MODE_FP16_OVFL_HWREG, 1);
                unsigned a = __builtin_bit_cast(unsigned, (h2){(_Float16)x01[0], (_Float16)x01[1]});
                unsigned b = __builtin_bit_cast(unsigned, (h2){(_Float16)x23[0], (_Float16)x23[1]});
                pin(a); pin(b);
                if (!(IL_KO & 8)) __builtin_amdgcn_s_setreg(AVX_MODE_FP16_OVFL_HWREG, 0);
                hA[i] = a; hB[i] = b;
            }
        } else {
            constexpr int v = r - 4 * NS;
            if constexpr (v < 2) {                       // lane bit 5 <-> group bit 0: registers of groups 2 v, 2 v + 1
                if (!(IL_KO & 4)) {
                    pin(hA[2 * v]); pin(hB[2 * v]);
                    const auto ra = __builtin_amdgcn_permlane32_swap(hA[2 * v], hA[2 * v + 1], false, false);
                    const auto rb = __builtin_amdgcn_permlane32_swap(hB[2 * v], hB[2 * v + 1], false, false);
                    hA[2 * v] = ra[0]; hA[2 * v + 1] = ra[1];
                    hB[2 * v] = rb[0]; hB[2 * v + 1] = rb[1];
                }
            } else if constexpr (v < 4) {                // lane bit 4 <-> the register bit that now holds the source lane's bit 5
                constexpr int h = v - 2;
                if (!(IL_KO & 4)) {
                    pin(hA[2 * h]); pin(hB[2 * h]);
                    const auto ra = __builtin_amdgcn_permlane16_swap(hA[2 * h], hA[2 * h + 1], false, false);
                    const auto rb = __builtin_amdgcn_permlane16_swap(hB[2 * h], hB[2 * h + 1], false, false);
                    hA[2 * h] = ra[0]; hA[2 * h + 1] = ra[1];
                    hB[2 * h] = rb[0]; hB[2 * h + 1] = rb[1];
                }
            } else {
                if (!(IL_KO & 1)) {
                    typedef int i32x4_st __attribute__((ext_vector_type(4)));
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        pin(hA[2 * h]);
                        const i32x4_st d = {(int)hA[2 * h], (int)hB[2 * h], (int)hA[2 * h + 1], (int)hB[2 * h + 1]};
                        __builtin_amdgcn_raw_buffer_store_b128(d, rs, st_voff + 64 * h, j * (16 * N * 2), IL_ST_AUX);
                    }
                }
            }
        }
    };
    auto out_rsrc = [&](int em0, int en0) __attribute__((always_inline)) -> __amdgpu_buffer_rsrc_t {
        const uint64_t a = (uint64_t)(out + (int64_t)(em0 % ring_rows + 64 * wn) * N + en0 + 64 * wm);      // ring_rows = M: the plain output; fewer: the same rows over and over (does the Infinity Cache absorb the stores?)
        return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)a)),
                                                 0, 64 * N * 2, 0x00020000);
    };

    // ---- one pass: 12 K-tiles into accumulator set SET, the other set's epilogue between the MFMAs when DRAIN ----
    auto pass = [&](auto SET, auto DRAIN, int pp) __attribute__((always_inline)) {
        constexpr int set = decltype(SET)::value, other = set ^ 1;
        constexpr bool drain = decltype(DRAIN)::value != 0;
        const bool has_next = pp + 1 < npass;
        PassCoord ec = {0, 0};
        if (drain) ec = coords(pp - 1);
        const __amdgpu_buffer_rsrc_t rs = out_rsrc(ec.m0, ec.n0);
        const int en0 = ec.n0;
        static_for<0, NK>([&](auto KT) __attribute__((always_inline)) {
            constexpr int kt = decltype(KT)::value;
            constexpr int st = kt % 3, st2 = (kt + 2) % 3;
            // ---- load segment
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const char* r = smem + st * STAGE + xfrag + (16 * j) * 128;
                xf[j][0] = *(const v8*)(r + foff0); xf[j][1] = *(const v8*)(r + foff1);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const char* r = smem + st * STAGE + wfrag + (16 * i) * 128;
                wf[i][0] = *(const v8*)(r + foff0); wf[i][1] = *(const v8*)(r + foff1);
            }
            if constexpr (kt == NK - 3) {
                if (has_next) {
                    const PassCoord c = coords(pp + 1);
                    wb_nxt = (const char*)(W + (int64_t)c.n0 * K);
                    xb_nxt = (const char*)(A + (int64_t)c.m0 * K);
                }
            }
            if constexpr (kt + 2 < NK) dma(wb_cur, xb_cur, kt + 2, st2);
            else {
                if (has_next) dma(wb_nxt, xb_nxt, kt + 2 - NK, st2);
                else VMCNT(0);            // the last pass requests nothing here: the counted wait below would count the requests it has to wait for
            }
            // K-tile kt + 1 (requested one K-tile ago) must have landed; what may stay in flight: this K-tile's 6 DMAs and the stores that the
            // MFMA segment between the two requests issued (younger than the request that is waited for)
            {
                constexpr bool sprev = drain && !(IL_KO & 1) && stores_in_ktile((kt + NK - 1) % NK);
                if constexpr (sprev && kt != 0) VMCNT(8);
                else if constexpr (sprev) {      // K-tile 0: the stores are those of the pass before, which drained if it was not the first
                    if (pp >= 2) VMCNT(8);
                    else VMCNT(6);
                } else VMCNT(6);
            }
            LGKM0();
            BAR();
            __builtin_amdgcn_s_setprio(1);
            // ---- MFMA segment: 32 MFMAs, the epilogue steps of slots 32 kt .. 32 kt + 31 between them
            static_for<0, 32>([&](auto Q) __attribute__((always_inline)) {
                constexpr int q = decltype(Q)::value;
                constexpr int ks = q >> 4, i = (q >> 2) & 3, jj = q & 3, j = (i & 1) ? 3 - jj : jj;
                if constexpr (kt == 0 && ks == 0) acc[set][i][j] = mfma16(wf[i][ks], xf[j][ks], (f32x4){0.f, 0.f, 0.f, 0.f});
                else acc[set][i][j] = mfma16(wf[i][ks], xf[j][ks], acc[set][i][j]);
                if constexpr (drain) {
                    constexpr int slot_id = 32 * kt + q;
                    FENCE();
                    static_for<step_lo(slot_id), step_lo(slot_id + 1)>([&](auto S) __attribute__((always_inline)) {
                        if constexpr (!is_deferred_store(decltype(S)::value)) estep(IC<other>{}, S, en0, rs);
                    });
                    if constexpr (IL_DEFER && slot_id % 96 == 2 && slot_id >= 96) estep(IC<other>{}, IC<(slot_id / 96) * CH_STEPS - 1>{}, en0, rs);
                    FENCE();
                }
            });
            __builtin_amdgcn_s_setprio(0);
            BAR();
        });
        if (has_next) { wb_cur = wb_nxt; xb_cur = xb_nxt; }
    };

    pass(IC<0>{}, IC<0>{}, 0);
    for (int pp = 1;; pp += 2) {
        pass(IC<1>{}, IC<1>{}, pp);
        if (pp + 1 >= npass) break;
        pass(IC<0>{}, IC<1>{}, pp + 1);
    }
    {   // the last pass's accumulators (set 1: npass is even), matrix pipe idle
        const PassCoord ec = coords(npass - 1);
        const __amdgpu_buffer_rsrc_t rs = out_rsrc(ec.m0, ec.n0);
        static_for<0, PASS_STEPS>([&](auto S) __attribute__((always_inline)) { estep(IC<1>{}, S, ec.n0, rs); });
    }
    if (tid == 0 && clk != nullptr) {
        clk[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
        clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

static double run(const _Float16* A, const _Float16* W, _Float16* out, const float* bias, int M, int N, unsigned long long* clk, int iters, double* ghz, int ring) {
    CK(hipFuncSetAttribute((const void*)il_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(il_kernel, dim3(256), dim3(512), LDS_BYTES, 0, A, W, out, bias, M, N, clk, ring);      // clocks settle under load
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(il_kernel, dim3(256), dim3(512), LDS_BYTES, 0, A, W, out, bias, M, N, clk, ring);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(512);
    CK(hipMemcpy(h.data(), clk, sizeof(unsigned long long) * 512, hipMemcpyDeviceToHost));
    double s = 0;
    for (int b = 0; b < 256; ++b) s += (double)h[2 * b] / ((double)h[2 * b + 1] * 10.0);      // cycles per ns (s_memrealtime: 100 MHz)
    *ghz = s / 256;
    return ms * 1e3 / iters;
}

int main(int argc, char** argv) {
    const bool check = argc > 1 && !strcmp(argv[1], "check");
    const int M = check ? 8192 : 126976, N = 3072;      // check: 384 tiles on 256 workgroups, one or two tiles each
    const int iters = (!check && argc > 1) ? atoi(argv[1]) : 200;
    const int ring = (!check && argc > 2) ? atoi(argv[2]) : M;      // second argument: output rows wrap at this many (a multiple of 256)
    _Float16 *A, *W, *out;
    float* bias;
    unsigned long long* clk;
    CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&W, (size_t)N * K * 2)); CK(hipMalloc(&out, (size_t)M * N * 2));
    CK(hipMalloc(&bias, N * 4)); CK(hipMalloc(&clk, 512 * 8));
    std::vector<_Float16> hAm((size_t)M * K), hW((size_t)N * K);
    std::vector<float> hb(N);
    {   // pseudo-random halves (the power the MFMAs draw depends on the data)
        unsigned s = 12345u;
        for (size_t i = 0; i < hAm.size(); ++i) { s = s * 1664525u + 1013904223u; hAm[i] = (_Float16)(((int)(s >> 9) % 2001 - 1000) * 1e-3f); }
        for (size_t i = 0; i < hW.size(); ++i) { s = s * 1664525u + 1013904223u; hW[i] = (_Float16)(((int)(s >> 9) % 2001 - 1000) * 1e-4f); }
        for (int i = 0; i < N; ++i) hb[i] = 0.05f * (float)(i % 17 - 8);
        CK(hipMemcpy(A, hAm.data(), hAm.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(W, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(bias, hb.data(), N * 4, hipMemcpyHostToDevice));
    }
    if (check) {
        CK(hipMemset(out, 0xff, (size_t)M * N * 2));
        CK(hipFuncSetAttribute((const void*)il_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        hipLaunchKernelGGL(il_kernel, dim3(256), dim3(512), LDS_BYTES, 0, A, W, out, bias, M, N, clk, M);
        CK(hipDeviceSynchronize());
        std::vector<_Float16> ho((size_t)M * N);
        CK(hipMemcpy(ho.data(), out, ho.size() * 2, hipMemcpyDeviceToHost));
        double worst = 0;
        size_t bad = 0, seen = 0;
        std::vector<int> hist_tile(32 * 24), hist_j(4), hist_i(4), hist_w(8), hist_lc(16), hist_f(16);
        std::vector<float> wf32((size_t)N * K);
        for (size_t i = 0; i < wf32.size(); ++i) wf32[i] = (float)hW[i];
        for (int m = 0; m < M; m += 13) {
            float a[K];
            for (int k = 0; k < K; ++k) a[k] = (float)hAm[(size_t)m * K + k];
            for (int n = 0; n < N; ++n) {
                double acc = hb[n];
                const float* w = &wf32[(size_t)n * K];
                for (int k = 0; k < K; ++k) acc += (double)a[k] * (double)w[k];
                const double ref = 0.5 * acc * (1.0 + erf(acc * 0.7071067811865476));
                const double got = (double)(float)ho[(size_t)m * N + n];
                const double err = fabs(got - ref);
                if (!(err <= 1e-3 + 1.5e-3 * fabs(ref))) {
                    if (bad < 5) printf("  out[%d][%d] = %g, reference %g\n", m, n, got, ref);
                    ++bad;
                    ++hist_tile[(m / 256) * 24 + n / 128]; ++hist_j[(m % 64) / 16]; ++hist_i[(n % 64) / 16]; ++hist_w[((n % 128) / 64) * 4 + (m % 256) / 64]; ++hist_lc[m % 16]; ++hist_f[n % 16];
                }
                if (err > worst) worst = err;
                ++seen;
            }
        }
        if (bad) {
            printf("  bad per (row tile, 128-column pass):");
            for (int t = 0; t < 32 * 24; ++t) if (hist_tile[t]) printf(" (%d,%d):%d", t / 24, t % 24, hist_tile[t]);
            printf("\n  per chunk j:"); for (int x : hist_j) printf(" %d", x);
            printf("\n  per group i:"); for (int x : hist_i) printf(" %d", x);
            printf("\n  per wave (wm*4+wn):"); for (int x : hist_w) printf(" %d", x);
            printf("\n  per token%%16:"); for (int x : hist_lc) printf(" %d", x);
            printf("\n  per feature%%16:"); for (int x : hist_f) printf(" %d", x);
            printf("\n");
        }
        printf("check: %zu outputs compared, %zu outside 1e-3 + 1.5e-3 |ref|, worst |error| %.3g  -> %s\n", seen, bad, worst, bad ? "FAILED" : "ok");
        return bad ? 1 : 0;
    }
    const double flop = 2.0 * M * N * K;
    for (int rep = 0; rep < 3; ++rep) {
        double g;
        const double us = run(A, W, out, bias, M, N, clk, iters, &g, ring);
        if (ring != M) printf("(output rows wrap at %d: %.0f MB) ", ring, (double)ring * N * 2 / 1e6);
        printf("IL_KO %d: two passes of 128 x 256, the epilogue between the next pass's MFMAs: %7.1f us (%6.1f TFLOP/s, %.3f GHz)\n", IL_KO, us, flop / us / 1e6, g);
    }
    return 0;
}
