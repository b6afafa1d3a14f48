// The full-row residual GEMM:  out[m, 0:768] = half( resid[m, :] * alpha + A[m, :] W^T + bias ),  one workgroup owns ALL 768 columns of
// its 128 rows.  Built for the attention output projection (reference: backbone.py:572 out_proj, :360-362 residual * alpha + LayerNorm):
// K = 768, N = 768 is the one layer product of the 256-tile streaming kernel (gemm.hip) that does not run at the board's power cap -- it
// runs at the memory system, and a third of what it fetches are its A panels read again by the three column tiles of a row panel on three
// CUs (profiles/r05_traffic.json: 1.28 x the algorithmic bytes).  With the whole row in one workgroup
//   * A is fetched exactly once (W, 1.2 MB, is every XCD's L2 resident and streams through LDS once per tile);
//   * the folded LayerNorm's row statistics are finished IN the tile: the epilogue has every 64-column segment sum of a row in LDS, adds
//     them in ln_rowstats_kernel's order and writes (rstd, -mu rstd) itself -- no partial-statistics tensor, no ln_rowstats launch;
//   * the arithmetic is the streaming kernel's EPI 2 operation for operation (same MFMA chain over k, same fused multiply-adds, same
//     statistics helpers): outputs and statistics are BIT-IDENTICAL to gemm256p_kernel<T, 2, LN> + ln_rowstats_kernel (tests).
//
// Shape of the work (gfx950): 512 threads = 8 waves; wave w owns columns [96 w, 96 w + 96) of all 128 rows: 6 x 8 tiles of
// v_mfma_f32_16x16x32 = 192 accumulator registers (of the 256 a wave has at two waves per SIMD), weights as the MFMA A operand (a lane ends up with 4 consecutive columns of one
// row).  K advances in steps of 32: a step's operands are W [768 x 32] = 48 KiB and X [128 x 32] = 8 KiB, 64-byte LDS rows, 16-byte chunk c
// of row r in slot c ^ (3 * ((r >> 2) & 1)) (applied on the LDS-DMA source address and on the ds_read_b128 address: every fragment read is
// conflict-free for the b128 lane groups).  A wave's W rows are PRIVATE to it -- it DMAs them and only it reads them -- so the two W slots
// need no barrier: a wave refills a slot as soon as its own six fragment reads have returned, which keeps almost two k-steps of W in flight
// with two slots.  X is shared: four slots, refilled three steps ahead, one wave instruction per wave and step.  A step is two phases,
//     L: wait (counted vmcnt) for my W(k) and X(k+1) - 6 + 4 fragment reads - issue W(k+2)          M: issue X(k+3) - 48 MFMAs, the other
//        four X fragments read into the first four's registers under them
// separated by raw s_barriers; waves 4-7 run one barrier behind waves 0-3, so on every SIMD one wave is in M while its partner is in L.
// LDS: 2 x 48 + 4 x 8 = 128 KiB of stages; the epilogue's transpose slab aliases them (the pipeline is drained at a tile's end).
//
// MEASURED (profiles/r06b_gemm_row.txt): 224 - 227 us on the attention output projection against 197 - 208 for the streaming kernel + ln_rowstats;
// per tile 3.0 us prologue + 34.5 us K loop (MFMA floor 18.4; the operand stream alone 26.6 = 50 GB/s per CU) + 17.1 us epilogue.  Not the
// default: avexhip_gemm variant 8, or AVEX_AMD_GEMM_ROW=1.
#include <stdlib.h>

#include "common.h"
#include "gemm_epi.h"

#ifndef GEMM_HW_SAT
#define GEMM_HW_SAT 1
#endif
#ifndef GEMM_NT
#define GEMM_NT 1
#endif
#ifndef ROW_KO
#define ROW_KO 0      // diagnostic builds (wrong results): 1 no W DMA, 2 no X DMA, 4 no MFMAs, 8 no fragment reads in L -- what bounds the K loop (scripts/gemm_row_stamps.py)
#endif

// Diagnostic builds only (AVEX_AMD_LIB_SUFFIX=rowst AVEX_AMD_EXTRA_CFLAGS=-DGEMM_ROW_STAMPS=1 python -m avex_amd.build): s_memrealtime (100 MHz)
// at a tile's start, after its prologue wait, at the end of its K loop and of its epilogue, per tile (thread 0 of each workgroup);
// scripts/gemm_row_stamps.py reads them.  The product library contains none of it.
#ifndef GEMM_ROW_STAMPS
#define GEMM_ROW_STAMPS 0
#endif
#if GEMM_ROW_STAMPS
__device__ unsigned long long g_row_stamps[4 * 4096];
extern "C" int avexhip_debug_row_stamps(unsigned long long* host_out, int n_tiles) {
    if (!host_out || n_tiles <= 0) return -1;
    if (n_tiles > 4096) n_tiles = 4096;
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_row_stamps), sizeof(unsigned long long) * 4 * n_tiles) == hipSuccess ? 0 : -2;
}
#define ROW_STAMP(i) do { if (tid == 0 && tile < 4096) g_row_stamps[4 * tile + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define ROW_STAMP(i) do { } while (0)
#endif

namespace {

typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

constexpr int RN = 768, RM = 128, RK = 32;
constexpr int NI = 6, NJ = 8;                        // a wave's 16 x 16 tiles along n (96 columns) and m (128 rows)
constexpr int W_SLOT = RN * RK * 2;                  // 49152
constexpr int X_SLOT = RM * RK * 2;                  // 8192
constexpr int L_X = 2 * W_SLOT;                      // 98304
constexpr int L_VEC = L_X + 4 * X_SLOT;              // 131072: bias' [768], alpha gamma [768] (fp32), resident for the kernel's life
constexpr int L_STATS = L_VEC + 2 * RN * 4;          // 137216: [128 rows][12 segments] (sum, sum of squares)
constexpr int L_ROWST = L_STATS + RM * 12 * 8;       // 149504: [128] (rstd, -mu rstd) of the residual rows (LNR)
constexpr int L_DUMMY = L_ROWST + RM * 8;            // 150528: 8 bytes per lane that nobody reads (the statistics stores of the lanes that hold no segment sum)
constexpr int L_TOTAL = L_DUMMY + 64 * 8;            // 151040
constexpr int SLAB_LD = RN * 4 + 16;                 // 3088-byte slab rows: 16 lanes writing one column of 16 rows hit 16 different bank groups
constexpr int SLAB_BYTES = 32 * SLAB_LD;             // 98816: one slab of 32 rows, over the (drained) stages
static_assert(SLAB_BYTES <= L_VEC && L_TOTAL <= 160 * 1024, "LDS plan");

#define ROW_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define ROW_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define ROW_BAR()                                   \
    __builtin_amdgcn_sched_barrier(0);              \
    __builtin_amdgcn_s_barrier();                   \
    __builtin_amdgcn_sched_barrier(0);

template <typename V>
static __device__ __forceinline__ f32x4 row_mfma(const V& a, const V& b, f32x4 c) {
    if (ROW_KO & 4) { asm volatile("" ::"v"(a), "v"(b)); return c; }
    return mfma16(a, b, c);
}

template <typename T, bool LNR, bool STATS>
__global__ __launch_bounds__(512) void gemm_row_kernel(avx::GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename Half<T>::v8 v8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wid >> 2;                        // waves 4-7 run one barrier behind waves 0-3
    const int KT = p.K / RK;                         // >= 4 (launcher)
    __builtin_assume(KT >= 4);
    const int tiles = (p.M + RM - 1) / RM;
    const T* __restrict__ A = (const T*)p.A;
    const T* __restrict__ W = (const T*)p.W;

    // column vectors, once per workgroup (visible behind the first barrier of the first tile)
    float* vec_b = (float*)(smem + L_VEC);
    float* vec_g = vec_b + RN;
    for (int i = tid; i < RN; i += 512) {
        vec_b[i] = LNR ? p.lnr_beta[i] : p.bias[i];      // LNR: alpha beta + bias, ready-made (GemmArgs::lnr_prefolded)
        if (LNR) vec_g[i] = p.lnr_gamma[i];              //      alpha gamma
    }

    // LDS-DMA lane constants.  One wave instruction fills 16 LDS rows of 64 bytes (lane i -> row i >> 2, slot i & 3); the lane fetches the
    // chunk that belongs in that slot.
    const int drow = lane >> 2;
    const int dchunk = (lane & 3) ^ (3 * ((lane >> 4) & 1));
    const unsigned woff = (unsigned)(((int64_t)drow * p.ldw + dchunk * 8) * 2);      // + the wave's / instruction's row base and k (uniform)
    // fragment reads: lane l takes row (l & 15), chunk (l >> 4) of a 16-row tile
    const int foff = (lane & 15) * 64 + ((((lane >> 4)) ^ (3 * ((lane >> 2) & 1))) << 4);
    const int wfrag = (96 * wid) * 64 + foff;
    unsigned long long ovf_lanes = 0ull;

    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int m0 = tile * RM;
        int xr = 16 * wid + drow;
        const int last = p.M - 1 - m0;
        xr = xr < last ? xr : last;                      // rows past M read row M - 1 (never stored)
        const unsigned xoff = (unsigned)(((int64_t)xr * p.lda + dchunk * 8) * 2);
        const char* wbase = (const char*)(W + (int64_t)(96 * wid) * p.ldw);
        const char* xbase = (const char*)(A + (int64_t)m0 * p.lda);

        auto dma_w = [&](int kt) __attribute__((always_inline)) {      // my 96 weight rows of k-step kt -> W slot kt & 1
            if (ROW_KO & 1) return;
            char* dst = smem + (kt & 1) * W_SLOT + (96 * wid) * 64;
            const char* src = wbase + (int64_t)kt * (RK * 2);
            // scalar base + ONE 32-bit lane offset for all six instructions (left visible, the offset's zero-extension is hoisted out of the
            // loop as six 64-bit register pairs -- twelve registers this kernel does not have; the empty asm keeps it at the instruction)
#pragma unroll
            for (int q = 0; q < NI; ++q) {
                unsigned w0 = woff;
                asm volatile("" : "+v"(w0));      // (per instruction: one shared opaque copy is added to the scalar base ONCE, as a 64-bit vector value, and the six bases follow as vector adds)
                __builtin_amdgcn_global_load_lds((gptr_t*)(src + (int64_t)(16 * q) * p.ldw * 2 + w0), (lptr_t*)(dst + q * 1024), 16, 0, 0);
            }
        };
        auto dma_x = [&](int kt) __attribute__((always_inline)) {      // my 16 activation rows of k-step kt -> X slot kt & 3
            if (ROW_KO & 2) return;
            char* dst = smem + L_X + (kt & 3) * X_SLOT + (16 * wid) * 64;
            unsigned x0 = xoff;
            asm volatile("" : "+v"(x0));
            __builtin_amdgcn_global_load_lds((gptr_t*)(xbase + (int64_t)kt * (RK * 2) + x0), (lptr_t*)dst, 16, 0, 0);
        };

        f32x4 acc[NI][NJ];
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        v8 wf[NI], xf[4];
        if (ROW_KO & 8) {
#pragma unroll
            for (int i = 0; i < NI; ++i) wf[i] = (v8)(T)1.0f;
#pragma unroll
            for (int j = 0; j < 4; ++j) xf[j] = (v8)(T)1.0f;
        }

        ROW_STAMP(0);
        // prologue: in the steady state's issue order  W(k) . X(k+1) . W(k+1) . X(k+2)
        dma_w(0); dma_x(0); dma_x(1); dma_w(1); dma_x(2);
        ROW_VMCNT(7);                                    // W(0), X(0), X(1) are mine and landed
        ROW_BAR();                                       // ... everyone's: X(0), X(1) complete (and the previous tile's epilogue is over)
        if (grp == 1) { ROW_BAR(); }                     // stagger
        ROW_STAMP(1);

        for (int kt = 0; kt < KT; ++kt) {
            // ---- L: fragments of this step; refill my W slot
            if (kt > 0) {
                if (kt < KT - 2) { ROW_VMCNT(7); }       // all but W(kt+1) x 6 and X(kt+2): W(kt) and X(kt+1) have landed
                else if (kt == KT - 2) { ROW_VMCNT(6); }
                else { ROW_VMCNT(0); }
            }
            const char* ws = smem + (kt & 1) * W_SLOT + wfrag;
            const char* xs = smem + L_X + (kt & 3) * X_SLOT + foff;
            if (!(ROW_KO & 8)) {
#pragma unroll
                for (int i = 0; i < NI; ++i) wf[i] = *(const v8*)(ws + i * 1024);
#pragma unroll
                for (int j = 0; j < 4; ++j) xf[j] = *(const v8*)(xs + j * 1024);
            }
            ROW_LGKM0();
            if (kt + 2 < KT) dma_w(kt + 2);
            ROW_BAR();
            // ---- M: 48 MFMAs; the second half of the X fragments arrives under the first half's MFMAs
            if (kt + 3 < KT) dma_x(kt + 3);              // into the slot of X(kt-1): every wave left M(kt-1) before this barrier instance
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                acc[i][0] = row_mfma(wf[i], xf[0], acc[i][0]);
                acc[i][1] = row_mfma(wf[i], xf[1], acc[i][1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            xf[0] = *(const v8*)(xs + 4 * 1024);
            xf[1] = *(const v8*)(xs + 5 * 1024);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                acc[i][2] = row_mfma(wf[i], xf[2], acc[i][2]);
                acc[i][3] = row_mfma(wf[i], xf[3], acc[i][3]);
            }
            __builtin_amdgcn_sched_barrier(0);
            xf[2] = *(const v8*)(xs + 6 * 1024);
            xf[3] = *(const v8*)(xs + 7 * 1024);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                acc[i][4] = row_mfma(wf[i], xf[0], acc[i][4]);
                acc[i][5] = row_mfma(wf[i], xf[1], acc[i][5]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                acc[i][6] = row_mfma(wf[i], xf[2], acc[i][6]);
                acc[i][7] = row_mfma(wf[i], xf[3], acc[i][7]);
            }
            __builtin_amdgcn_s_setprio(0);
            ROW_LGKM0();
            ROW_BAR();
        }
        if (grp == 0) { ROW_BAR(); }                     // re-align: every wave has left the loop, the stages are free
        ROW_STAMP(2);

        // ---- epilogue -----------------------------------------------------------------------------------------------------------
        // The accumulators go through an fp32 LDS slab so that a lane owns 8 consecutive columns of one row and 8 lanes a 64-column segment
        // -- the unit of the streaming kernel's EPI 2, whose operations follow one for one.
        int le = lane;
        asm volatile("" : "+v"(le));                     // (lane constants of the epilogue must not be hoisted above the K loop: they would live, spilled, through it)
        if (GEMM_HW_SAT) AVX_F16_SAT_BEGIN();
        float ovf_mx = 0.f;
        const T* __restrict__ resid = (const T*)(LNR ? p.lnr_y : p.resid_half);
        const int ldres = LNR ? p.ldy : (int)p.ldrh;
        const int ldh = (int)p.ldh;
        int vrows = p.M - m0;
        vrows = vrows < RM ? vrows : RM;
        const __amdgpu_buffer_rsrc_t obuf = buf_rsrc((const T*)p.out_half + (int64_t)m0 * p.ldh, (unsigned)vrows * (unsigned)ldh * 2u);
        const __amdgpu_buffer_rsrc_t rbuf = buf_rsrc(resid + (int64_t)m0 * ldres, (unsigned)vrows * (unsigned)ldres * 2u);
        if (LNR && tid < RM) {                           // the residual rows' (rstd, -mu rstd)
            int m = m0 + tid;
            m = m < p.M ? m : p.M - 1;
            ((float2*)(smem + L_ROWST))[tid] = ((const float2*)p.lnr_rows)[m];
        }
        const float alpha = p.alpha;
        // Four passes of 32 rows (two 16-row accumulator chunks: 48 registers free per pass).  Every wave writes its 96 columns of the 32
        // rows into one fp32 slab, then owns four whole rows of it: 384 (row, 8-column group) pairs = six per lane, flat = 64 q + lane.
        const int wcol = (le & 15) * SLAB_LD + (96 * wid + 4 * (le >> 4)) * 4;      // slab write: row (lane & 15), my 4 columns of n-tile 0
        auto slot_row = [&](int q) __attribute__((always_inline)) -> int {           // which of my four rows lane-slot q is in
            return q == 0 ? 0 : q == 1 ? (le >= 32 ? 1 : 0) : q == 2 ? 1 : q == 3 ? 2 : q == 4 ? (le >= 32 ? 3 : 2) : 3;
        };
        // a pass's residual vectors are requested one pass early, behind the slab writes that free the registers they land in (none ahead
        // exposes the memory latency once per pass; with the requests in front of the writes the kernel spills, and every spill reload is
        // a vmcnt(0): it waits for all stores in flight)
        v8 rhq[4][6];
        auto request = [&](int pp) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const int rs = slot_row(q);
                rhq[pp][q] = buf_ld16<v8>(rbuf, ((32 * pp + 4 * wid + rs) * ldres + 8 * (64 * q + le - 96 * rs)) * 2);
            }
        };
        request(0);
        char* slab = smem;
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                *(f32x4*)(slab + wcol + i * 64) = acc[i][2 * pp];
                *(f32x4*)(slab + 16 * SLAB_LD + wcol + i * 64) = acc[i][2 * pp + 1];
            }
            __builtin_amdgcn_sched_barrier(0);
            if (pp + 1 < 4) request(pp + 1);
            ROW_LGKM0();
            ROW_BAR();
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const int rs = slot_row(q);
                const int row = 4 * wid + rs;                        // of the pass
                const int col8 = 64 * q + le - 96 * rs;
                const char* src = slab + row * SLAB_LD + col8 * 32;
                const f32x4 v0 = *(const f32x4*)src, v1 = *(const f32x4*)(src + 16);
                const f32x4 b0 = *(const f32x4*)(vec_b + 8 * col8), b1 = *(const f32x4*)(vec_b + 8 * col8 + 4);
                const v8 rh = rhq[pp][q];
                f32x4 o0, o1;
                if (LNR) {
                    const f32x4 g0 = *(const f32x4*)(vec_g + 8 * col8), g1 = *(const f32x4*)(vec_g + 8 * col8 + 4);
                    const float2 st = ((const float2*)(smem + L_ROWST))[32 * pp + row];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        o0[e] = __builtin_fmaf(__builtin_fmaf((float)rh[e], st.x, st.y), g0[e], b0[e]) + v0[e];
                        o1[e] = __builtin_fmaf(__builtin_fmaf((float)rh[4 + e], st.x, st.y), g1[e], b1[e]) + v1[e];
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        o0[e] = __builtin_fmaf((float)rh[e], alpha, v0[e] + b0[e]);
                        o1[e] = __builtin_fmaf((float)rh[4 + e], alpha, v1[e] + b1[e]);
                    }
                }
                ovf_see4<T>(ovf_mx, o0); ovf_see4<T>(ovf_mx, o1);
                asm volatile("" : "+v"(ovf_mx));      // the running maximum is taken HERE (left to the scheduler, all 24 lane-slots' maxima sink to the end of the epilogue and 192 output values stay live -- spilled -- until then)
                v8 h;
#pragma unroll
                for (int e = 0; e < 4; ++e) { h[e] = Half<T>::from_hw(o0[e]); h[4 + e] = Half<T>::from_hw(o1[e]); }
                buf_st16<GEMM_NT ? 2 : 0>(h, obuf, ((32 * pp + row) * ldh + 8 * col8) * 2);
                if (STATS) {
                    float s1, s2;
                    stats8(o0, o1, s1, s2);
                    seg8_sum2(s1, s2);
                    // (an address select, not a branch: behind `if ((le & 7) == 0)` the compiler collects all 24 stores of a tile in one block at
                    //  the end of the epilogue and keeps -- spills -- the 48 values until then)
                    const int sidx = (le & 7) == 0 ? L_STATS + ((32 * pp + row) * 12 + (col8 >> 3)) * 8 : L_DUMMY + le * 8;
                    *(float2*)(smem + sidx) = make_float2(s1, s2);
                }
                if (q & 1) __builtin_amdgcn_sched_barrier(0);      // two lane-slots side by side, not six: their temporaries would not fit beside the accumulators
            }
            ROW_LGKM0();
            if (pp + 1 < 4) { ROW_BAR(); }                   // the slab is read: the next pass may overwrite it
        }
        ovf_lanes |= ovf_mask<T>(ovf_mx);
        if (GEMM_HW_SAT) AVX_F16_SAT_END();
        ROW_BAR();                                       // the statistics of all 128 rows are in LDS; the slabs are free (= the next tile's stages)
        if (STATS && tid < RM && m0 + tid < p.M) {
            const f32x4* src = (const f32x4*)(smem + L_STATS) + tid * 6;
            if (p.stats_out) {                           // the partial statistics themselves, [M][12][2]: the contract of GemmArgs::stats_out
                f32x4* dst = (f32x4*)(p.stats_out + (int64_t)(m0 + tid) * 24);
#pragma unroll
                for (int q = 0; q < 6; ++q) dst[q] = src[q];
            }
            if (p.rows_out) {                            // ln_rowstats_kernel's arithmetic (elementwise.hip), on the same partial sums
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int q = 0; q < 6; ++q) { const f32x4 v = src[q]; s1 += v[0] + v[2]; s2 += v[1] + v[3]; }
                const float inv = 1.0f / (float)(64 * 12);
                const float mu = s1 * inv;
                const float var = fmaxf(__builtin_fmaf(-mu, mu, s2 * inv), 0.f);
                const float rstd = __builtin_amdgcn_rsqf(var + p.rows_eps);
                ((float2*)p.rows_out)[m0 + tid] = make_float2(rstd, -mu * rstd);
            }
        }
        ROW_STAMP(3);
        // (the next tile's prologue barrier orders these LDS reads before the first refill that could land on them: L_STATS lies above the stages)
    }
    ovf_commit<T>(p.ovf, ovf_lanes);
}

template <typename T, bool LNR, bool STATS>
int launch_row(const avx::GemmArgs& a, int grid, hipStream_t s) {
    AVX_ENSURE_LDS((gemm_row_kernel<T, LNR, STATS>), L_TOTAL);
    hipLaunchKernelGGL((gemm_row_kernel<T, LNR, STATS>), dim3(grid), dim3(512), L_TOTAL, s, a);
    AVX_LAUNCH_CHECK();
    return AVEXHIP_OK;
}

template <typename T>
int launch_row_any(const avx::GemmArgs& a, hipStream_t s) {
    int n_cu = 256;
    { const int rc_ = avx::device_cu_count(&n_cu); if (rc_ != AVEXHIP_OK) return rc_; }
    const int tiles = (a.M + RM - 1) / RM;
    int grid = tiles < n_cu ? tiles : n_cu;
    if (const char* fg = getenv("AVEX_AMD_GEMM_GRID")) { const int g = atoi(fg); if (g >= 1 && g < grid) grid = g; }      // tests: many tiles per workgroup
    const bool stats = a.stats_out || a.rows_out;
    if (a.lnr_y) return stats ? launch_row<T, true, true>(a, grid, s) : launch_row<T, true, false>(a, grid, s);
    return stats ? launch_row<T, false, true>(a, grid, s) : launch_row<T, false, false>(a, grid, s);
}

}  // namespace

namespace avx {

// what the full-row kernel takes: the streaming kernel's fast residual form (EPI 2) at N = 768
bool gemm_row_ok(const GemmArgs& a) {
    const bool scaled = a.half_scale != 0.f && a.half_scale != 1.f;
    return a.N == RN && a.K % RK == 0 && a.K >= 4 * RK && a.M >= 1 && a.out_half && a.bias && !a.out_f32 && !a.out_raw && !a.pool_part && !a.resid && !a.row_zero &&
           !scaled && (a.resid_half || a.lnr_y) && !(a.resid_half && a.lnr_y) && !a.gelu && !a.ln_rows && !a.post_ln_w && !(a.n_store > 0 && a.n_store < a.N) && !a.a_scale &&
           a.lda % 8 == 0 && a.ldw % 8 == 0 && a.ldh % 8 == 0 && (a.lnr_y ? (a.ldy % 8 == 0 && a.lnr_rows && a.lnr_gamma && a.lnr_beta && a.lnr_prefolded) : a.ldrh % 8 == 0) &&
           (int64_t)RM * a.ldh * 2 < (1ll << 31) && (int64_t)RM * (a.lnr_y ? a.ldy : a.ldrh) * 2 < (1ll << 31) && (int64_t)RM * a.lda * 2 < (1ll << 31) && (int64_t)RN * a.ldw * 2 < (1ll << 31);
}

int gemm_row(const GemmArgs& a, int dtype, hipStream_t s) {
    AVX_REQUIRE(gemm_row_ok(a), "gemm_row: takes N = 768, K %% 32 == 0, K >= 128, a half output with bias and a half (or LayerNorm-folded) residual (N=%d K=%d)", a.N, a.K);
    if (dtype == AVEXHIP_F16) return launch_row_any<_Float16>(a, s);
    if (dtype == AVEXHIP_BF16) return launch_row_any<__bf16>(a, s);
    avexhip_set_error("gemm_row: unknown dtype %d", dtype);
    return AVEXHIP_ERR_INVALID;
}

}  // namespace avx
