// Variant 3 of the gated relative-position-bias attention (head_dim 64, up to 512 tokens): the persistent, LDS-DMA-streamed
// kernel of attention.hip (variant 2) rebuilt on v_mfma_f32_16x16x32.
//
// Restates _MultiheadAttention.forward (avex/models/beats/backbone.py:494-574) after the q/k/v projections, exactly as attention.hip
// does:  softmax( q k^T / 8 + gate(b,h,i) * bias[h, j-i]  [+ -inf on padded keys] ) v,  gate from backbone.py:543-551.
//
// Why another form (round 5).  Variant 2's SQ counters: matrix pipe 31 % busy, vector pipe 50 %, both at once 18 %
// (profiles/r02c_attention_analysis.txt); its packed-fp32 softmax arithmetic issues at 10.7 cycles beside 32x32x16 MFMAs and at 4.2
// beside 16x16x32 ones (scripts/micro/valu_rates.hip), and under the board's power cap the 16x16x32 shape costs ~10 % less energy per
// flop (profiles/r01h_mfma_power.txt).  What changes against variant 2:
//   * both products are 16x16x32: S^T[key][query] = K Q^T per (16 keys x 16 queries) block, O^T[d][query] += V^T P^T per (16 d x 16
//     queries) block.  A wave owns 64 queries as FOUR blocks of 16 (lane c = lane & 15 is the query, lane group g = lane >> 4 holds keys
//     4g .. 4g+3 of a 16-key block); the two 16-key blocks of a 32-key tile give a lane 8 scores per query block, which ARE the eight
//     k-slots 8g .. 8g+7 of the P^T operand of the second product (k-slot 8g + j <-> key (j < 4 ? 4g + j : 16 + 4g + j - 4)); the V^T
//     operand is gathered with the same map by two ds_read_b64_tr_b16 per 16 d.  P never leaves registers.
//   * the tile body is eight stages of four MFMAs (S of query blocks 0..3, then P V of 0..3) with another block's vector work between
//     the MFMAs of each stage: exponentials of block n under S of block n + 1 (block 3 under P V of block 0), the next key tile's
//     accumulator starts (gate * bias - m) under P V of blocks 1 and 2.
//   * the Toeplitz bias needs TWO new 16-byte LDS reads per key tile and wave (variant 2: four): the run a lane needs for (key block
//     b, query block n) of key tile k depends on 2k + b - n only, so five runs are live and two are new per tile.
//   * a row's first key tile no longer re-does its exponentials: the running reference starts at 0 and stays there while every
//     half-row sum is inside [2^-6, 2^12) (f16-safe numerators, no subnormal loss); only rows outside take the rescale path.
//   * V image in LDS: [8-key group][16-d block][key & 7][32 B], so a 32-lane half's transposed read covers 256 contiguous bytes
//     (conflict-free); K image as in variant 2 (128-B rows, 16-byte chunks XOR-swizzled by (row >> 1) & 7: conflict-free for the
//     16-row x 4-chunk read of this operand map as well, checked with the bank rules of the LDS table).
// Longer clips stay on variant 2, except 513 .. 544 tokens without a bias table (EAT): the XT instantiation below.
#include <stdlib.h>

#include "common.h"

namespace {

typedef int a3_i32x2 __attribute__((ext_vector_type(2)));
typedef int a3_i32x4 __attribute__((ext_vector_type(4)));
template <int N> struct a3_ic { static constexpr int value = N; };
template <typename T> struct a3_v2;
template <> struct a3_v2<_Float16> { typedef _Float16 type __attribute__((ext_vector_type(2))); };
template <> struct a3_v2<__bf16> { typedef __bf16 type __attribute__((ext_vector_type(2))); };

constexpr int A3_TMAX = 512;
constexpr int A3_KBUF = 32768;                            // 256 keys x 128 B
constexpr int A3_HALF = 65536;                            // K + V of 256 keys
constexpr int A3_TPAD = 64;                               // floats in front of each shifted copy (indices down to -64 are read, never used)
constexpr int A3_TLD = A3_TPAD + 1040;                    // floats per shifted copy of the bias row
constexpr int A3_TAB_OFF = 2 * A3_HALF;
// XT (no bias table only): up to 32 keys beyond 512 -- EAT's 513 tokens -- ride in the second half's phase as a NINTH key tile instead of a phase of
// their own: buffers of 288 keys (the LDS the bias table would take is free), a key mask of 544 entries; the queries beyond 512 go to the tail kernel
constexpr int A3X_KBUF = 9 * 4096;                         // 288 keys x 128 B
constexpr int A3X_HALF = 2 * A3X_KBUF;
constexpr int A3X_KN = 544;                                // key-mask entries per item
constexpr int A3X_KADD_OFF = 2 * A3X_HALF;
constexpr int A3X_GW_OFF = A3X_KADD_OFF + 2 * A3X_KN * 4;
constexpr int ATT3X_LDS = A3X_GW_OFF + 136 * 4;
constexpr int A3_TAB_BYTES = 4 * A3_TLD * 4;
constexpr int A3_KADD_OFF = A3_TAB_OFF + A3_TAB_BYTES;
constexpr int A3_GW_OFF = A3_KADD_OFF + 2 * A3_TMAX * 4;
constexpr int ATT3_LDS = A3_GW_OFF + 136 * 4;
constexpr float A3_THR = 8.f;                             // deferred reference: it moves when a tile's maximum has grown past 2^8
constexpr float A3_SUM_HI = 4096.f;                       // a half-row sum below 2^12 proves every numerator of the lane is below 2^12
constexpr float A3_SUM_LO = 0.015625f;                    // first tile: a sum above 2^-6 means the row's numerators are not all deep in the subnormals

#ifndef ATT3_DMA_NT
#define ATT3_DMA_NT 0    // 1: the K / V LDS-DMA with the nt policy (every byte is read by one workgroup, once).  MEASURED: level (256-clip step 24.910 vs 24.903 ms in one process)
#endif
#ifndef ATT3_PRIO
#define ATT3_PRIO 1      // 1: waves 4-7 at s_setprio 1 (one static raise); 2: waves 0-3 instead; 0: none
#endif
#ifndef ATT3_WG_STAGGER
#define ATT3_WG_STAGGER 0   // workgroup w starts ((w >> 3) & 7) x this many 64-cycle sleeps late: the CUs' item ends (64 KiB of stores each) stop coinciding (A/B builds)
#endif
#ifndef ATT3_FULL_LINES
#define ATT3_FULL_LINES 0   // 1: output stores as whole 128-byte lines (see store_item; measured level with the default: 285.2 vs 285.1 us, profiles/r05a_attention16.txt)
#endif
#ifndef ATT3_STAGGER
#define ATT3_STAGGER 0   // waves 4-7 enter each phase's key tiles this many 64-cycle sleeps late (A/B builds)
#endif
// Diagnostics of the standalone harness (scripts/micro/att16_bench.hip); the library is built with both at 0, which compiles them out.
#ifndef A3_KO
#define A3_KO 0          // knock-outs: 1 no range checks, 2 no exponentials, 4 no S MFMAs, 8 no P V MFMAs, 16 no accumulator starts, 32 no scheduling fences
#endif
#ifndef A3_STAMPS
#define A3_STAMPS 0      // 1: s_memtime stamps per phase and key tile into g_a3_stamps
#endif
#if A3_STAMPS
__device__ unsigned long long g_a3_stamps[3 * 16 * 8 * 16 * 16];      // [block < 16][wave][phase < 16][16]; with A3_STAMPS == 2 two more copies: the stage stamps of key tiles 1 and 6
#endif

// all-reduce over the four 16-lane rows of a wave (the lanes that share a query row) without LDS: two lane swaps
static __device__ __forceinline__ float a3_rows_sum(float x) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);      // {x0 x0 x2 x2}, {x1 x1 x3 x3}
    const float y = __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
    const unsigned v = __builtin_bit_cast(unsigned, y);
    const auto q = __builtin_amdgcn_permlane32_swap(v, v, false, false);      // {lo lo}, {hi hi}
    return __builtin_bit_cast(float, (unsigned)q[0]) + __builtin_bit_cast(float, (unsigned)q[1]);
}
static __device__ __forceinline__ float a3_rows_max(float x) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float y = fmaxf(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
    const unsigned v = __builtin_bit_cast(unsigned, y);
    const auto q = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return fmaxf(__builtin_bit_cast(float, (unsigned)q[0]), __builtin_bit_cast(float, (unsigned)q[1]));
}

static __device__ __forceinline__ void a3_dma16(const void* src, const char* lds_dst) {
    // (inline assembly on purpose: see a2_dma16 in attention.hip -- behind the builtin the compiler drains vmcnt in front of every LDS read)
    const unsigned lds = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) const char*)lds_dst;
#if ATT3_DMA_NT
    asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off nt" ::"v"(src), "s"(lds) : "memory");
#else
    asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds) : "memory");
#endif
}

template <typename T, bool BIAS, bool XT = false>
__global__ __launch_bounds__(512) void attention3_kernel(const T* __restrict__ qkv, int Tn, int Tq, int H, int Bc, int per_block,
                                                        const float* __restrict__ bias_tab, const float* __restrict__ grep_w,
                                                        const float* __restrict__ grep_b, const float* __restrict__ grep_a,
                                                        const uint8_t* __restrict__ key_pad, T* __restrict__ out, int q_log2e) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename Half<T>::v8 v8;
    typedef typename Half<T>::v4 v4;
    static_assert(!XT || !BIAS, "the nine-tile last phase takes the bias table's LDS");
    // Tn = keys (tokens of the clip), Tq = query rows this kernel computes (= Tn, or 512 with XT: the rows beyond go to the tail kernel)
    constexpr int NB = 4, NW = 8, NT = 512, NKT = XT ? 9 : 8;
    constexpr int KBUF = XT ? A3X_KBUF : A3_KBUF, HALF = XT ? A3X_HALF : A3_HALF, KN = XT ? A3X_KN : A3_TMAX;
    float* tab = (float*)(smem + A3_TAB_OFF);
    float* kadd = (float*)(smem + (XT ? A3X_KADD_OFF : A3_KADD_OFF));
    T* gwh = (T*)(smem + (XT ? A3X_GW_OFF : A3_GW_OFF));                     // gate weights: 4 rows of 64 halves
    float* gwb = (float*)(smem + (XT ? A3X_GW_OFF : A3_GW_OFF) + 512);      // the gate's two biases

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g = lane >> 4;
    const int n_units = Bc * H;
    const int w0 = blockIdx.x * per_block;
    const int w1 = w0 + per_block < n_units ? w0 + per_block : n_units;
    if (w0 >= w1) return;
    const int E = H * 64;
    const int64_t ld = 3 * (int64_t)E;
    const float NEG_INF = -__builtin_inff();
    const int nh = Tn > 256 ? 2 : 1;
    const int np = (w1 - w0) * nh;
    const int nkt = (Tn + 31) >> 5;
    const int Q0 = 64 * wave;
    const bool has_q = Q0 < Tq;                           // wave-uniform

    int h_cur = w0 / Bc, b_cur = w0 - h_cur * Bc;
    int half = 0;
    int h_ld = h_cur, b_ld = b_cur, half_ld = 0;
    int item_par = 0;

    // (the lane index goes through an empty asm in the three address-heavy blocks -- DMA issue, Q loads, output stores: the compiler would
    //  otherwise hoist their per-lane 64-bit offsets out of the phase loop, run out of registers and SPILL them, and a scratch reload drains
    //  vmcnt -- i.e. waits for the whole next K/V buffer -- in the middle of a phase: 4.5k cycles per phase and 12k per item, measured)
    // One eighth of a wave's share of phase ph's buffer: piece u (0 .. 3) = one 1 KiB K piece and one 1 KiB V piece.  The pieces of phase
    // ph + 1 go out INSIDE phase ph's first four key tiles (one per tile, in the stage that has no vector work), not as a burst at the phase
    // boundary where every wave of the CU issued its eight at once and nothing else ran (2 - 4k cycles per phase, measured).
    auto dma_piece = [&](int ph, int u) __attribute__((always_inline)) {
        const T* base = qkv + (int64_t)b_ld * Tn * ld + h_ld * 64;
        char* buf = smem + (ph & 1) * HALF;
        int lane = tid & 63;
        asm volatile("" : "+v"(lane));
        const int ri = 4 * wave + u;                                   // 8-key group inside the half
        const int kl = 8 * ri + (lane >> 3);
        int key = half_ld * 256 + kl;
        key = key < Tn ? key : Tn - 1;                                 // clamped rows are masked by kadd
        const int chunk = (lane & 7) ^ ((kl >> 1) & 7);
        a3_dma16(base + (int64_t)key * ld + E + chunk * 8, buf + ri * 1024);
        int vkey = half_ld * 256 + 8 * ri + ((lane >> 1) & 7);         // piece = [16-d block = lane >> 4][key & 7][two 16-byte chunks]
        vkey = vkey < Tn ? vkey : Tn - 1;
        a3_dma16(base + (int64_t)vkey * ld + 2 * E + 16 * (lane >> 4) + 8 * (lane & 1), buf + KBUF + ri * 1024);
        if (XT && u == 3 && half_ld == nh - 1 && wave < 4) {               // the ninth key tile: four more 8-key groups, one per wave 0 .. 3, with the wave's last piece
            const int rx = 32 + wave;
            const int klx = 8 * rx + (lane >> 3);
            int kx = half_ld * 256 + klx;
            kx = kx < Tn ? kx : Tn - 1;
            a3_dma16(base + (int64_t)kx * ld + E + ((lane & 7) ^ ((klx >> 1) & 7)) * 8, buf + rx * 1024);
            int vx = half_ld * 256 + 8 * rx + ((lane >> 1) & 7);
            vx = vx < Tn ? vx : Tn - 1;
            a3_dma16(base + (int64_t)vx * ld + 2 * E + 16 * (lane >> 4) + 8 * (lane & 1), buf + KBUF + rx * 1024);
        }
    };
    auto dma_advance = [&]() __attribute__((always_inline)) {
        if (++half_ld == nh) { half_ld = 0; if (++b_ld == Bc) { b_ld = 0; ++h_ld; } }
    };
    auto write_kadd = [&](int slot, int b) __attribute__((always_inline)) {
        for (int j = tid; j < KN; j += NT) {
            bool ok = j < Tn;
            if (ok && key_pad) ok = key_pad[(int64_t)b * Tn + j] == 0;
            kadd[slot * KN + j] = ok ? 0.f : NEG_INF;
        }
    };

    v8 qf[NB][2];
    auto load_q = [&](int h, int b) __attribute__((always_inline)) {
        const T* base = qkv + (int64_t)b * Tn * ld + h * 64;
        int lane = tid & 63;
        asm volatile("" : "+v"(lane));
        const int c = lane & 15, g = lane >> 4;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            int q = Q0 + 16 * nb + c;
            q = q < Tq ? q : Tq - 1;                     // clamped for loads; stores are masked
            const T* src = base + (int64_t)q * ld + 8 * g;
            // Inline assembly: the compiler must not know these loads.  It cannot count the conditional output stores issued behind them, so any
            // wait IT places for a Q register waits for every store as well (and an asm statement that redefines the registers gets a vmcnt(0)
            // in front): the kernel waits for them itself, at the next phase boundary (A3_WAIT_Q below).
            asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:64" : "=&v"(qf[nb][0]), "=&v"(qf[nb][1]) : "v"(src) : "memory");
        }
    };
    if (ATT3_WG_STAGGER > 0) { for (int i = 0; i < (int)((blockIdx.x >> 3) & 7); ++i) __builtin_amdgcn_s_sleep(ATT3_WG_STAGGER); }
    if ((ATT3_PRIO == 1 && wave >= NW / 2) || (ATT3_PRIO == 2 && wave < NW / 2)) __builtin_amdgcn_s_setprio(1);
    load_q(h_cur, b_cur);                                // older than the DMA below: waiting for it never waits for the DMA
#pragma unroll
    for (int u = 0; u < 4; ++u) dma_piece(0, u);
    dma_advance();
    write_kadd(0, b_cur);
    if (tid < 64) {
        float a = 0.f, bb = 0.f;
        if (grep_w) {
            a = (grep_w[0 * 64 + tid] + grep_w[1 * 64 + tid]) + (grep_w[2 * 64 + tid] + grep_w[3 * 64 + tid]);
            bb = (grep_w[4 * 64 + tid] + grep_w[5 * 64 + tid]) + (grep_w[6 * 64 + tid] + grep_w[7 * 64 + tid]);
        }
        // The gate's two dot products q . wa, q . wb run on the matrix pipe against the SCALED query fragment (q log2(e) / 8 in either
        // mode of q_log2e), so the weights take 8 ln 2; each is split into two halves-type rows hi + lo (hi = the rounded weight, lo = the
        // rounded rest: 22 significant bits for f16, 16 for bf16), rows {wa_hi, wb_hi, wa_lo, wb_lo}.
        const float wsc = 8.0f * 0.6931471805599453f;
        const float wa = a * wsc, wb = bb * wsc;
        const T wah = (T)wa, wbh = (T)wb;
        gwh[0 * 64 + tid] = wah; gwh[1 * 64 + tid] = wbh;
        gwh[2 * 64 + tid] = (T)(wa - (float)wah); gwh[3 * 64 + tid] = (T)(wb - (float)wbh);
        if (tid == 0) {
            gwb[0] = grep_w ? (grep_b[0] + grep_b[1]) + (grep_b[2] + grep_b[3]) : 0.f;
            gwb[1] = grep_w ? (grep_b[4] + grep_b[5]) + (grep_b[6] + grep_b[7]) : 0.f;
        }
    }
    if (tid < 4 * A3_TPAD) tab[(tid >> 6) * A3_TLD + (tid & 63)] = 0.f;      // the front pads (read, never used: finite)

    int h_tab = -1;
    f32x2 gm[NB];                                        // per query block: {gate, running reference m} (one register pair: see acc_start)
    float l_run[NB];
    unsigned long long refm[NB];                         // lane mask: the row's reference has been looked at (0 accepted, or set to a tile maximum)
    f32x4 o[NB][4];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        gm[nb] = (f32x2){1.f, 0.f}; l_run[nb] = 0.f; refm[nb] = 0ull;
#pragma unroll
        for (int db = 0; db < 4; ++db) o[nb][db] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const int v_lane = 1024 * (g >> 1) + 128 * (g & 1) + 32 * (c >> 2) + 8 * (c & 3);     // transposed-read address, per lane
    const float cs = q_log2e ? 0.125f : 0.125f * 1.4426950408889634f;

    // Output of item (h, b): lane (c, g) holds query row c's d = 16 db + 4 g + 0..3 of every 16-d block; one v_permlane16_swap per register of a
    // block pair gives the lanes of even g the 8 consecutive d (8 (g >> 1) ..) of block 2 p and those of odd g the same of block 2 p + 1:
    // a row is written in 16-byte pieces, two stores per 16-query block.
    int n_st = 0;
    auto store_item = [&](int h, int b) __attribute__((always_inline)) {
        n_st = 0;
        if (!has_q) return;
        int lane = tid & 63;
        asm volatile("" : "+v"(lane));
        const int c = lane & 15, g = lane >> 4;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) n_st += Q0 + 16 * nb < Tq ? 2 : 0;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            if (Q0 + 16 * nb >= Tq) continue;                              // wave-uniform
            const float l_tot = a3_rows_sum(l_run[nb]);
            const float inv = __builtin_amdgcn_rcpf(l_tot);
            const f32x2 inv2 = {inv, inv};
            typedef typename a3_v2<T>::type v2;
            // two halves per register; the empty asm keeps each conversion a plain packed multiply + packed convert (left alone, hipcc rebuilt the
            // eight values of a store from mixed-precision FMAs, packs and align-bits: 26 instructions per store instead of 10)
            auto pk = [&](const f32x4 v, int hi) __attribute__((always_inline)) -> unsigned {
                const f32x2 x = (f32x2){v[2 * hi], v[2 * hi + 1]} * inv2;
                // plain conversions: an output row is a convex combination of V rows, it cannot leave the operand type's range
                const v2 h = {(T)x[0], (T)x[1]};
                unsigned u = __builtin_bit_cast(unsigned, h);
                asm volatile("" : "+v"(u));
                return u;
            };
            // after the swaps lane (c, g) owns two 16-byte chunks of row c: chunk k0 = 2 (g & 1) + (g >> 1) of the row's first 64 bytes (w[0]) and the
            // same chunk of its second 64 bytes (w[1])
            int w[2][4];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const auto s0 = __builtin_amdgcn_permlane16_swap(pk(o[nb][2 * p], 0), pk(o[nb][2 * p + 1], 0), false, false);
                const auto s1 = __builtin_amdgcn_permlane16_swap(pk(o[nb][2 * p], 1), pk(o[nb][2 * p + 1], 1), false, false);
                w[p][0] = (int)s0[0]; w[p][1] = (int)s1[0]; w[p][2] = (int)s0[1]; w[p][3] = (int)s1[1];
            }
#if ATT3_FULL_LINES
            // Full 128-byte lines per store instruction (8 rows x 128 B instead of 16 rows x 64 B; the store path is issue-bound at the end of an
            // item and half lines cost it twice the requests): lanes c and c + 8 of a 16-lane row trade one chunk -- lane c < 8 gives its second-half
            // chunk and takes lane c + 8's first-half chunk -- so the first store writes rows 0 .. 7 whole and the second rows 8 .. 15.  Three DPP
            // moves per register (row rotate by 8 with a bank mask).
            a3_i32x4 wa, wb;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int t = __builtin_amdgcn_update_dpp(0, w[0][e], 0x128, 0xF, 0xF, false);            // t[i] = w0[(i + 8) % 16]
                wa[e] = __builtin_amdgcn_update_dpp(w[0][e], w[1][e], 0x128, 0xF, 0xC, false);            // lanes 8-15: w1 of lane i - 8; lanes 0-7 keep w0
                wb[e] = __builtin_amdgcn_update_dpp(w[1][e], t, 0xE4, 0xF, 0x3, false);                   // lanes 0-7: w0 of lane i + 8; lanes 8-15 keep w1
            }
            const int qrow = Q0 + 16 * nb + (c & 7);
            T* orow = out + ((int64_t)b * Tn + qrow) * E + h * 64 + 16 * (g & 1) + 8 * (g >> 1) + 32 * (c >> 3);
            if (qrow < Tq) *(a3_i32x4*)orow = wa;
            if (qrow + 8 < Tq) *(a3_i32x4*)(orow + 8 * (int64_t)E) = wb;
#else
            const int qrow = Q0 + 16 * nb + c;
            T* orow = out + ((int64_t)b * Tn + qrow) * E + h * 64 + 16 * (g & 1) + 8 * (g >> 1);
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const a3_i32x4 wv = {w[p][0], w[p][1], w[p][2], w[p][3]};
                if (qrow < Tq) *(a3_i32x4*)(orow + 32 * p) = wv;
            }
#endif
        }
    };

    for (int ph = 0; ph < np; ++ph) {
        const bool last_half = half == nh - 1;
        const bool more_items = ph + (nh - half) < np;
#define A3_PT(i) if (A3_STAMPS) { __builtin_amdgcn_sched_barrier(0); pt[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
        unsigned long long pt[16];
        if (A3_STAMPS) { for (int i = 0; i < 16; ++i) pt[i] = 0; pt[15] = __builtin_amdgcn_s_memrealtime(); }
        A3_PT(0)
        // phase boundary: this wave's DMA for phase ph (issued one phase ago) has landed and every wave has finished reading the other
        // buffer.  When the previous phase ended an item, the next item's 8 Q loads and this item's stores are younger than that DMA.
        // At an item's first phase the youngest outstanding operations are the previous item's n_st output stores, issued after the next item's
        // Q loads: the counted wait covers the DMA and the Q loads and lets the stores drain under the tiles.  The statement DEFINES the Q fragment
        // for the compiler (the loads themselves are inline assembly, see load_q): ONE statement, so that the eight registers have one definition
        // (five statements in five branches cost sixteen register copies each and a compiler-placed vmcnt(0) in front).
        {
            const int nwait = (half == 0 && ph > 0) ? n_st : 0;      // wave-uniform, one of 0 2 4 6 8
            asm volatile("s_cmp_eq_u32 %8, 8\n\ts_cbranch_scc1 1f\n\t"
                         "s_cmp_eq_u32 %8, 6\n\ts_cbranch_scc1 2f\n\t"
                         "s_cmp_eq_u32 %8, 4\n\ts_cbranch_scc1 3f\n\t"
                         "s_cmp_eq_u32 %8, 2\n\ts_cbranch_scc1 4f\n\t"
                         "s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_branch 5f\n"
                         "1:\n\ts_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_branch 5f\n"
                         "2:\n\ts_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_branch 5f\n"
                         "3:\n\ts_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_branch 5f\n"
                         "4:\n\ts_waitcnt vmcnt(2) lgkmcnt(0)\n"
                         "5:"
                         : "+v"(qf[0][0]), "+v"(qf[0][1]), "+v"(qf[1][0]), "+v"(qf[1][1]), "+v"(qf[2][0]), "+v"(qf[2][1]), "+v"(qf[3][0]), "+v"(qf[3][1])
                         : "s"(nwait)
                         : "scc", "memory");
        }
        __builtin_amdgcn_s_barrier();
        A3_PT(1)
        if (half == 0) {
            if (BIAS && h_cur != h_tab) {                // workgroup-uniform
                for (int r = tid; r < 1040; r += NT) {
                    float v = 0.f;
                    if (r < 2 * Tn - 1) v = bias_tab[(int64_t)h_cur * (2 * Tn - 1) + r] * 1.4426950408889634f;
#pragma unroll
                    for (int sft = 0; sft < 4; ++sft)
                        if (r - sft >= 0) tab[sft * A3_TLD + A3_TPAD + (r - sft)] = v;
                }
                h_tab = h_cur;
                __syncthreads();
            }
            if (more_items) {                            // the next item's key mask, read two barriers from now
                int bn = b_cur + 1;
                bn = bn == Bc ? 0 : bn;
                write_kadd(item_par ^ 1, bn);
            }
            // this item's fragment arrives unscaled; from here on it carries the score scale (1/8 is exact in the operand type; log2(e) is not
            // -- bf16: 0.18 % off, a temperature error on every logit -- so the handles fold it into W_q in fp32: q_log2e)
            const T qs = (T)cs;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int e8 = 0; e8 < 8; ++e8) qf[nb][s][e8] = qf[nb][s][e8] * qs;
            // gate(b, h, i) = ga (gb a_h - 1) + 2, (ga, gb) = sigmoid(q . wa + ba, q . wb + bb) (backbone.py:543-551): two MFMAs per query
            // block against the four weight rows repeated down the 16 operand rows, so EVERY lane ends with its query's four partial products
            // in its own accumulator registers -- no cross-lane traffic (as vector code: 60 instructions and four lane exchanges per block)
            if (grep_w) {
                const v8 gA0 = *(const v8*)(gwh + (c & 3) * 64 + 8 * g), gA1 = *(const v8*)(gwh + (c & 3) * 64 + 32 + 8 * g);
                const float ba = gwb[0], bb = gwb[1], ah = grep_a[h_cur];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    f32x4 acc = mfma16(gA0, qf[nb][0], (f32x4){0.f, 0.f, 0.f, 0.f});
                    acc = mfma16(gA1, qf[nb][1], acc);
                    const float sa = (acc[0] + acc[2]) + ba, sb = (acc[1] + acc[3]) + bb;
                    const float ga = __builtin_amdgcn_rcpf(1.f + __expf(-sa)), gb = __builtin_amdgcn_rcpf(1.f + __expf(-sb));
                    gm[nb][0] = ga * (gb * ah - 1.f) + 2.f;
                }
            } else {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) gm[nb][0] = 1.f;
            }
        }
        const bool do_dma = ph + 1 < np;
        int dma_done = 0;                                 // pieces of phase ph + 1 this wave has issued (wave-uniform)
        A3_PT(2)
        if (has_q) {
            const char* Kb = smem + (ph & 1) * HALF;
            const float* kad = kadd + item_par * KN + half * 256 + 4 * g;     // + 32 kt + 16 kb: the lane's four keys of a block
            const int kt_lim = (XT && last_half) ? 9 : 8;
            int kt_end = nkt - half * 8 < kt_lim ? nkt - half * 8 : kt_lim;
            const char* kp[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) kp[s] = Kb + c * 128 + (((4 * s + g) ^ ((c >> 1) & 7)) << 4);
            // bias run of (key tile k, key block b, query block n): floats tab[e0 + 16 (2k + b - n) + 0..3], e0 = 4g - (Q0 + c) + (Tn - 1) + 256 half
            const int e0 = 4 * g - (Q0 + c) + (Tn - 1) + 256 * half;
            const float* tp = tab + (e0 & 3) * A3_TLD + A3_TPAD + (e0 & ~3);
            const unsigned vaddr = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) const char*)(Kb + KBUF + v_lane);

            if (ATT3_STAGGER > 0 && wave >= NW / 2) __builtin_amdgcn_s_sleep(ATT3_STAGGER);
            v8 kf[2][2];                                 // [k-step][key block]
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) kf[s][kb] = *(const v8*)(kp[s] + 2048 * kb);
            f32x4 run[5];                                // slot (w + 3) % 5 holds run w = 2k + b - n
            f32x4 Sq[NB][2];
            auto masked_at = [&](int ktl) __attribute__((always_inline)) { return key_pad != nullptr || (half * 8 + ktl) * 32 + 32 > Tn; };
            // accumulator start of (query block, key block): gate * bias - m on two column pairs.  One packed FMA each, both scalars from ONE
            // register pair: source 0 takes gm's low half (the gate) for both results, source 2 its high half (m) for both, negated.  (hipcc, given
            // the same arithmetic, built {gate, gate} pairs with two moves per block and fell back to scalar FMAs for a third of them.)  The
            // select [0,0,1] is not the form isa_lint rejects (second source's high half for the low result).
            auto acc_start = [&](int nb, int kb, const f32x4 t4) __attribute__((always_inline)) {
                const f32x2 ta = {t4[0], t4[1]}, tb = {t4[2], t4[3]};
                f32x2 ea, eb;
                asm("v_pk_fma_f32 %0, %1, %2, %1 op_sel:[0,0,1] op_sel_hi:[0,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(ea) : "v"(gm[nb]), "v"(ta));
                asm("v_pk_fma_f32 %0, %1, %2, %1 op_sel:[0,0,1] op_sel_hi:[0,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(eb) : "v"(gm[nb]), "v"(tb));
                Sq[nb][kb] = (f32x4){ea[0], ea[1], eb[0], eb[1]};
            };
            auto acc_plain = [&](int nb) __attribute__((always_inline)) {
                const float nm = -gm[nb][1];
                Sq[nb][0] = (f32x4){nm, nm, nm, nm};
                Sq[nb][1] = (f32x4){nm, nm, nm, nm};
            };
            auto mask_acc = [&](int ktl) __attribute__((always_inline)) {
                const f32x4 k0 = *(const f32x4*)(kad + 32 * ktl), k1 = *(const f32x4*)(kad + 32 * ktl + 16);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) { Sq[nb][0] += k0; Sq[nb][1] += k1; }
            };
            if (BIAS) {
#pragma unroll
                for (int w = -3; w <= 1; ++w) run[w + 3] = *(const f32x4*)(tp + 16 * w);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) acc_start(nb, kb, run[kb - nb + 3]);
            } else {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc_plain(nb);
            }
            if (masked_at(0)) mask_acc(0);
#define A3_FENCE() do { if (!(A3_KO & 32)) __builtin_amdgcn_sched_barrier(0); } while (0)
            auto tile = [&](auto KT) __attribute__((always_inline)) {
                constexpr int kt = decltype(KT)::value;
                typedef typename a3_v2<T>::type v2;
                a3_i32x2 vt[2][4];                       // [16-key block][16-d block]
                f32x2 pp[NB][4];                         // a block's exponentials, until they are converted and summed one stage later
                unsigned pfw[NB][4];                     // P fragment of a block: four registers of two halves
                f32x2 lsa[NB], lsb[NB];
                float sum[NB];
                unsigned long long badm[NB];
                // E: exponentials of elements 2i, 2i + 1 of the lane's eight scores of block nb
                auto E = [&](int nb, int i) __attribute__((always_inline)) {
                    const int kb = i >> 1, r = 2 * (i & 1);
                    if (A3_KO & 2) { pp[nb][i] = (f32x2){Sq[nb][kb][r], Sq[nb][kb][r + 1]}; return; }
                    // The scores go through an empty asm ("modified" in place) in front of the exponentials: nothing orders a pure operation behind a
                    // scheduling fence, and as plain builtins the eight exponentials of a block were hoisted in front of the stage's first MFMA.  (Not
                    // v_exp_f32 in inline assembly: the compiler's hazard tables -- MFMA result read, and a vector write to a register an MFMA in
                    // flight still reads as its C operand -- do not look inside an asm statement.)
                    float x0 = Sq[nb][kb][r], x1 = Sq[nb][kb][r + 1];
                    asm volatile("" : "+v"(x0), "+v"(x1));
                    Sq[nb][kb][r] = x0; Sq[nb][kb][r + 1] = x1;
                    pp[nb][i] = (f32x2){__builtin_amdgcn_exp2f(x0), __builtin_amdgcn_exp2f(x1)};
                };
                auto cvt2 = [&](const f32x2 x) __attribute__((always_inline)) -> unsigned {
                    const v2 h = {(T)x[0], (T)x[1]};     // p in [0, 2^12): no saturation needed
                    unsigned u = __builtin_bit_cast(unsigned, h);
                    asm volatile("" : "+v"(u));          // (keeps the conversion here, between the MFMAs, instead of sunk behind the range check)
                    return u;
                };
                // X: the work on block nb's exponentials ONE STAGE after E, in four pieces that go between that stage's MFMAs: conversions and
                // half-row sums, then the range masks -- so that nothing a branch waits for was computed just in front of it
                auto X = [&](int nb, int i) __attribute__((always_inline)) {
                    if (i == 0) { pfw[nb][0] = cvt2(pp[nb][0]); pfw[nb][1] = cvt2(pp[nb][1]); lsa[nb] = pp[nb][0] + pp[nb][1]; }
                    if (i == 1) { pfw[nb][2] = cvt2(pp[nb][2]); pfw[nb][3] = cvt2(pp[nb][3]); lsb[nb] = pp[nb][2] + pp[nb][3]; }
                    if (i == 2) { lsa[nb] += lsb[nb]; sum[nb] = hsum2(lsa[nb]); }
                    if (i == 3) {
                        // rows with a reference need the upper bound only; rows without one keep 0 while the sum sits in the safe band (an all-masked
                        // tile has sum 0: it goes through redo, which leaves the row unset)
                        badm[nb] = __builtin_amdgcn_ballot_w64(!(sum[nb] < A3_SUM_HI)) | (__builtin_amdgcn_ballot_w64(!(sum[nb] >= A3_SUM_LO)) & ~refm[nb]);
                    }
                };
                // the rare path: a row's reference moves (or is set from a first tile outside the safe band); exponentials redone against it
                auto redo = [&](int nb) __attribute__((always_inline)) {
                    float mx = fmaxf(fmaxf(fmaxf(Sq[nb][0][0], Sq[nb][0][1]), fmaxf(Sq[nb][0][2], Sq[nb][0][3])),
                                     fmaxf(fmaxf(Sq[nb][1][0], Sq[nb][1][1]), fmaxf(Sq[nb][1][2], Sq[nb][1][3])));
                    mx = a3_rows_max(mx);
                    const bool rs = (refm[nb] >> lane) & 1ull;
                    // a row without a reference takes its tile maximum (unless every key so far is masked); one with a reference moves it past the threshold
                    const bool need = rs ? mx > A3_THR : mx != NEG_INF;
                    const float d = need ? mx : 0.f;
                    const float alpha = (need && rs) ? __builtin_amdgcn_exp2f(-d) : 1.f;      // (a row without a reference has accumulated nothing yet)
                    refm[nb] |= __builtin_amdgcn_ballot_w64(need);
                    gm[nb][1] += d;
                    l_run[nb] *= alpha;
#pragma unroll
                    for (int db = 0; db < 4; ++db) o[nb][db] *= alpha;
                    f32x2 ls = {0.f, 0.f};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int kb = i >> 1, r = 2 * (i & 1);
                        const f32x2 q2 = {__builtin_amdgcn_exp2f(Sq[nb][kb][r] - d), __builtin_amdgcn_exp2f(Sq[nb][kb][r + 1] - d)};
                        ls += q2;
                        const v2 h = {(T)q2[0], (T)q2[1]};
                        pfw[nb][i] = __builtin_bit_cast(unsigned, h);
                    }
                    sum[nb] = hsum2(ls);
                };
                auto check = [&](int nb) __attribute__((always_inline)) {
                    if (A3_KO & 1) { }
                    else if (__builtin_expect(badm[nb] != 0ull, 0)) redo(nb);
                    else refm[nb] = ~0ull;
                    l_run[nb] += sum[nb];
                };
                auto pfrag = [&](int nb) __attribute__((always_inline)) -> v8 {
                    typedef unsigned a3_u32x4 __attribute__((ext_vector_type(4)));
                    const a3_u32x4 w = {pfw[nb][0], pfw[nb][1], pfw[nb][2], pfw[nb][3]};
                    return __builtin_bit_cast(v8, w);
                };
                // the next key tile's accumulator start number j (0 .. 7): query block j >> 1, key block j & 1
                auto acc_next = [&](int j) __attribute__((always_inline)) {
                    if (kt + 1 >= NKT) return;
                    const int nn = j >> 1, kb = j & 1;
                    if ((A3_KO & 16) || !BIAS) { if (kb == 0) acc_plain(nn); }
                    else acc_start(nn, kb, run[(2 * (kt + 1) + kb - nn + 3) % 5]);
                };
#define A3_TS(i) if (A3_STAMPS == 2 && (kt == 6 || kt == 1)) { __builtin_amdgcn_sched_barrier(0); ts[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
                unsigned long long ts[12];
                A3_TS(0)
                // ---- stage 0: S of query block 0
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int s = i >> 1, kb = i & 1;
                    if (!(A3_KO & 4)) Sq[0][kb] = mfma16(kf[s][kb], qf[0][s], Sq[0][kb]);
                    A3_FENCE();
                }
                if (kt < 4 && do_dma) { dma_piece(ph + 1, kt); dma_done = kt + 1; A3_FENCE(); }
                {
                    // this tile's V fragments (transposed reads) and the next tile's bias runs go out AFTER every K fragment has met its first
                    // MFMA: the compiler counts only its own LDS reads, so a wait it places for a K fragment behind these would wait for them too
#define A3_TR(dst, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(vaddr), "n"(OFF))
                    A3_TR(vt[0][0], 4096 * kt + 0);    A3_TR(vt[1][0], 4096 * kt + 2048 + 0);
                    A3_TR(vt[0][1], 4096 * kt + 256);  A3_TR(vt[1][1], 4096 * kt + 2048 + 256);
                    A3_TR(vt[0][2], 4096 * kt + 512);  A3_TR(vt[1][2], 4096 * kt + 2048 + 512);
                    A3_TR(vt[0][3], 4096 * kt + 768);  A3_TR(vt[1][3], 4096 * kt + 2048 + 768);
#undef A3_TR
                    // the two new bias runs of the next key tile (w = 2 kt + 2, 2 kt + 3) replace the two oldest (dead since the last tile's starts)
                    if (BIAS && kt + 1 < NKT) {
                        run[(2 * kt + 5) % 5] = *(const f32x4*)(tp + 16 * (2 * kt + 2));
                        run[(2 * kt + 6) % 5] = *(const f32x4*)(tp + 16 * (2 * kt + 3));
                    }
                    A3_FENCE();
                }
                A3_TS(1)
                // ---- stages 1 .. 3: S of block n | exponentials of block n - 1 | conversions, sums, masks of block n - 2
#pragma unroll
                for (int nb = 1; nb < NB; ++nb) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int s = i >> 1, kb = i & 1;
                        if (!(A3_KO & 4)) Sq[nb][kb] = mfma16(kf[s][kb], qf[nb][s], Sq[nb][kb]);
                        A3_FENCE();
                        E(nb - 1, i);
                        if (nb >= 2) X(nb - 2, i);
                        A3_FENCE();
                    }
                    A3_TS(1 + nb)
                }
                check(0);
                A3_FENCE();
                // the V fragments (issued after stage 0); then the next tile's K fragment goes out behind them
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(vt[0][0]), "+v"(vt[0][1]), "+v"(vt[0][2]), "+v"(vt[0][3]), "+v"(vt[1][0]), "+v"(vt[1][1]), "+v"(vt[1][2]), "+v"(vt[1][3]));
                v8 vf[4];
#pragma unroll
                for (int db = 0; db < 4; ++db) {
                    const v4 lo = __builtin_bit_cast(v4, vt[0][db]), hi = __builtin_bit_cast(v4, vt[1][db]);
                    vf[db] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
                if (kt + 1 < NKT) {
#pragma unroll
                    for (int s = 0; s < 2; ++s)
#pragma unroll
                        for (int kb = 0; kb < 2; ++kb) kf[s][kb] = *(const v8*)(kp[s] + 4096 * (kt + 1) + 2048 * kb);
                }
                A3_FENCE();
                A3_TS(5)
                // ---- stage 4: P V of block 0 | exponentials of block 3 | conversions, sums, masks of block 2
                {
                    const v8 pf = pfrag(0);
#pragma unroll
                    for (int db = 0; db < 4; ++db) {
                        if (!(A3_KO & 8)) o[0][db] = mfma16(vf[db], pf, o[0][db]);
                        A3_FENCE();
                        E(3, db);
                        X(2, db);
                        A3_FENCE();
                    }
                }
                check(1);
                A3_FENCE();
                A3_TS(6)
                // ---- stage 5: P V of block 1 | conversions, sums, masks of block 3 | the next key tile's accumulator starts 0, 1
                {
                    const v8 pf = pfrag(1);
#pragma unroll
                    for (int db = 0; db < 4; ++db) {
                        if (!(A3_KO & 8)) o[1][db] = mfma16(vf[db], pf, o[1][db]);
                        A3_FENCE();
                        X(3, db);
                        if (db >= 2) acc_next(db - 2);
                        A3_FENCE();
                    }
                }
                check(2);
                A3_FENCE();
                A3_TS(7)
                // ---- stage 6: P V of block 2 | starts 2, 3, 4
                {
                    const v8 pf = pfrag(2);
#pragma unroll
                    for (int db = 0; db < 4; ++db) {
                        if (!(A3_KO & 8)) o[2][db] = mfma16(vf[db], pf, o[2][db]);
                        A3_FENCE();
                        if (db < 3) { acc_next(2 + db); A3_FENCE(); }
                    }
                }
                check(3);
                A3_FENCE();
                A3_TS(8)
                // ---- stage 7: P V of block 3 | starts 5, 6, 7
                {
                    const v8 pf = pfrag(3);
#pragma unroll
                    for (int db = 0; db < 4; ++db) {
                        if (!(A3_KO & 8)) o[3][db] = mfma16(vf[db], pf, o[3][db]);
                        A3_FENCE();
                        if (db < 3) { acc_next(5 + db); A3_FENCE(); }
                    }
                }
                A3_TS(9)
                if (kt + 1 < NKT && masked_at(kt + 1)) mask_acc(kt + 1);
#if A3_STAMPS == 2
                if ((kt == 6 || kt == 1) && blockIdx.x < 16 && lane == 0 && ph < 16) {
                    unsigned long long* d = g_a3_stamps + (((size_t)(16 + blockIdx.x) * 8 + wave) * 16 + ph) * 16;
                    if (kt == 6) d += 16 * 8 * 16 * 16;
                    for (int i = 0; i < 10; ++i) d[i] = ts[i];
                }
#endif
#undef A3_TS
            };
            A3_PT(3)
            if (0 < kt_end) tile(a3_ic<0>{});
            A3_PT(4)
            if (1 < kt_end) tile(a3_ic<1>{});
            A3_PT(5)
            if (2 < kt_end) tile(a3_ic<2>{});
            A3_PT(6)
            if (3 < kt_end) tile(a3_ic<3>{});
            A3_PT(7)
            if (4 < kt_end) tile(a3_ic<4>{});
            A3_PT(8)
            if (5 < kt_end) tile(a3_ic<5>{});
            A3_PT(9)
            if (6 < kt_end) tile(a3_ic<6>{});
            A3_PT(10)
            if (7 < kt_end) tile(a3_ic<7>{});
            if constexpr (XT) { if (8 < kt_end) tile(a3_ic<8>{}); }
#undef A3_FENCE
        }
        if (do_dma) {
            for (int u = dma_done; u < 4; ++u) dma_piece(ph + 1, u);      // short clips, waves without query rows
            dma_advance();
        }
        A3_PT(11)
        if (last_half) {
            // the item is complete: its Q fragment is dead, the next item's goes straight into the same registers; then normalise and store
            int hn = h_cur, bn = b_cur + 1;
            if (bn == Bc) { bn = 0; ++hn; }
            if (more_items) load_q(hn, bn);
            A3_PT(13)
            store_item(h_cur, b_cur);
            A3_PT(14)
            item_par ^= 1;
            h_cur = hn; b_cur = bn;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                gm[nb][1] = 0.f; l_run[nb] = 0.f; refm[nb] = 0ull;
#pragma unroll
                for (int db = 0; db < 4; ++db) o[nb][db] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
        half = last_half ? 0 : half + 1;
        A3_PT(12)
#undef A3_PT
#if A3_STAMPS
        if (blockIdx.x < 16 && lane == 0 && ph < 16) {
            unsigned long long* d = g_a3_stamps + (((size_t)blockIdx.x * 8 + wave) * 16 + ph) * 16;
            for (int i = 0; i < 16; ++i) d[i] = pt[i];
        }
#endif
    }
}

template <typename T>
int launch3(const void* qkv, int B, int Tn, int H, const float* bias_tab, const float* grep_w, const float* grep_b, const float* grep_a,
            const uint8_t* key_pad, void* out, int q_log2e, int n_wg, hipStream_t s) {
    const int n_units = B * H;
    const int per_block = (n_units + n_wg - 1) / n_wg;
    const int grid = (n_units + per_block - 1) / per_block;
    const int Tq = Tn < A3_TMAX ? Tn : A3_TMAX;
    if (Tn > A3_TMAX) {
        AVX_ENSURE_LDS((attention3_kernel<T, false, true>), ATT3X_LDS);
        hipLaunchKernelGGL((attention3_kernel<T, false, true>), dim3(grid), dim3(512), ATT3X_LDS, s, (const T*)qkv, Tn, Tq, H, B, per_block, bias_tab, grep_w,
                           grep_b, grep_a, key_pad, (T*)out, q_log2e);
    } else if (bias_tab) {
        AVX_ENSURE_LDS((attention3_kernel<T, true>), ATT3_LDS);
        hipLaunchKernelGGL((attention3_kernel<T, true>), dim3(grid), dim3(512), ATT3_LDS, s, (const T*)qkv, Tn, Tq, H, B, per_block, bias_tab, grep_w, grep_b,
                           grep_a, key_pad, (T*)out, q_log2e);
    } else {
        AVX_ENSURE_LDS((attention3_kernel<T, false>), ATT3_LDS);
        hipLaunchKernelGGL((attention3_kernel<T, false>), dim3(grid), dim3(512), ATT3_LDS, s, (const T*)qkv, Tn, Tq, H, B, per_block, bias_tab, grep_w, grep_b,
                           grep_a, key_pad, (T*)out, q_log2e);
    }
    AVX_LAUNCH_CHECK();
    return AVEXHIP_OK;
}

}  // namespace

namespace avx {

// Variant 3: called by avx::attention (attention.hip), which has validated the arguments.  T <= 512: the whole product.  512 < T <= 544 without a
// bias table (EAT's 513 tokens): query rows 0 .. 511 against all T keys (nine-tile last phase); the caller computes the rows beyond with the tail kernel.
int attention16(const void* qkv, int B, int T, int H, const float* bias_tab, const float* grep_w, const float* grep_b, const float* grep_a,
                const uint8_t* key_pad, void* out, int dtype, int q_log2e, int n_wg, hipStream_t s) {
    AVX_REQUIRE(T > 0 && (T <= A3_TMAX || (T <= A3_TMAX + 32 && !bias_tab)), "attention16: T=%d tokens (1..512, or up to 544 without a bias table)", T);
    AVX_REQUIRE((int64_t)B * H < (1ll << 31), "attention16: too many (clip, head) items");
    if (dtype == AVEXHIP_F16) return launch3<_Float16>(qkv, B, T, H, bias_tab, grep_w, grep_b, grep_a, key_pad, out, q_log2e, n_wg, s);
    if (dtype == AVEXHIP_BF16) return launch3<__bf16>(qkv, B, T, H, bias_tab, grep_w, grep_b, grep_a, key_pad, out, q_log2e, n_wg, s);
    avexhip_set_error("attention16: unknown dtype %d", dtype);
    return AVEXHIP_ERR_INVALID;
}

}  // namespace avx
