// Gated relative-position-bias self-attention for BEATs (head_dim 64, any T) on gfx950.
//
// Restates _MultiheadAttention.forward (avex/models/beats/backbone.py:494-574) after the q/k/v
// projections:  softmax( q k^T / 8 + gate(b,h,i) * bias[h, j-i]  [+ -inf on padded keys] ) v
// with gate = ga * (gb * grep_a[h] - 1) + 2, (ga, gb) = sigmoid(grep_linear(q).view(2,4).sum(-1))
// (backbone.py:543-551).  The reference materialises a [B,H,T,T] fp32 mask (3 GB at B=256); here the
// bias is a per-head Toeplitz row of 2T-1 floats in LDS and nothing of size T^2 ever exists.
//
// One 1024-thread workgroup (16 waves, one 32-query tile each at T = 496) per (clip, head).  The head's whole K (T x 64, row-major, XOR-swizzled
// 16-byte chunks) and V^T (64 x T, rows padded to 1032 B) live in LDS (129 KiB of the CU's 160), so
// K/V are read from HBM exactly once.  Each wave owns 32 queries at a time and keeps the QUERY on the
// MFMA lane for both products (v_mfma_f32_32x32x16):
//     S^T[key][query] = K[key][:] . Q[query][:]      (A = K rows from LDS, B = Q from registers)
//     O^T[d][query]  += V^T[d][key] * P^T[key][query] (A = V^T rows from LDS, B = P^T straight
//                                                     from the S^T accumulator registers)
// so the online-softmax state (running max / sum, rescale factor) is per-lane scalar state, the row
// reductions are 15 in-register ops + one cross-half shuffle, and P never touches LDS.  The k-order
// inside a PV k-step is the accumulator's register order: element j of lane-half h is key
// 16 s + 8 (j>>2) + 4 h + (j&3); the V^T fragment is gathered with exactly that map (two 8-byte reads).
#include <stdlib.h>

#include "common.h"

namespace {

// The AVEX_AMD_ATT_DEBUG experiment knobs (skip tiles, skip the DMA, non-temporal stores) exist only in the diagnostic build
// (-DAVEX_DIAG); in the product library `dbg` is the constant 0 and the branches on it are compiled out.
#ifdef AVEX_DIAG
#define AVX_ATT_DBG(arg) const int dbg = (arg);
#else
#define AVX_ATT_DBG(arg) constexpr int dbg = 0; (void)(arg);
#endif

constexpr int TMAX = 512;
constexpr int VT_LD = 516;                       // halves per V^T row (1032 B: conflict-free b64 reads)
constexpr int KS_BYTES = TMAX * 128;             // 65536
constexpr int VT_BYTES = 64 * VT_LD * 2;         // 66048
constexpr int TAB_LD = 1040;                     // floats per shifted copy of the bias row
constexpr int TAB_BYTES = 4 * TAB_LD * 4;        // 4 copies, copy s holds tab[k + s]: every lane reads 16-byte aligned
constexpr int KADD_BYTES = TMAX * 4;             // 2048
constexpr int GW_BYTES = 136 * 4;                // wa[64] wb[64] ba bb (+pad)
constexpr int ATT_LDS = KS_BYTES + VT_BYTES + TAB_BYTES + KADD_BYTES + GW_BYTES;

template <typename T>
__global__ __launch_bounds__(1024) void attention_kernel(const T* __restrict__ qkv, int Tn, int H,
                                                        const float* __restrict__ bias_tab,
                                                        const float* __restrict__ grep_w,
                                                        const float* __restrict__ grep_b,
                                                        const float* __restrict__ grep_a,
                                                        const uint8_t* __restrict__ key_pad,
                                                        T* __restrict__ out, int q_log2e, int dbg_arg) {
    AVX_ATT_DBG(dbg_arg)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename Half<T>::v8 v8;
    typedef typename Half<T>::v4 v4;
    char* Ks = smem;
    T* Vt = (T*)(smem + KS_BYTES);
    float* tab = (float*)(smem + KS_BYTES + VT_BYTES);
    float* kadd = (float*)(smem + KS_BYTES + VT_BYTES + TAB_BYTES);
    float* gw = (float*)(smem + KS_BYTES + VT_BYTES + TAB_BYTES + KADD_BYTES);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.x % H, b = blockIdx.x / H;
    const int E = H * 64;
    const int64_t ld = 3 * (int64_t)E;
    const T* base = qkv + (int64_t)b * Tn * ld + h * 64;
    const float NEG_INF = -__builtin_inff();

    // ---- stage K (swizzled rows), V^T, bias row, key mask, gate weights -----------------------
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int idx = tid + 1024 * it;
        const int row = idx >> 3, c = idx & 7;
        uint4 kv = make_uint4(0, 0, 0, 0);
        v8 vv;
#pragma unroll
        for (int e = 0; e < 8; ++e) vv[e] = (T)0.0f;
        if (row < Tn) {
            kv = *(const uint4*)(base + (int64_t)row * ld + E + c * 8);
            vv = *(const v8*)(base + (int64_t)row * ld + 2 * E + c * 8);
        }
        *(uint4*)(Ks + row * 128 + ((c ^ ((row >> 1) & 7)) << 4)) = kv;
#pragma unroll
        for (int e = 0; e < 8; ++e) Vt[(c * 8 + e) * VT_LD + row] = vv[e];
    }
    for (int r = tid; r < TAB_LD; r += 1024) {
        // bias row pre-multiplied by log2(e): the softmax runs in base 2 (v_exp_f32 is 2^x).
        // copy s stores tab[k + s] at k, so a lane whose first index is tb reads copy (tb & 3) at tb - (tb & 3).
        float v = 0.f;
        if (bias_tab && r < 2 * Tn - 1) v = bias_tab[(int64_t)h * (2 * Tn - 1) + r] * 1.4426950408889634f;
#pragma unroll
        for (int sft = 0; sft < 4; ++sft)
            if (r - sft >= 0) tab[sft * TAB_LD + (r - sft)] = v;
    }
    if (tid < TMAX) {
        const int j = tid;
        bool ok = j < Tn;
        if (ok && key_pad) ok = key_pad[(int64_t)b * Tn + j] == 0;
        kadd[j] = ok ? 0.f : NEG_INF;
    }
    if (tid < 64) {
        float a = 0.f, bb = 0.f;
        if (grep_w) {
            a = (grep_w[0 * 64 + tid] + grep_w[1 * 64 + tid]) + (grep_w[2 * 64 + tid] + grep_w[3 * 64 + tid]);
            bb = (grep_w[4 * 64 + tid] + grep_w[5 * 64 + tid]) + (grep_w[6 * 64 + tid] + grep_w[7 * 64 + tid]);
        }
        // q_log2e: the caller's Q columns already carry log2(e) (folded into W_q / b_q in fp32 when the weights were packed, so that Q is
        // rounded to the operand type once and no rounded constant enters the scores); the gate wants the plain q: its weights take 1 / log2(e)
        const float ginv = q_log2e ? 0.6931471805599453f : 1.0f;
        gw[tid] = a * ginv;
        gw[64 + tid] = bb * ginv;
        if (tid == 0) {
            gw[128] = grep_w ? (grep_b[0] + grep_b[1]) + (grep_b[2] + grep_b[3]) : 0.f;
            gw[129] = grep_w ? (grep_b[4] + grep_b[5]) + (grep_b[6] + grep_b[7]) : 0.f;
        }
    }
    __syncthreads();
    const float sscale = q_log2e ? 0.125f : 0.125f * 1.4426950408889634f;

    const int nqt = (Tn + 31) >> 5;
    const int hh = lane >> 5, r32 = lane & 31;
    const float head_a = grep_w ? grep_a[h] : 0.f;

    for (int qt = wave; qt < nqt; qt += 16) {
        const int i = qt * 32 + r32;
        const int iq = i < Tn ? i : Tn - 1;
        v8 qf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = *(const v8*)(base + (int64_t)iq * ld + 16 * s + 8 * hh);

        float gate = 1.f;
        if (grep_w) {
            float pa = 0.f, pb = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float qv = (float)qf[s][j];
                    pa += gw[16 * s + 8 * hh + j] * qv;
                    pb += gw[64 + 16 * s + 8 * hh + j] * qv;
                }
            pa += __shfl_xor(pa, 32, 64);
            pb += __shfl_xor(pb, 32, 64);
            const float ga = 1.f / (1.f + __expf(-(pa + gw[128])));
            const float gb = 1.f / (1.f + __expf(-(pb + gw[129])));
            gate = ga * (gb * head_a - 1.f) + 2.f;
        }

        float m_run = NEG_INF, l_run = 0.f;
        f32x16 o0, o1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }

        const int nkt_run = dbg == 1 ? 0 : (dbg == 2 ? 1 : nqt);
        for (int kt = 0; kt < nkt_run; ++kt) {
            f32x16 S;
            const int krow = kt * 32 + r32;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int chunk = hh + 2 * s;
                const v8 kf = *(const v8*)(Ks + krow * 128 + ((chunk ^ ((krow >> 1) & 7)) << 4));
                if (s == 0) {
                    f32x16 z;
#pragma unroll
                    for (int r = 0; r < 16; ++r) z[r] = 0.f;
                    S = mfma32(kf, qf[0], z);
                } else {
                    S = mfma32(kf, qf[s], S);
                }
            }
            const int jb = kt * 32 + 4 * hh;
            const int tb = jb - iq + (Tn - 1);
            const float* tp = tab + (tb & 3) * TAB_LD + (tb & ~3);
            float tv[16];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4 t4 = *(const f32x4*)(tp + 8 * g4);
                tv[4 * g4 + 0] = t4[0]; tv[4 * g4 + 1] = t4[1]; tv[4 * g4 + 2] = t4[2]; tv[4 * g4 + 3] = t4[3];
            }
            float sc[16];
            float mx = NEG_INF;
            // scores in log2 units: s * (log2e / 8) + gate * (bias * log2e)
            const bool masked_tile = key_pad != nullptr || (kt * 32 + 32 > Tn);   // wave-uniform
            if (masked_tile) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int jo = (r & 3) + 8 * (r >> 2);
                    sc[r] = __builtin_fmaf(S[r], sscale, gate * tv[r]) + kadd[jb + jo];
                    mx = fmaxf(mx, sc[r]);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int jo = (r & 3) + 8 * (r >> 2);
                    sc[r] = __builtin_fmaf(S[r], sscale, gate * tv[r]);
                    mx = fmaxf(mx, sc[r]);
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx);
            const float m_use = m_new == NEG_INF ? 0.f : m_new;
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
            float ls = 0.f;
            float p[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                p[r] = __builtin_amdgcn_exp2f(sc[r] - m_use);
                ls += p[r];
            }
            l_run = __builtin_fmaf(l_run, alpha, ls);
            m_run = m_new;
            if (__any(alpha != 1.f)) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
            }
            v8 pf[2];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[s2][j] = (T)p[8 * s2 + j];   // p in [0, 1]: no saturation needed
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int key0 = kt * 32 + 16 * s2 + 4 * hh;
                {
                    const T* vr = Vt + r32 * VT_LD + key0;
                    const v4 lo = *(const v4*)vr, hi = *(const v4*)(vr + 8);
                    v8 vf;
                    vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
                    vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
                    o0 = mfma32(vf, pf[s2], o0);
                }
                {
                    const T* vr = Vt + (32 + r32) * VT_LD + key0;
                    const v4 lo = *(const v4*)vr, hi = *(const v4*)(vr + 8);
                    v8 vf;
                    vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
                    vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
                    o1 = mfma32(vf, pf[s2], o1);
                }
            }
        }
        const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
        const float inv = 1.f / l_tot;
        if (i < Tn) {
            T* orow = out + ((int64_t)b * Tn + i) * E + h * 64;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                v4 a, c;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    a[e] = Half<T>::from(o0[4 * g + e] * inv);
                    c[e] = Half<T>::from(o1[4 * g + e] * inv);
                }
                *(v4*)(orow + 8 * g + 4 * hh) = a;
                *(v4*)(orow + 32 + 8 * g + 4 * hh) = c;
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------
// Variant 2: persistent, K/V streamed by LDS-DMA under the compute.
//
// Variant 1 above stages a whole head (K, V^T, bias row) and only then computes: at B = 256 its staging is
// HBM-bound (15 of 42 us per workgroup) and nothing overlaps it.  Here one workgroup per CU walks a contiguous
// range of (head, clip) items; the keys of an item are split in two halves of 256 and the two 64 KiB LDS
// buffers are filled by LDS-DMA one phase ahead (item i half 1 while half 0 is computed, item i+1 half 0 while
// half 1 is computed), so the HBM stream runs continuously beside the MFMA/softmax work.  V stays row-major
// in 8-key x 32-column subtiles and is consumed with ds_read_b64_tr_b16 (the hardware transpose read), which
// removes the scalar transposing stores of variant 1.  The next item's Q fragment is prefetched into registers
// during the last phase of the current one, and an item's output is stored after the next phase's DMA has been
// issued.  The softmax uses a deferred running maximum (rescale only when a row's maximum grows by more than 2^8):
// the S accumulators START at gate * bias - reference (+ key mask) and the query fragment carries log2(e) / 8, so the S chain delivers
// the exponent itself: a score costs half a packed FMA (the start), one v_exp_f32, half a packed add and half a convert.
// A key tile is issued as four stages, each one query tile's MFMA chain with the other tile's vector work between its instructions
// (ATT_IL, below): an MFMA waiting for the matrix pipe blocks the SIMD's vector issue for the OTHER wave, not for its own.
// Instantiations: <T, LONG> (more than 512 tokens: query blocks of 512, bias windows per phase) x <BIAS> (no table: EAT, wav2vec2).
// ---------------------------------------------------------------------------------------------
typedef __attribute__((address_space(1))) const void a_gptr_t;
typedef __attribute__((address_space(3))) void a_lptr_t;
typedef short a_s4v __attribute__((__vector_size__(4 * sizeof(short))));
typedef int a_i32x2 __attribute__((ext_vector_type(2)));
template <int N> struct a_ic { static constexpr int value = N; };

constexpr int A2_KBUF = 32768;                            // 256 keys x 128 B, 16-byte chunks XOR-swizzled
constexpr int A2_HALF = 65536;                            // + 256 keys of V as 32 subtile rows of 1 KiB
constexpr int A2_TAB_OFF = 2 * A2_HALF;
constexpr int A2_KADD_OFF = A2_TAB_OFF + TAB_BYTES;
constexpr int A2_GW_OFF = A2_KADD_OFF + 2 * KADD_BYTES;
constexpr int ATT2_LDS = A2_GW_OFF + GW_BYTES;
// Long clips (more than 512 tokens): queries in blocks of 512, keys in blocks of 256; a phase is one (query block, key block)
// pair and carries its own WINDOW of the bias row (the 767 offsets j - i that pair can meet, four shifted copies) and its own
// 256-entry key mask, both double-buffered and written one phase ahead.  Nothing in LDS depends on T.
constexpr int A2_WLD = 776;                               // floats per shifted copy of a window
constexpr int A2_WIN_BYTES = 4 * A2_WLD * 4;              // 12 416
constexpr int A2L_KADD_OFF = A2_TAB_OFF + 2 * A2_WIN_BYTES;
constexpr int A2L_GW_OFF = A2L_KADD_OFF + 2 * 256 * 4;
constexpr int ATT2L_LDS = A2L_GW_OFF + GW_BYTES;
// ... and for the nine-tile last phase (no bias table): 288 keys per buffer, no windows
constexpr int A2X_KBUF = 9 * 4096;                        // 36 864
constexpr int A2X_HALF = 2 * A2X_KBUF;                    // 73 728
constexpr int A2X_KADD_OFF = 2 * A2X_HALF;
constexpr int A2X_GW_OFF = A2X_KADD_OFF + 2 * 288 * 4;
constexpr int ATT2X_LDS = A2X_GW_OFF + GW_BYTES;
constexpr float A2_THR = 8.f;
#if defined(ATT_STAMPS) && ATT_STAMPS
__device__ unsigned long long g_att_stamps[64 * 8 * 32 * 8];   // [block < 64][wave][phase < 32][7 x s_memtime, s_memrealtime]
#endif
#ifndef ATT_STAGGER
#define ATT_STAGGER 0    // waves 4-7 enter each phase's key tiles this many 64-cycle sleeps after waves 0-3 (SIMD partners out of lockstep)
#endif
#ifndef ATT_PRIO
#define ATT_PRIO 1       // 1: waves 4-7 run at s_setprio 1
#endif
#ifndef ATT_IL
#define ATT_IL 1         // 1: the key tile's MFMAs and vector work interleaved instruction by instruction in every wave (below); 0: in segments
#endif
#ifndef ATT_STAMPS
#define ATT_STAMPS 0     // diagnostic build: -DATT_STAMPS=1 prints one tile's cycle stamps (AVEX_AMD_ATT_DEBUG=4)
#endif                             // deferred-max threshold, log2 units (p <= 256)

// One LDS-DMA wave instruction (64 lanes x 16 B -> 1 KiB of LDS at a wave-uniform base), written as inline assembly on
// purpose: behind a __builtin_amdgcn_global_load_lds the compiler puts an s_waitcnt vmcnt(0) in front of every later LDS
// read it cannot prove disjoint (here: all of them), which stalls the first key tile of each phase until the whole next
// buffer has landed and undoes the overlap.  The kernel orders DMA and reads itself (explicit vmcnt + barrier per phase).
static __device__ __forceinline__ void a2_dma16(const void* src, const char* lds_dst) {
    const unsigned lds = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) const char*)lds_dst;
    asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds) : "memory");
}

#ifndef ATT_BIAS_REUSE
#define ATT_BIAS_REUSE 1      // 0: A/B build, each query tile reads its bias vectors from LDS (round 2's form; tiles 256 rows apart)
#endif
// XT (long clips without a bias table only): a LAST key block of 257 .. 288 keys runs as one phase of NINE key tiles instead of a ninth
// tile's worth of keys getting a phase of their own (EAT: 513 keys = 256 + 257).  The buffers grow to 288 keys; the LDS the bias windows
// would take is free without a table.
template <typename T, bool LONG, bool BIAS, bool XT = false>      // BIAS false: no relative-position table (EAT, wav2vec2): the S accumulators start at -m alone
__global__ __launch_bounds__(512) void attention2_kernel(const T* __restrict__ qkv, int Tn, int H, int Bc, int per_block, int nqb_main,
                                                        const float* __restrict__ bias_tab,
                                                        const float* __restrict__ grep_w,
                                                        const float* __restrict__ grep_b,
                                                        const float* __restrict__ grep_a,
                                                        const uint8_t* __restrict__ key_pad,
                                                        T* __restrict__ out, int q_log2e, int dbg_arg) {
    AVX_ATT_DBG(dbg_arg)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename Half<T>::v8 v8;
    typedef typename Half<T>::v4 v4;
    constexpr int NQ = 2, NW = 8, NT = 512;              // 8 waves, two 32-query tiles each (tile wave + 8 u)
    static_assert(!XT || (LONG && !BIAS && ATT_IL), "the nine-tile last phase is built for long clips without a bias table (whose windows' LDS it takes)");
    constexpr int NKT = XT ? 9 : 8;                      // key tiles a phase buffer holds
    constexpr int KBUF = XT ? A2X_KBUF : A2_KBUF;        // bytes of K per buffer (V follows)
    constexpr int HALF = XT ? A2X_HALF : A2_HALF;        // bytes per buffer
    constexpr int KSLOT = XT ? 288 : 256;                // key-mask entries per phase (LONG)
    float* tab = (float*)(smem + A2_TAB_OFF);
    float* kadd = (float*)(smem + (XT ? A2X_KADD_OFF : (LONG ? A2L_KADD_OFF : A2_KADD_OFF)));
    float* gw = (float*)(smem + (XT ? A2X_GW_OFF : (LONG ? A2L_GW_OFF : A2_GW_OFF)));

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // The unit of work is (item, query block of 512): a workgroup takes per_block consecutive units, so a few long clips still fill the
    // chip (1 clip x 12 heads x 4 query blocks = 48 workgroups instead of 12); short clips have one query block per item.
    const int nqb = LONG ? nqb_main : 1;                             // query blocks per item (a short last block may be left to the tail kernel)
    const int n_units = Bc * H * nqb;
    const int w0 = blockIdx.x * per_block;
    const int w1 = w0 + per_block < n_units ? w0 + per_block : n_units;
    if (w0 >= w1) return;
    const int it0 = LONG ? w0 / nqb : w0;
    const int qb0 = LONG ? w0 - it0 * nqb : 0;
    const int E = H * 64;
    const int64_t ld = 3 * (int64_t)E;
    const float NEG_INF = -__builtin_inff();
    const int nh = XT ? Tn >> 8 : (LONG ? (Tn + 255) >> 8 : (Tn > 256 ? 2 : 1));      // key blocks of 256 per query block (XT: the last one takes the 1 .. 32 keys beyond)
    const int np = (w1 - w0) * nh;
    const int nkt = (Tn + 31) >> 5;
    const int hh = lane >> 5, r32 = lane & 31;
    int qi[NQ], iq[NQ];
    bool has_q = false;                                  // wave-uniform: tile u = 0 of the current query block exists
    auto set_qblock = [&](int qb) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            qi[u] = qb * 512 + (ATT_BIAS_REUSE ? (NQ * wave + u) : (wave + NW * u)) * 32 + r32;   // ADJACENT tiles: tile 1's bias values at key tile k are tile 0's at key tile k - 1 (below)
            iq[u] = qi[u] < Tn ? qi[u] : Tn - 1;         // clamped for loads; stores are masked
        }
        has_q = qb * 512 + (ATT_BIAS_REUSE ? NQ * wave : wave) * 32 < Tn;
    };
    set_qblock(qb0);

    // Items are h * Bc + b, walked in order: (h, b) of the item being computed and of the one being loaded are kept
    // incrementally (no division in the loop).  A workgroup's range stays on one head, or two at a seam.
    int h_cur = it0 / Bc, b_cur = it0 - h_cur * Bc;      // item of the phase being computed
    int qb_cur = qb0, half = 0;                          // its query block and key block
    int h_ld = h_cur, b_ld = b_cur, half_ld = 0, qb_ld = qb0;   // (item, query block, key block) the next DMA fetches
    int item_par = 0;                                    // parity of the item being computed: its kadd slot

    auto issue_next = [&](int ph) __attribute__((always_inline)) {      // DMA for phase ph = (h_ld, b_ld, half_ld), then advance
        const T* base = qkv + (int64_t)b_ld * Tn * ld + h_ld * 64;
        char* buf = smem + (ph & 1) * HALF;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int ri = 4 * wave + u;                                   // 8-key group inside the half
            const int kl = 8 * ri + (lane >> 3);
            int key = half_ld * 256 + kl;
            key = key < Tn ? key : Tn - 1;                                 // clamped rows are masked by kadd
            const int chunk = (lane & 7) ^ ((kl >> 1) & 7);
            a2_dma16(base + (int64_t)key * ld + E + chunk * 8, buf + ri * 1024);
            int vkey = half_ld * 256 + 8 * ri + ((lane >> 2) & 7);
            vkey = vkey < Tn ? vkey : Tn - 1;
            const int ch = 4 * (lane >> 5) + (lane & 3);
            a2_dma16(base + (int64_t)vkey * ld + 2 * E + ch * 8, buf + KBUF + ri * 1024);
        }
        if (XT && half_ld == nh - 1 && wave < 4) {                         // the ninth key tile: four more 8-key groups, one per wave 0 .. 3
            // (the counted waits of the phase loop count operations YOUNGER than the DMA -- Q loads, output stores -- so a wave may issue more)
            const int ri = 32 + wave;
            const int kl = 8 * ri + (lane >> 3);
            int key = half_ld * 256 + kl;
            key = key < Tn ? key : Tn - 1;
            const int chunk = (lane & 7) ^ ((kl >> 1) & 7);
            a2_dma16(base + (int64_t)key * ld + E + chunk * 8, buf + ri * 1024);
            int vkey = half_ld * 256 + 8 * ri + ((lane >> 2) & 7);
            vkey = vkey < Tn ? vkey : Tn - 1;
            const int ch = 4 * (lane >> 5) + (lane & 3);
            a2_dma16(base + (int64_t)vkey * ld + 2 * E + ch * 8, buf + KBUF + ri * 1024);
        }
        if (++half_ld == nh) { half_ld = 0; if (++qb_ld == nqb) { qb_ld = 0; if (++b_ld == Bc) { b_ld = 0; ++h_ld; } } }
    };
    // LONG: bias window and key mask of phase ph = the load-side state (call BEFORE issue_next(ph) advances it)
    auto write_window = [&](int ph) __attribute__((always_inline)) {
        float* win = tab + (ph & 1) * (4 * A2_WLD);
        const int r0 = half_ld * 256 - qb_ld * 512 - 511 + (Tn - 1);     // bias-row index of window entry 0
        for (int r = tid; r < A2_WLD && (BIAS || !ATT_IL); r += NT) {   // (the interleaved tile body does not read the window without a table)
            const int g = r0 + r;
            float v = 0.f;
            if (bias_tab && g >= 0 && g < 2 * Tn - 1) v = bias_tab[(int64_t)h_ld * (2 * Tn - 1) + g] * 1.4426950408889634f;
#pragma unroll
            for (int sft = 0; sft < 4; ++sft)
                if (r - sft >= 0) win[sft * A2_WLD + (r - sft)] = v;
        }
        if (tid < KSLOT) {
            const int j = half_ld * 256 + tid;
            bool ok = j < Tn;
            if (ok && key_pad) ok = key_pad[(int64_t)b_ld * Tn + j] == 0;
            kadd[(ph & 1) * KSLOT + tid] = ok ? 0.f : NEG_INF;
        }
    };
    auto write_kadd = [&](int slot, int b) __attribute__((always_inline)) {
        for (int j = tid; j < TMAX; j += NT) {
            bool ok = j < Tn;
            if (ok && key_pad) ok = key_pad[(int64_t)b * Tn + j] == 0;
            kadd[slot * TMAX + j] = ok ? 0.f : NEG_INF;
        }
    };

    v8 qf[NQ][4];
    auto load_q = [&](int h, int b) __attribute__((always_inline)) {
        const T* base = qkv + (int64_t)b * Tn * ld + h * 64;
#pragma unroll
        for (int u = 0; u < NQ; ++u)
#pragma unroll
            for (int s = 0; s < 4; ++s) qf[u][s] = *(const v8*)(base + (int64_t)iq[u] * ld + 16 * s + 8 * hh);
    };
    if (ATT_PRIO && wave >= NW / 2) __builtin_amdgcn_s_setprio(1);   // the second-dispatched half loses every VALU arbitration otherwise (one static raise, no per-segment flips)
    load_q(h_cur, b_cur);                                // older than the DMA below: waiting for it never waits for the DMA
    if (LONG) write_window(0);
    issue_next(0);
    if (!LONG) write_kadd(0, b_cur);
    if (tid < 64) {
        float a = 0.f, bb = 0.f;
        if (grep_w) {
            a = (grep_w[0 * 64 + tid] + grep_w[1 * 64 + tid]) + (grep_w[2 * 64 + tid] + grep_w[3 * 64 + tid]);
            bb = (grep_w[4 * 64 + tid] + grep_w[5 * 64 + tid]) + (grep_w[6 * 64 + tid] + grep_w[7 * 64 + tid]);
        }
        const float ginv = q_log2e ? 0.6931471805599453f : 1.0f;      // (see attention_kernel: the gate reads the unscaled q)
        gw[tid] = a * ginv;
        gw[64 + tid] = bb * ginv;
        if (tid == 0) {
            gw[128] = grep_w ? (grep_b[0] + grep_b[1]) + (grep_b[2] + grep_b[3]) : 0.f;
            gw[129] = grep_w ? (grep_b[4] + grep_b[5]) + (grep_b[6] + grep_b[7]) : 0.f;
        }
    }

    int h_tab = -1;
    float gate[NQ], m_run[NQ], l_run[NQ];
    bool ref_set[NQ];
    f32x16 o0[NQ], o1[NQ];
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
        gate[u] = 1.f; m_run[u] = 0.f; l_run[u] = 0.f; ref_set[u] = false;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[u][r] = 0.f; o1[u][r] = 0.f; }
    }
    const int li = lane & 15;
    const int v_lane = 64 * (4 * hh + (li >> 2)) + 32 * ((lane >> 4) & 1) + 8 * (li & 3);   // transposed-read address, per lane
    // 1/8 is exact in the operand type; log2(e) is not (bf16: 0.18 % off, a temperature error on every logit): handles fold it into W_q in fp32
    const float cs = q_log2e ? 0.125f : 0.125f * 1.4426950408889634f;

    // Output of item (h, b): the lane pair (l, l + 32) holds a query row's d = 8g + 4hh + 0..3; two v_permlane32_swap per
    // pair of g give each lane 8 consecutive d, so a row is written in 16-byte pieces (four stores per 32-query tile).
    int n_st = 0;                                        // store instructions this wave issued for the last finished item (wave-uniform)
    auto store_item = [&](int h, int b) __attribute__((always_inline)) {
        n_st = 0;
        if (!has_q || (dbg & 32)) return;
#pragma unroll
        for (int u = 0; u < NQ; ++u) n_st += __builtin_amdgcn_readfirstlane(qi[u] - r32) < Tn ? 4 : 0;
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            const float l_tot = l_run[u] + __shfl_xor(l_run[u], 32, 64);
            const float inv = __builtin_amdgcn_rcpf(l_tot);       // (1 ulp; the quotient is rounded to 11 or 8 bits two lines below)
            T* orow = out + ((int64_t)b * Tn + qi[u]) * E + h * 64 + 8 * hh;
#pragma unroll
            for (int oh = 0; oh < 2; ++oh) {
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    v4 x, y;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        // plain conversions: an output row is a convex combination of V rows, it cannot leave the operand type's range
                        x[e] = (T)((oh ? o1[u] : o0[u])[8 * gp + e] * inv);
                        y[e] = (T)((oh ? o1[u] : o0[u])[8 * gp + 4 + e] * inv);
                    }
                    a_i32x2 xi = __builtin_bit_cast(a_i32x2, x), yi = __builtin_bit_cast(a_i32x2, y);
                    int x0 = xi[0], x1 = xi[1], y0 = yi[0], y1 = yi[1];
                    // swap x's upper-half lanes with y's lower-half lanes: low lanes end with d 16gp + 0..7, high lanes with 8..15
                    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3\n\ts_nop 1" : "+v"(x0), "+v"(x1), "+v"(y0), "+v"(y1));
                    typedef int i32x4_t __attribute__((ext_vector_type(4)));
                    const i32x4_t w = {x0, x1, y0, y1};
                    if (qi[u] < Tn) {
                        if (dbg == 8) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(orow + 32 * oh + 16 * gp), "v"(w) : "memory");   // A/B: AVEX_AMD_ATT_DEBUG=8
                        else *(i32x4_t*)(orow + 32 * oh + 16 * gp) = w;
                    }
                }
            }
        }
    };

    for (int ph = 0; ph < np; ++ph) {
        const bool last_half = half == nh - 1;
        const bool more_items = ph + (nh - half) < np;   // another (item, query block) follows the one being computed
        // phase boundary: this wave's DMA for phase ph (issued one phase ago) has landed and every wave has finished
        // reading the other buffer.  When the previous phase ended an item, the 8 loads of the next item's Q are younger
        // than that DMA and stay in flight.
#define AVX_PT(i) if (ATT_STAMPS) { __builtin_amdgcn_sched_barrier(0); pt[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
        unsigned long long pt[8];
        if (ATT_STAMPS) pt[7] = __builtin_amdgcn_s_memrealtime();
        AVX_PT(0)
        if (half == 0 && ph > 0) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        AVX_PT(1)
        __builtin_amdgcn_s_barrier();
        AVX_PT(2)
        AVX_PT(3)
        bool vm_young = LONG && (BIAS || key_pad != nullptr);      // a global load younger than the previous item's output stores is pending (long clips: the next window's bias row / key mask)
        if (half == 0) {
            if (!LONG && h_cur != h_tab) {               // workgroup-uniform
                vm_young = true;
                for (int r = tid; r < TAB_LD; r += NT) {
                    float v = 0.f;
                    if (bias_tab && r < 2 * Tn - 1) v = bias_tab[(int64_t)h_cur * (2 * Tn - 1) + r] * 1.4426950408889634f;
#pragma unroll
                    for (int sft = 0; sft < 4; ++sft)
                        if (r - sft >= 0) tab[sft * TAB_LD + (r - sft)] = v;
                }
                h_tab = h_cur;
                __syncthreads();
            }
            if (!LONG && more_items) {                   // the next item's key mask, read two barriers from now
                int bn = b_cur + 1;
                bn = bn == Bc ? 0 : bn;
                write_kadd(item_par ^ 1, bn);
                vm_young = vm_young || key_pad != nullptr;
            }
#pragma unroll
            for (int u = 0; u < NQ; ++u) {
                gate[u] = 1.f;
                if (grep_w && !(dbg & 16)) {
                    f32x2 pa = {0.f, 0.f}, pb = {0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
#pragma unroll
                        for (int j4 = 0; j4 < 2; ++j4) {
                            const f32x4 wa = *(const f32x4*)(gw + 16 * s + 8 * hh + 4 * j4);
                            const f32x4 wb = *(const f32x4*)(gw + 64 + 16 * s + 8 * hh + 4 * j4);
                            const f32x2 q01 = {(float)qf[u][s][4 * j4], (float)qf[u][s][4 * j4 + 1]};
                            const f32x2 q23 = {(float)qf[u][s][4 * j4 + 2], (float)qf[u][s][4 * j4 + 3]};
                            pa = __builtin_elementwise_fma((f32x2){wa[0], wa[1]}, q01, pa);
                            pa = __builtin_elementwise_fma((f32x2){wa[2], wa[3]}, q23, pa);
                            pb = __builtin_elementwise_fma((f32x2){wb[0], wb[1]}, q01, pb);
                            pb = __builtin_elementwise_fma((f32x2){wb[2], wb[3]}, q23, pb);
                        }
                    }
                    float sa = hsum2(pa), sb = hsum2(pb);
                    sa += __shfl_xor(sa, 32, 64);
                    sb += __shfl_xor(sb, 32, 64);
                    const float ga = 1.f / (1.f + __expf(-(sa + gw[128])));
                    const float gb = 1.f / (1.f + __expf(-(sb + gw[129])));
                    gate[u] = ga * (gb * grep_a[h_cur] - 1.f) + 2.f;
                }
            }
        }
        // Every global load of the setup above (bias row, key mask) and the item's Q fragment (8 loads issued at the end of the
        // previous item) complete HERE, before the DMA goes out: after it a compiler-placed vmcnt wait would wait for the DMA
        // as well.  Passing qf through the statement makes this the definition the tiles see, in every phase, so no wait
        // for those loads is placed inside the tiles.  From here to the next boundary a phase touches only LDS and registers.
        if (LONG && ph + 1 < np) write_window(ph + 1);   // next phase's bias window + key mask (other slot; read after the next barrier)
        // At an item's first phase the youngest outstanding operations are the previous item's output stores (n_st of them, issued
        // after the Q loads): the counted wait lets them drain under the tiles instead of stalling every wave for their round trip
        // to HBM (3.4k of an item's 37k cycles).  Anything younger (mask / bias-row loads above) forces the full wait.
        if (half == 0 && ph > 0 && !vm_young && n_st == 8)
            asm volatile("s_waitcnt vmcnt(8)"
                         : "+v"(qf[0][0]), "+v"(qf[0][1]), "+v"(qf[0][2]), "+v"(qf[0][3]), "+v"(qf[1][0]), "+v"(qf[1][1]), "+v"(qf[1][2]), "+v"(qf[1][3]));
        else if (half == 0 && ph > 0 && !vm_young && n_st == 4)
            asm volatile("s_waitcnt vmcnt(4)"
                         : "+v"(qf[0][0]), "+v"(qf[0][1]), "+v"(qf[0][2]), "+v"(qf[0][3]), "+v"(qf[1][0]), "+v"(qf[1][1]), "+v"(qf[1][2]), "+v"(qf[1][3]));
        else
            asm volatile("s_waitcnt vmcnt(0)"
                         : "+v"(qf[0][0]), "+v"(qf[0][1]), "+v"(qf[0][2]), "+v"(qf[0][3]), "+v"(qf[1][0]), "+v"(qf[1][1]), "+v"(qf[1][2]), "+v"(qf[1][3]));
        if (half == 0) {
            // this (item, query block)'s fragment arrives unscaled (the gate above wants it so); from here on it carries the score scale
            const T qs = (T)cs;
#pragma unroll
            for (int u = 0; u < NQ; ++u)
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                    for (int e8 = 0; e8 < 8; ++e8) qf[u][s4][e8] = qf[u][s4][e8] * qs;
        }
        if (ph + 1 < np && dbg != 3) issue_next(ph + 1);
        AVX_PT(4)
        if (has_q) {
            const char* Kb = smem + (ph & 1) * HALF;
            const float* kad = LONG ? kadd + (ph & 1) * KSLOT - half * 256 : kadd + item_par * TMAX;   // indexed by the global key
            const int kt_lim = (XT && last_half) ? 9 : 8;
            int kt_end = nkt - half * 8 < kt_lim ? nkt - half * 8 : kt_lim;
            if (dbg == 1) kt_end = 0;
            if (dbg == 2) kt_end = 1;
            // per-phase address registers; everything inside the 8 unrolled key tiles is base + immediate
            const int swz = (r32 >> 1) & 7;
            const char* kp[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) kp[s] = Kb + r32 * 128 + (((hh + 2 * s) ^ swz) << 4);
            const float* tp[NQ];
#pragma unroll
            for (int u = 0; u < NQ; ++u) {
                if (LONG) {
                    const int tb = 4 * hh + 511 - (iq[u] - qb_cur * 512);        // window coordinates, 0 .. 515
                    tp[u] = tab + (ph & 1) * (4 * A2_WLD) + (tb & 3) * A2_WLD + (tb & ~3);
                } else {
                    const int tb = half * 256 + 4 * hh - iq[u] + (Tn - 1);
                    tp[u] = tab + (tb & 3) * TAB_LD + (tb & ~3);
                }
            }
            const unsigned vaddr = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) const char*)(Kb + KBUF + v_lane);

            v8 kf[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) kf[s] = *(const v8*)(kp[s]);
            if (ATT_STAGGER > 0 && wave >= NW / 2) __builtin_amdgcn_s_sleep(ATT_STAGGER);
#if ATT_IL
            // A wave issues in order, and an MFMA that finds the matrix pipe busy holds the SIMD's vector issue port until it is
            // accepted (scripts/micro/seg_cost.hip: a vector wave beside a wave of back-to-back MFMAs runs at an eighth of its rate).
            // So the overlap of matrix and vector work is made INSIDE each wave: every stage below is one MFMA chain of one query
            // tile with the other tile's vector work placed between its instructions, and the scheduler is fenced so that it stays
            // there.  Stage 1: S chain of tile 0.  Stage 2: S chain of tile 1 | exponentials of tile 0.  Stage 3: P V of tile 0 |
            // exponentials of tile 1.  Stage 4: P V of tile 1 | the next key tile's accumulator start (gate * bias - m) of both.
            f32x16 Sq[NQ];
            auto masked_at = [&](int ktl) __attribute__((always_inline)) { return key_pad != nullptr || (half * 8 + ktl) * 32 + 32 > Tn; };
            auto mask_acc = [&](int u, int ktl) __attribute__((always_inline)) {
                const int jb = (half * 8 + ktl) * 32 + 4 * hh;
#pragma unroll
                for (int r = 0; r < 16; ++r) Sq[u][r] += kad[jb + (r & 3) + 8 * (r >> 2)];
            };
            auto acc_start = [&](int u, int g4, const f32x4 t4) __attribute__((always_inline)) {
                const f32x2 g2 = {gate[u], gate[u]}, nm2 = {-m_run[u], -m_run[u]};
                const f32x2 ea = __builtin_elementwise_fma(g2, (f32x2){t4[0], t4[1]}, nm2);
                const f32x2 eb = __builtin_elementwise_fma(g2, (f32x2){t4[2], t4[3]}, nm2);
                Sq[u][4 * g4] = ea[0]; Sq[u][4 * g4 + 1] = ea[1]; Sq[u][4 * g4 + 2] = eb[0]; Sq[u][4 * g4 + 3] = eb[1];
            };
            // without a bias table (EAT, wav2vec2) the start is -m alone: no table reads, no FMAs (workgroup-uniform branch)
            constexpr bool has_bias = BIAS || !ATT_IL;
            auto acc_plain = [&](int u) __attribute__((always_inline)) {
#pragma unroll
                for (int r = 0; r < 16; ++r) Sq[u][r] = -m_run[u];
            };
            // The bias is Toeplitz (it depends on key - query only) and a wave's two query tiles are 32 rows apart, so the 16 values tile 1
            // needs at key tile k are the ones tile 0 used at key tile k - 1: they stay in registers (tcur) instead of coming from LDS a
            // second time.  The kernel is LDS-bandwidth-bound (scripts/micro/att_lds16.hip: 2 449 cycles per key tile with its LDS reads,
            // 1 498 without), and the bias vectors were half of a wave's 16 KB per key tile; now 12 KB.
            f32x4 tcur[4];
            if (has_bias) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) { tcur[g4] = *(const f32x4*)(tp[0] + 8 * g4); acc_start(0, g4, tcur[g4]); }
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) acc_start(1, g4, *(const f32x4*)(tp[1] + 8 * g4));
            } else { acc_plain(0); acc_plain(1); }
            if (masked_at(0)) { mask_acc(0, 0); mask_acc(1, 0); }
#define AVX_FENCE() __builtin_amdgcn_sched_barrier(0)
            auto tile = [&](auto KT) __attribute__((always_inline)) {
                constexpr int ktl = decltype(KT)::value;
                unsigned long long ts[8];
#define AVX_TS(i) if (ATT_STAMPS && ktl == 3 && ph == 6) { __builtin_amdgcn_sched_barrier(0); ts[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
                AVX_TS(0)
                a_i32x2 vt[2][2][2];
#define AVX_TR(dst, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(vaddr), "n"(OFF))
                AVX_TR(vt[0][0][0], 1024 * (ktl * 4 + 0) + 0);   AVX_TR(vt[0][0][1], 1024 * (ktl * 4 + 1) + 0);
                AVX_TR(vt[0][1][0], 1024 * (ktl * 4 + 0) + 512); AVX_TR(vt[0][1][1], 1024 * (ktl * 4 + 1) + 512);
                AVX_TR(vt[1][0][0], 1024 * (ktl * 4 + 2) + 0);   AVX_TR(vt[1][0][1], 1024 * (ktl * 4 + 3) + 0);
                AVX_TR(vt[1][1][0], 1024 * (ktl * 4 + 2) + 512); AVX_TR(vt[1][1][1], 1024 * (ktl * 4 + 3) + 512);
#undef AVX_TR
                v8 pf[NQ][2];
                f32x2 ls[NQ];
                auto exp8 = [&](int u, int r0) __attribute__((always_inline)) {
#pragma unroll
                    for (int r = r0; r < r0 + 8; r += 2) {
                        const f32x2 pp = {__builtin_amdgcn_exp2f(Sq[u][r]), __builtin_amdgcn_exp2f(Sq[u][r + 1])};
                        ls[u] += pp;
                        pf[u][r >> 3][r & 7] = (T)pp[0];
                        pf[u][r >> 3][(r & 7) + 1] = (T)pp[1];
                    }
                };
                // (see the segment form below for the deferred reference: a half-row sum below 2^12 proves the tile's p are f16-safe)
                auto redo_u = [&](int u) __attribute__((always_inline)) {
                    float mx = Sq[u][0];
#pragma unroll
                    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, Sq[u][r]);
                    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                    const bool need = ref_set[u] ? mx > A2_THR : mx != NEG_INF;
                    const float d = need ? mx : 0.f;
                    const float alpha = (need && ref_set[u]) ? __builtin_amdgcn_exp2f(-d) : 1.f;
                    ref_set[u] = ref_set[u] || need;
                    m_run[u] += d;
                    l_run[u] *= alpha;
                    ls[u] = (f32x2){0.f, 0.f};
#pragma unroll
                    for (int r = 0; r < 16; ++r) { o0[u][r] *= alpha; o1[u][r] *= alpha; }
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const f32x2 pp = {__builtin_amdgcn_exp2f(Sq[u][r] - d), __builtin_amdgcn_exp2f(Sq[u][r + 1] - d)};
                        ls[u] += pp;
                        pf[u][r >> 3][r & 7] = (T)pp[0];
                        pf[u][r >> 3][(r & 7) + 1] = (T)pp[1];
                    }
                };
                // stage 1
#pragma unroll
                for (int s = 0; s < 4; ++s) Sq[0] = mfma32(kf[s], qf[0][s], Sq[0]);
                AVX_FENCE();
                AVX_TS(1)
                // stage 2
                Sq[1] = mfma32(kf[0], qf[1][0], Sq[1]);
                Sq[1] = mfma32(kf[1], qf[1][1], Sq[1]);
                AVX_FENCE();
                ls[0] = (f32x2){0.f, 0.f};
                exp8(0, 0);
                AVX_FENCE();
                Sq[1] = mfma32(kf[2], qf[1][2], Sq[1]);
                AVX_FENCE();
                exp8(0, 8);
                AVX_FENCE();
                Sq[1] = mfma32(kf[3], qf[1][3], Sq[1]);
                AVX_FENCE();
                AVX_TS(2)
                if (__any(!ref_set[0] || !(hsum2(ls[0]) < 4096.f))) redo_u(0);
                l_run[0] += hsum2(ls[0]);
                // the V fragments (issued at the top), then the next tile's K fragment and bias values go out behind them
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(vt[0][0][0]), "+v"(vt[0][0][1]), "+v"(vt[0][1][0]), "+v"(vt[0][1][1]),
                               "+v"(vt[1][0][0]), "+v"(vt[1][0][1]), "+v"(vt[1][1][0]), "+v"(vt[1][1][1]));
                v8 vf[2][2];
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int dh = 0; dh < 2; ++dh) {
                        const v4 lo = __builtin_bit_cast(v4, vt[s2][dh][0]), hi = __builtin_bit_cast(v4, vt[s2][dh][1]);
                        vf[s2][dh] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    }
                f32x4 t4n[4];
                if (ktl + 1 < NKT) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) kf[s] = *(const v8*)(kp[s] + (ktl + 1) * 4096);
                    if (has_bias) {
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) t4n[g4] = *(const f32x4*)(tp[0] + (ktl + 1) * 32 + 8 * g4);
                    }
                }
                AVX_FENCE();
                AVX_TS(3)
                // stage 3
                o0[0] = mfma32(vf[0][0], pf[0][0], o0[0]);
                o1[0] = mfma32(vf[0][1], pf[0][0], o1[0]);
                AVX_FENCE();
                ls[1] = (f32x2){0.f, 0.f};
                exp8(1, 0);
                AVX_FENCE();
                o0[0] = mfma32(vf[1][0], pf[0][1], o0[0]);
                AVX_FENCE();
                exp8(1, 8);
                AVX_FENCE();
                o1[0] = mfma32(vf[1][1], pf[0][1], o1[0]);
                AVX_FENCE();
                AVX_TS(4)
                if (__any(!ref_set[1] || !(hsum2(ls[1]) < 4096.f))) redo_u(1);
                l_run[1] += hsum2(ls[1]);
                AVX_FENCE();
                AVX_TS(5)
                // stage 4
                if (has_bias) {
                    o0[1] = mfma32(vf[0][0], pf[1][0], o0[1]);
                    if (ktl + 1 < NKT) { acc_start(0, 0, t4n[0]); acc_start(0, 1, t4n[1]); }
                    AVX_FENCE();
                    o1[1] = mfma32(vf[0][1], pf[1][0], o1[1]);
                    if (ktl + 1 < NKT) { acc_start(0, 2, t4n[2]); acc_start(0, 3, t4n[3]); }
                    AVX_FENCE();
                    o0[1] = mfma32(vf[1][0], pf[1][1], o0[1]);
                    if (!ATT_BIAS_REUSE && ktl + 1 < NKT) {
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) tcur[g4] = *(const f32x4*)(tp[1] + (ktl + 1) * 32 + 8 * g4);      // A/B build: tile 1's values from LDS as before
                    }
                    if (ktl + 1 < NKT) { acc_start(1, 0, tcur[0]); acc_start(1, 1, tcur[1]); }       // tile 1 at key tile k + 1 = tile 0's values at key tile k
                    AVX_FENCE();
                    o1[1] = mfma32(vf[1][1], pf[1][1], o1[1]);
                    if (ktl + 1 < NKT) {
                        acc_start(1, 2, tcur[2]); acc_start(1, 3, tcur[3]);
                        if (ATT_BIAS_REUSE) {
#pragma unroll
                            for (int g4 = 0; g4 < 4; ++g4) tcur[g4] = t4n[g4];
                        }
                    }
                    AVX_FENCE();
                } else {
                    o0[1] = mfma32(vf[0][0], pf[1][0], o0[1]);
                    o1[1] = mfma32(vf[0][1], pf[1][0], o1[1]);
                    if (ktl + 1 < NKT) acc_plain(0);
                    AVX_FENCE();
                    o0[1] = mfma32(vf[1][0], pf[1][1], o0[1]);
                    o1[1] = mfma32(vf[1][1], pf[1][1], o1[1]);
                    if (ktl + 1 < NKT) acc_plain(1);
                    AVX_FENCE();
                }
                AVX_TS(6)
                if (ktl + 1 < NKT && masked_at(ktl + 1)) { mask_acc(0, ktl + 1); mask_acc(1, ktl + 1); }
#if ATT_STAMPS
                if (ktl == 3 && ph == 6 && blockIdx.x < 64 && lane == 0) {
                    unsigned long long* d = g_att_stamps + (((size_t)blockIdx.x * 8 + wave) * 32 + 31) * 8;
                    for (int i = 0; i < 7; ++i) d[i] = ts[i];
                }
#endif
#undef AVX_TS
            };
#undef AVX_FENCE
#else
            auto tile = [&](auto KT) __attribute__((always_inline)) {
                constexpr int ktl = decltype(KT)::value;
                const int kt = half * 8 + ktl;
                const int jb = kt * 32 + 4 * hh;
                const bool masked_tile = key_pad != nullptr || kt * 32 + 32 > Tn;   // wave-uniform
                // V fragments of this key tile, transposed by the LDS: issued now, waited for after the softmax
                a_i32x2 vt[2][2][2];
#define AVX_TR(dst, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(vaddr), "n"(OFF))
                AVX_TR(vt[0][0][0], 1024 * (ktl * 4 + 0) + 0);   AVX_TR(vt[0][0][1], 1024 * (ktl * 4 + 1) + 0);
                AVX_TR(vt[0][1][0], 1024 * (ktl * 4 + 0) + 512); AVX_TR(vt[0][1][1], 1024 * (ktl * 4 + 1) + 512);
                AVX_TR(vt[1][0][0], 1024 * (ktl * 4 + 2) + 0);   AVX_TR(vt[1][0][1], 1024 * (ktl * 4 + 3) + 0);
                AVX_TR(vt[1][1][0], 1024 * (ktl * 4 + 2) + 512); AVX_TR(vt[1][1][1], 1024 * (ktl * 4 + 3) + 512);
#undef AVX_TR
                // Both query tiles in one straight-line block.  The MFMA accumulators START at the additive part of the score,
                // gate * bias - m_run (+ the key mask), in log2 units, and the query fragment is pre-multiplied by log2(e) / 8 (once per
                // item, below): the S chain then delivers  score - m_run  ready for the exponential -- no VALU between the matrix
                // pipe and v_exp_f32, and the bias FMAs sit in front of the chain where they overlap the partner wave's MFMAs.
                // The running reference is checked AFTER the exponentials, on the half-row sums that are needed anyway: a sum below
                // 2^12 proves every p of this lane is below 2^12 (f16-safe), so the common path has no maximum at all.
                v8 pf[NQ][2];
                f32x2 ls[NQ];
                f32x16 Sq[NQ];
                auto init_acc = [&](int u) __attribute__((always_inline)) {
                    const f32x2 g2 = {gate[u], gate[u]}, nm2 = {-m_run[u], -m_run[u]};
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const f32x4 t4 = (dbg & 512) ? (f32x4){0.f, 0.f, 0.f, 0.f} : *(const f32x4*)(tp[u] + ktl * 32 + 8 * g4);
                        const f32x2 ea = __builtin_elementwise_fma(g2, (f32x2){t4[0], t4[1]}, nm2);
                        const f32x2 eb = __builtin_elementwise_fma(g2, (f32x2){t4[2], t4[3]}, nm2);
                        Sq[u][4 * g4] = ea[0]; Sq[u][4 * g4 + 1] = ea[1]; Sq[u][4 * g4 + 2] = eb[0]; Sq[u][4 * g4 + 3] = eb[1];
                    }
                    if (masked_tile) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) Sq[u][r] += kad[jb + (r & 3) + 8 * (r >> 2)];
                    }
                };
                // exponentials of tile u -> P fragment + half-row sums; true if the running reference has to move
                auto softmax_u = [&](int u) __attribute__((always_inline)) -> bool {
                    ls[u] = (f32x2){0.f, 0.f};
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const f32x2 pp = (dbg & 128) ? (f32x2){Sq[u][r], Sq[u][r + 1]} : (f32x2){__builtin_amdgcn_exp2f(Sq[u][r]), __builtin_amdgcn_exp2f(Sq[u][r + 1])};
                        ls[u] += pp;
                        pf[u][r >> 3][r & 7] = (T)pp[0];
                        pf[u][r >> 3][(r & 7) + 1] = (T)pp[1];
                    }
                    // !(sum < 2^12) also catches the overflowed (inf) and the invalid (NaN) sum
                    return !ref_set[u] || !(hsum2(ls[u]) < 4096.f);
                };
                // a row's first unmasked tile sets its reference to the row maximum; later it moves when a tile has grown past it.
                // The exponentials of this tile are redone against the new reference.
                auto redo_u = [&](int u) __attribute__((always_inline)) {
                    float mx = Sq[u][0];
#pragma unroll
                    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, Sq[u][r]);
                    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                    const bool need = ref_set[u] ? mx > A2_THR : mx != NEG_INF;
                    const float d = need ? mx : 0.f;
                    const float alpha = (need && ref_set[u]) ? __builtin_amdgcn_exp2f(-d) : 1.f;
                    ref_set[u] = ref_set[u] || need;
                    m_run[u] += d;
                    l_run[u] *= alpha;
                    ls[u] = (f32x2){0.f, 0.f};
#pragma unroll
                    for (int r = 0; r < 16; ++r) { o0[u][r] *= alpha; o1[u][r] *= alpha; }
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const f32x2 pp = {__builtin_amdgcn_exp2f(Sq[u][r] - d), __builtin_amdgcn_exp2f(Sq[u][r + 1] - d)};
                        ls[u] += pp;
                        pf[u][r >> 3][r & 7] = (T)pp[0];
                        pf[u][r >> 3][(r & 7) + 1] = (T)pp[1];
                    }
                };
                v8 vf[2][2];
                auto wait_v = [&]() __attribute__((always_inline)) {
                    asm volatile("s_waitcnt lgkmcnt(0)"
                                 : "+v"(vt[0][0][0]), "+v"(vt[0][0][1]), "+v"(vt[0][1][0]), "+v"(vt[0][1][1]),
                                   "+v"(vt[1][0][0]), "+v"(vt[1][0][1]), "+v"(vt[1][1][0]), "+v"(vt[1][1][1]));
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                        for (int dh = 0; dh < 2; ++dh) {
                            const v4 lo = __builtin_bit_cast(v4, vt[s2][dh][0]), hi = __builtin_bit_cast(v4, vt[s2][dh][1]);
                            vf[s2][dh] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                        }
                };
                auto next_k = [&]() __attribute__((always_inline)) {
                    // the K fragment is dead once both chains are issued: the next tile's goes into the same registers now
                    if (ktl + 1 < NKT) {
#pragma unroll
                        for (int s = 0; s < 4; ++s) kf[s] = *(const v8*)(kp[s] + (ktl + 1) * 4096);
                    }
                };
                init_acc(0); init_acc(1);
#pragma unroll
                for (int s = 0; s < 4; ++s) {            // the two tiles' chains interleaved: consecutive MFMAs never share an accumulator
#pragma unroll
                    for (int u = 0; u < NQ; ++u) if (!(dbg & 64)) Sq[u] = mfma32(kf[s], qf[u][s], Sq[u]);
                }
                next_k();
                bool moves = softmax_u(0);
                moves = softmax_u(1) || moves;
                if (__any(moves)) { redo_u(0); redo_u(1); }
                l_run[0] += hsum2(ls[0]); l_run[1] += hsum2(ls[1]);
                wait_v();
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int dh = 0; dh < 2; ++dh)
#pragma unroll
                        for (int u = 0; u < NQ; ++u) {
                            if (dbg & 256) continue;
                            if (dh == 0) o0[u] = mfma32(vf[s2][0], pf[u][s2], o0[u]);
                            else o1[u] = mfma32(vf[s2][1], pf[u][s2], o1[u]);
                        }
            };
#endif
            if (0 < kt_end) tile(a_ic<0>{});
            if (1 < kt_end) tile(a_ic<1>{});
            if (2 < kt_end) tile(a_ic<2>{});
            if (3 < kt_end) tile(a_ic<3>{});
            if (4 < kt_end) tile(a_ic<4>{});
            if (5 < kt_end) tile(a_ic<5>{});
            if (6 < kt_end) tile(a_ic<6>{});
            if (7 < kt_end) tile(a_ic<7>{});
            if constexpr (XT) { if (8 < kt_end) tile(a_ic<8>{}); }
        }
        AVX_PT(5)
        if (last_half) {
            // the item is complete: its Q fragment is dead, so the next item's goes straight into the same registers (these 8
            // loads are what the vmcnt(8) at the next boundary leaves in flight); then normalise and store
            int hn = h_cur, bn = b_cur, qn = qb_cur + 1;
            if (qn == nqb) { qn = 0; if (++bn == Bc) { bn = 0; ++hn; } }
            if (LONG) {
                // the next block's Q loads go out BEFORE this block's stores (as below), from row indices computed aside: the stores are
                // then the youngest operations and drain under the next tiles instead of in front of the boundary's counted wait
                if (more_items) {
                    const T* nbase = qkv + (int64_t)bn * Tn * ld + hn * 64;
#pragma unroll
                    for (int u = 0; u < NQ; ++u) {
                        int qin = qn * 512 + (ATT_BIAS_REUSE ? (NQ * wave + u) : (wave + NW * u)) * 32 + r32;
                        qin = qin < Tn ? qin : Tn - 1;
#pragma unroll
                        for (int s4 = 0; s4 < 4; ++s4) qf[u][s4] = *(const v8*)(nbase + (int64_t)qin * ld + 16 * s4 + 8 * hh);
                    }
                }
                store_item(h_cur, b_cur);                // its row indices are this block's: before set_qblock
                set_qblock(qn);
            } else {
                if (more_items) load_q(hn, bn);
                store_item(h_cur, b_cur);
            }
            if (qn == 0) item_par ^= 1;
            h_cur = hn; b_cur = bn; qb_cur = qn;
            // (stores are younger than the Q loads: the boundary's vmcnt(8) would have to be vmcnt(16) to skip them; it waits
            // for them instead, which also keeps the count independent of has_q)
#pragma unroll
            for (int u = 0; u < NQ; ++u) {
                m_run[u] = 0.f; l_run[u] = 0.f; ref_set[u] = false;
#pragma unroll
                for (int r = 0; r < 16; ++r) { o0[u][r] = 0.f; o1[u][r] = 0.f; }
            }
        }
        half = last_half ? 0 : half + 1;
        AVX_PT(6)
#undef AVX_PT
#if ATT_STAMPS
        if (blockIdx.x < 64 && lane == 0 && ph < 32) {
            unsigned long long* d = g_att_stamps + (((size_t)blockIdx.x * 8 + wave) * 32 + ph) * 8;
            for (int i = 0; i < 8; ++i) d[i] = pt[i];
        }
#endif
    }
}

// The last few query rows of a long clip (T = 512 n + r; in practice EAT's class-token row, 513 = 512 + 1): a further query block of
// the streamed kernel for them costs more than this.  One workgroup of four waves per (clip, head, group of RW rows): scores over all
// keys with the threads on the keys (every K row is read once for the RW rows), softmax through LDS, then each wave takes a quarter of
// the keys with its lanes on the 64 output dimensions (every V row read once for the RW rows) and the four partial outputs are added in
// wave order.  Same arithmetic as the main kernel (base-2 softmax, gate * bias, key mask), fp32 throughout except the operands and the
// stored output.  (One wave per row, round 2's form, was a chain of ~140 dependent memory round trips: 0.1 ms at 3 072 rows.)
template <typename T, int RW>
__global__ __launch_bounds__(256) void attention_tail_kernel(const T* __restrict__ qkv, int Tn, int H, int T0, int R, const float* __restrict__ bias_tab,
                                                             const float* __restrict__ grep_w, const float* __restrict__ grep_b, const float* __restrict__ grep_a,
                                                             const uint8_t* __restrict__ key_pad, T* __restrict__ out, int q_log2e) {
    extern __shared__ float sc[];                            // [RW][Tn] scores, then probabilities; [RW][64] q; [RW] gates; [4][RW][64] partials
    float* qs = sc + RW * Tn;
    float* gates = qs + RW * 64;
    float* part = gates + RW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = (R + RW - 1) / RW;
    const int grp = blockIdx.x % G, h = (blockIdx.x / G) % H, b = blockIdx.x / (G * H);
    const int E = H * 64;
    const int64_t ld = 3 * (int64_t)E;
    const T* base = qkv + (int64_t)b * Tn * ld + h * 64;
    const float NEG_INF = -__builtin_inff();
    auto row_of = [&](int r) { const int i = T0 + grp * RW + r; return i < T0 + R ? i : T0 + R - 1; };   // rows past the end repeat the last one (computed, not stored)
    // wave w loads q rows w, w + 4, ... and makes their gates
    for (int r = wave; r < RW; r += 4) {
        const float qv = (float)base[(int64_t)row_of(r) * ld + lane];
        qs[r * 64 + lane] = qv;
        float gate = 1.f;
        if (grep_w) {
            const float ginv = q_log2e ? 0.6931471805599453f : 1.0f;
            const float wa = ((grep_w[0 * 64 + lane] + grep_w[1 * 64 + lane]) + (grep_w[2 * 64 + lane] + grep_w[3 * 64 + lane])) * ginv;
            const float wb = ((grep_w[4 * 64 + lane] + grep_w[5 * 64 + lane]) + (grep_w[6 * 64 + lane] + grep_w[7 * 64 + lane])) * ginv;
            const float sa = wave_sum(qv * wa) + ((grep_b[0] + grep_b[1]) + (grep_b[2] + grep_b[3]));
            const float sb = wave_sum(qv * wb) + ((grep_b[4] + grep_b[5]) + (grep_b[6] + grep_b[7]));
            const float ga = 1.f / (1.f + __expf(-sa)), gb = 1.f / (1.f + __expf(-sb));
            gate = ga * (gb * grep_a[h] - 1.f) + 2.f;
        }
        if (lane == 0) gates[r] = gate;
    }
    __syncthreads();
    const float cs = q_log2e ? 0.125f : 0.125f * 1.4426950408889634f;
    // scores: the thread's K row is converted once (64 registers) and meets the RW query rows one after the other; q is read from LDS as
    // broadcast 16-byte vectors (kept there on purpose: in registers it would be 64 RW values per lane)
#pragma unroll 1
    for (int j = tid; j < Tn; j += 256) {
        typedef typename Half<T>::v8 v8;
        const T* krow = base + (int64_t)j * ld + E;
        float kf[64];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const v8 kv = *(const v8*)(krow + 8 * c);
#pragma unroll
            for (int e = 0; e < 8; ++e) kf[8 * c + e] = (float)kv[e];
        }
        const bool padded = key_pad && key_pad[(int64_t)b * Tn + j];
#pragma unroll 1
        for (int r = 0; r < RW; ++r) {
            const f32x4* q4 = (const f32x4*)(qs + r * 64);
            float d = 0.f;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const f32x4 q = q4[c];
                d = __builtin_fmaf(kf[4 * c], q[0], d); d = __builtin_fmaf(kf[4 * c + 1], q[1], d);
                d = __builtin_fmaf(kf[4 * c + 2], q[2], d); d = __builtin_fmaf(kf[4 * c + 3], q[3], d);
            }
            float sv = d * cs;
            if (bias_tab) sv = __builtin_fmaf(gates[r], bias_tab[(int64_t)h * (2 * Tn - 1) + (j - row_of(r)) + (Tn - 1)] * 1.4426950408889634f, sv);
            if (padded) sv = NEG_INF;
            sc[r * Tn + j] = sv;
        }
    }
    __syncthreads();
    // softmax: row maxima and sums over the four waves through `part`
    float l[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        float mxr = NEG_INF;
        for (int j = tid; j < Tn; j += 256) mxr = fmaxf(mxr, sc[r * Tn + j]);
        mxr = wave_max(mxr);
        if (lane == 0) part[r * 4 + wave] = mxr;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        float mxr = fmaxf(fmaxf(part[r * 4], part[r * 4 + 1]), fmaxf(part[r * 4 + 2], part[r * 4 + 3]));
        if (mxr == NEG_INF) mxr = 0.f;                       // every key masked: all probabilities 0 (the division below gives NaN like the reference)
        float lr = 0.f;
        for (int j = tid; j < Tn; j += 256) {
            const float pj = __builtin_amdgcn_exp2f(sc[r * Tn + j] - mxr);
            sc[r * Tn + j] = (float)(T)pj;                   // the numerator uses P rounded to the operand type, the row sum does not -- as in the main kernel
            lr += pj;
        }
        lr = wave_sum(lr);
        if (lane == 0) part[RW * 4 + r * 4 + wave] = lr;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RW; ++r) l[r] = (part[RW * 4 + r * 4] + part[RW * 4 + r * 4 + 1]) + (part[RW * 4 + r * 4 + 2] + part[RW * 4 + r * 4 + 3]);
    __syncthreads();                                         // `part` is reused for the partial outputs
    // P V: wave w takes keys [w Tq, (w + 1) Tq).  A lane owns 8 consecutive output columns (16-byte V loads) of every eighth key of the wave's range:
    // one wave instruction fetches eight whole V rows (1 KiB), eight of them in flight per lane.  (Round 5: with the lanes on the 64 output columns a
    // wave instruction fetched ONE row -- 2 bytes per lane, 128 wave loads per wave behind each other at 16 in flight: 16 of the kernel's ~30 us per
    // workgroup, and the tail kernel is a fifth of EAT's attention.)  The eight key-subsets of a wave are added with lane exchanges, the four waves
    // through `part` as before.
    const int Tq = (Tn + 3) >> 2;
    const int j0 = wave * Tq, j1 = (j0 + Tq) < Tn ? (j0 + Tq) : Tn;
    const int ks = lane >> 3, dc = lane & 7;
    typedef typename Half<T>::v8 v8t;
    const T* vbase = base + 2 * E + 8 * dc;
    float o[RW][8];
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int e = 0; e < 8; ++e) o[r][e] = 0.f;
    for (int jj = j0; jj < j1; jj += 64) {                   // eight 8-key groups per trip
        v8t vv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            int key = jj + 8 * u + ks;
            key = key < j1 ? key : j1 - 1;                   // (clamped: its probability is taken as 0 below)
            vv[u] = *(const v8t*)(vbase + (int64_t)key * ld);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int key = jj + 8 * u + ks;
            const bool on = key < j1;
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                const float pj = on ? sc[r * Tn + key] : 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[r][e] = __builtin_fmaf(pj, (float)vv[u][e], o[r][e]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float x = o[r][e];
            x += __shfl_xor(x, 8, 64); x += __shfl_xor(x, 16, 64); x += __shfl_xor(x, 32, 64);
            o[r][e] = x;
        }
    if (ks == 0) {
#pragma unroll
        for (int r = 0; r < RW; ++r)
#pragma unroll
            for (int e = 0; e < 8; ++e) part[(wave * RW + r) * 64 + 8 * dc + e] = o[r][e];      // (`part` is only 4-byte aligned: its offset depends on Tn)
    }
    __syncthreads();
    for (int r = wave; r < RW; r += 4) {
        if (T0 + grp * RW + r < T0 + R) {
            const float acc = (part[(0 * RW + r) * 64 + lane] + part[(1 * RW + r) * 64 + lane]) + (part[(2 * RW + r) * 64 + lane] + part[(3 * RW + r) * 64 + lane]);
            out[((int64_t)b * Tn + row_of(r)) * E + h * 64 + lane] = Half<T>::from(acc / l[r]);
        }
    }
}

template <typename T>
int launch(const void* qkv, int B, int Tn, int H, const float* bias_tab, const float* grep_w, const float* grep_b,
           const float* grep_a, const uint8_t* key_pad, void* out, int q_log2e, hipStream_t s) {
    static const int dbg = getenv("AVEX_AMD_ATT_DEBUG") ? atoi(getenv("AVEX_AMD_ATT_DEBUG")) : 0;
    // 1 = stage-then-compute, 2 = persistent streamed (32x32x16 MFMAs), 3 = persistent streamed on 16x16x32 MFMAs (attention16.hip; default up to 512 tokens)
    int variant = getenv("AVEX_AMD_ATT_VARIANT") ? atoi(getenv("AVEX_AMD_ATT_VARIANT")) : 3;
    if (Tn > TMAX) variant = 2;          // variants 1 and 3 are built for T <= 512
    if (variant == 3) {
        int n_cu = 256;
        { const int rc_ = avx::device_cu_count(&n_cu); if (rc_ != AVEXHIP_OK) return rc_; }
        int n_wg = n_cu;
        if (const char* fg = getenv("AVEX_AMD_ATT_GRID")) { const int g = atoi(fg); if (g > 0) n_wg = g; }
        return avx::attention16(qkv, B, Tn, H, bias_tab, grep_w, grep_b, grep_a, key_pad, out, __is_same(T, _Float16) ? AVEXHIP_F16 : AVEXHIP_BF16, q_log2e, n_wg, s);
    }
    if (variant == 2) {
        int n_cu = 256;
        { const int rc_ = avx::device_cu_count(&n_cu); if (rc_ != AVEXHIP_OK) return rc_; }
        int n_wg = n_cu;
        if (const char* fg = getenv("AVEX_AMD_ATT_GRID")) { const int g = atoi(fg); if (g > 0) n_wg = g; }   // tests: several units per workgroup
        // A last query block of one or two rows (EAT's class token: 513 = 512 + 1) goes to the tail kernel instead of a further query block
        // of the streamed kernel.  Only that: in a short last block the waves without query rows skip their tiles, so the block is cheap
        // -- measured at 3 072 (clip, head) items (scripts/att_513.py): 513 tokens 0.441 ms with the tail, 0.491 without; 520 tokens
        // 0.532 / 0.485; 544 tokens 0.949 / 0.491 (and 1.79 ms with round 2's one-row-per-wave tail up to 32 rows).
        const int rem = Tn % 512;
        const int tail_max = getenv("AVEX_AMD_ATT_TAIL_ROWS") ? atoi(getenv("AVEX_AMD_ATT_TAIL_ROWS")) : 2;      // (tests raise it to 32)
        const bool use_tail = Tn > TMAX && rem > 0 && rem <= tail_max && rem <= 32 && !getenv("AVEX_AMD_ATT_NO_TAIL");
        const int nqb_main = Tn > TMAX ? (use_tail ? Tn / 512 : (Tn + 511) / 512) : 1;
        AVX_REQUIRE((int64_t)B * H * nqb_main < (1ll << 31), "attention: too many (item, query block) units");
        const int n_units = B * H * nqb_main;            // (clip, head, query block of 512) units, dealt to the workgroups in consecutive runs
        const int per_block = (n_units + n_wg - 1) / n_wg;
        const int grid = (n_units + per_block - 1) / per_block;
        if (Tn > TMAX) {
            // EAT's shape (513 .. 544 tokens, no bias table, the rows beyond 512 in the tail kernel): the main block on variant 3's nine-tile form
            const bool v3_long = !(getenv("AVEX_AMD_ATT_VARIANT") && atoi(getenv("AVEX_AMD_ATT_VARIANT")) == 2);      // (read per launch: A/B in one process)
            if (!bias_tab && use_tail && nqb_main == 1 && Tn <= TMAX + 32 && v3_long && !getenv("AVEX_AMD_ATT_NO_XT")) {
                const int rc3 = avx::attention16(qkv, B, Tn, H, bias_tab, grep_w, grep_b, grep_a, key_pad, out, __is_same(T, _Float16) ? AVEXHIP_F16 : AVEXHIP_BF16, q_log2e, n_wg, s);
                if (rc3 != AVEXHIP_OK) return rc3;
            } else if (bias_tab) {
                AVX_ENSURE_LDS((attention2_kernel<T, true, true>), ATT2L_LDS);
                hipLaunchKernelGGL((attention2_kernel<T, true, true>), dim3(grid), dim3(512), ATT2L_LDS, s, (const T*)qkv, Tn, H, B, per_block, nqb_main, bias_tab,
                                   grep_w, grep_b, grep_a, key_pad, (T*)out, q_log2e, dbg);
            } else if (Tn % 256 >= 1 && Tn % 256 <= 32 && !getenv("AVEX_AMD_ATT_NO_XT")) {
                // the 1 .. 32 keys beyond a multiple of 256 ride in the last full key block's phase as a ninth key tile (EAT: 513 keys)
                AVX_ENSURE_LDS((attention2_kernel<T, true, false, true>), ATT2X_LDS);
                hipLaunchKernelGGL((attention2_kernel<T, true, false, true>), dim3(grid), dim3(512), ATT2X_LDS, s, (const T*)qkv, Tn, H, B, per_block, nqb_main, bias_tab,
                                   grep_w, grep_b, grep_a, key_pad, (T*)out, q_log2e, dbg);
            } else {
                AVX_ENSURE_LDS((attention2_kernel<T, true, false>), ATT2L_LDS);
                hipLaunchKernelGGL((attention2_kernel<T, true, false>), dim3(grid), dim3(512), ATT2L_LDS, s, (const T*)qkv, Tn, H, B, per_block, nqb_main, bias_tab,
                                   grep_w, grep_b, grep_a, key_pad, (T*)out, q_log2e, dbg);
            }
            AVX_LAUNCH_CHECK();
            if (use_tail) {
                AVX_REQUIRE((int64_t)B * H * rem < (1ll << 31), "attention: too many tail rows");
                // rows per wave: as many as the tail has (up to 8) and as fit the LDS (RW x (Tn + 64) floats)
                int rw = rem >= 8 ? 8 : (rem >= 4 ? 4 : (rem >= 2 ? 2 : 1));
                while (rw > 1 && sizeof(float) * (size_t)rw * ((size_t)Tn + 65 + 256) > 150 * 1024) rw >>= 1;
                const size_t lds = sizeof(float) * (size_t)rw * ((size_t)Tn + 65 + 256);      // scores, q, gate, four partial output rows
                const dim3 tgrid((unsigned)(B * H * ((rem + rw - 1) / rw)));
#define AVX_TAIL(RWV) do { AVX_ENSURE_LDS((attention_tail_kernel<T, RWV>), 160 * 1024); \
                hipLaunchKernelGGL((attention_tail_kernel<T, RWV>), tgrid, dim3(256), lds, s, (const T*)qkv, Tn, H, Tn - rem, rem, bias_tab, grep_w, grep_b, \
                                   grep_a, key_pad, (T*)out, q_log2e); } while (0)
                if (rw == 8) AVX_TAIL(8); else if (rw == 4) AVX_TAIL(4); else if (rw == 2) AVX_TAIL(2); else AVX_TAIL(1);
#undef AVX_TAIL
            }
        } else if (bias_tab) {
            AVX_ENSURE_LDS((attention2_kernel<T, false, true>), ATT2_LDS);
            hipLaunchKernelGGL((attention2_kernel<T, false, true>), dim3(grid), dim3(512), ATT2_LDS, s, (const T*)qkv, Tn, H, B, per_block, 1, bias_tab,
                               grep_w, grep_b, grep_a, key_pad, (T*)out, q_log2e, dbg);
        } else {
            AVX_ENSURE_LDS((attention2_kernel<T, false, false>), ATT2_LDS);
            hipLaunchKernelGGL((attention2_kernel<T, false, false>), dim3(grid), dim3(512), ATT2_LDS, s, (const T*)qkv, Tn, H, B, per_block, 1, bias_tab,
                               grep_w, grep_b, grep_a, key_pad, (T*)out, q_log2e, dbg);
        }
        AVX_LAUNCH_CHECK();
        return AVEXHIP_OK;
    }
    AVX_ENSURE_LDS(attention_kernel<T>, ATT_LDS);      // (variant 1 only: behind the dispatch)
    hipLaunchKernelGGL(attention_kernel<T>, dim3(B * H), dim3(1024), ATT_LDS, s, (const T*)qkv, Tn, H, bias_tab, grep_w,
                       grep_b, grep_a, key_pad, (T*)out, q_log2e, dbg);
    AVX_LAUNCH_CHECK();
    return AVEXHIP_OK;
}

}  // namespace

#if defined(AVEX_DIAG) && ATT_STAMPS
extern "C" int avexhip_debug_att_stamps(unsigned long long* host_out, int n) {
    if (!host_out || n <= 0) return -1;
    if (n > 64 * 8 * 32 * 8) n = 64 * 8 * 32 * 8;
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_att_stamps), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -2;
}
#endif

namespace avx {

int attention(const void* qkv, int B, int T, int H, const float* bias_tab, const float* grep_w,
              const float* grep_b, const float* grep_a, const uint8_t* key_pad, void* out, int dtype,
              hipStream_t s, int q_log2e) {
    AVX_REQUIRE(qkv && out, "attention: null buffer");
    AVX_REQUIRE(B > 0 && H > 0, "attention: bad B=%d H=%d", B, H);
    AVX_REQUIRE(T > 0 && T <= 32768, "attention: T=%d tokens unsupported (1..32768)", T);
    AVX_REQUIRE(!grep_w || (grep_b && grep_a), "attention: grep_b/grep_a required with grep_w");
    if (dtype == AVEXHIP_F16) return launch<_Float16>(qkv, B, T, H, bias_tab, grep_w, grep_b, grep_a, key_pad, out, q_log2e, s);
    if (dtype == AVEXHIP_BF16) return launch<__bf16>(qkv, B, T, H, bias_tab, grep_w, grep_b, grep_a, key_pad, out, q_log2e, s);
    avexhip_set_error("attention: unknown dtype %d", dtype);
    return AVEXHIP_ERR_INVALID;
}

}  // namespace avx
