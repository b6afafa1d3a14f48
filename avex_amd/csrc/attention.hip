// Gated relative-position-bias self-attention for BEATs (head_dim 64, T <= 512) on gfx950.
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
                                                        T* __restrict__ out, int dbg) {
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
        gw[tid] = a;
        gw[64 + tid] = bb;
        if (tid == 0) {
            gw[128] = grep_w ? (grep_b[0] + grep_b[1]) + (grep_b[2] + grep_b[3]) : 0.f;
            gw[129] = grep_w ? (grep_b[4] + grep_b[5]) + (grep_b[6] + grep_b[7]) : 0.f;
        }
    }
    __syncthreads();

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
                    sc[r] = __builtin_fmaf(S[r], 0.125f * 1.4426950408889634f, gate * tv[r]) + kadd[jb + jo];
                    mx = fmaxf(mx, sc[r]);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int jo = (r & 3) + 8 * (r >> 2);
                    sc[r] = __builtin_fmaf(S[r], 0.125f * 1.4426950408889634f, gate * tv[r]);
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

template <typename T>
int launch(const void* qkv, int B, int Tn, int H, const float* bias_tab, const float* grep_w, const float* grep_b,
           const float* grep_a, const uint8_t* key_pad, void* out, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        AVX_HIP_CHECK(hipFuncSetAttribute((const void*)attention_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, ATT_LDS));
        attr_set = true;
    }
    static const int dbg = getenv("AVEX_AMD_ATT_DEBUG") ? atoi(getenv("AVEX_AMD_ATT_DEBUG")) : 0;
    hipLaunchKernelGGL(attention_kernel<T>, dim3(B * H), dim3(1024), ATT_LDS, s, (const T*)qkv, Tn, H, bias_tab, grep_w,
                       grep_b, grep_a, key_pad, (T*)out, dbg);
    AVX_LAUNCH_CHECK();
    return AVEXHIP_OK;
}

}  // namespace

namespace avx {

int attention(const void* qkv, int B, int T, int H, const float* bias_tab, const float* grep_w,
              const float* grep_b, const float* grep_a, const uint8_t* key_pad, void* out, int dtype,
              hipStream_t s) {
    AVX_REQUIRE(qkv && out, "attention: null buffer");
    AVX_REQUIRE(B > 0 && H > 0, "attention: bad B=%d H=%d", B, H);
    AVX_REQUIRE(T > 0 && T <= TMAX, "attention: T=%d tokens unsupported (1..%d; clips up to ~10.3 s)", T, TMAX);
    AVX_REQUIRE(!grep_w || (grep_b && grep_a), "attention: grep_b/grep_a required with grep_w");
    if (dtype == AVEXHIP_F16) return launch<_Float16>(qkv, B, T, H, bias_tab, grep_w, grep_b, grep_a, key_pad, out, s);
    if (dtype == AVEXHIP_BF16) return launch<__bf16>(qkv, B, T, H, bias_tab, grep_w, grep_b, grep_a, key_pad, out, s);
    avexhip_set_error("attention: unknown dtype %d", dtype);
    return AVEXHIP_ERR_INVALID;
}

}  // namespace avx
