// Plain multi-head self-attention for head widths other than 64 (32, 96, 128; 64 is built too, as a cross-check of attention.hip) on gfx950.
//
// softmax( q k^T / sqrt(D) [+ -inf on padded keys] ) v  after a fused QKV projection, as torch.nn.MultiheadAttention computes it inside
// the reference's sequence probes: attention_probe.py:60-70 (embed 768, 8 heads -> head width 96) and the nn.TransformerEncoderLayer of
// transformer_probe.py:66-75 (configs/run_configs/*: 8 heads of 96).  No relative-position bias, no gate: the encoders' own attention
// (head width 64, with or without BEATs' gated bias) stays in attention.hip.
//
// One 512-thread workgroup (8 waves, 32 queries each) per (clip, head, block of 256 queries).  Keys go through LDS in chunks of 128:
// K rows padded by 16 bytes (conflict-free 16-byte fragment reads), V transposed on its way in (rows of 132 halves: conflict-free
// 8-byte reads).  Two workgroups share a CU (69 KB of LDS each at D = 128), so one stages while the other computes.  As in
// attention.hip's first variant the QUERY stays on the MFMA lane for both products (v_mfma_f32_32x32x16):
//     S^T[key][query] = K[key][:] . Q[query][:]        O^T[d][query] += V^T[d][key] * P^T[key][query]
// so the online-softmax state is per-lane scalar state and P never touches LDS.  Base-2 softmax: scores are scaled by log2(e) / sqrt(D)
// (by 1 / sqrt(D) alone when the caller folded log2(e) into W_q).
#include "common.h"

namespace {

constexpr int HD_KC = 128;                       // keys per staged chunk
constexpr int HD_VLD = HD_KC + 4;                // halves per V^T row
template <int D> constexpr int hd_krow() { return D * 2 + 16; }      // bytes per K row
template <int D> constexpr int hd_lds() { return HD_KC * hd_krow<D>() + D * HD_VLD * 2 + HD_KC * 4; }

template <typename T, int D>
__global__ __launch_bounds__(512) void attention_hd_kernel(const T* __restrict__ qkv, int Tn, int H, int nqb, const uint8_t* __restrict__ key_pad,
                                                           T* __restrict__ out, float sscale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename Half<T>::v8 v8;
    typedef typename Half<T>::v4 v4;
    constexpr int KROW = hd_krow<D>(), NS = D / 16, ND = D / 32, CPR = D / 8;      // 16-byte chunks per row
    char* Ks = smem;
    T* Vt = (T*)(smem + HD_KC * KROW);
    float* kadd = (float*)(smem + HD_KC * KROW + D * HD_VLD * 2);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qb = blockIdx.x % nqb, h = (blockIdx.x / nqb) % H, b = blockIdx.x / (nqb * H);
    const int E = H * D;
    const int64_t ld = 3 * (int64_t)E;
    const T* base = qkv + (int64_t)b * Tn * ld + h * D;
    const float NEG_INF = -__builtin_inff();
    const int hh = lane >> 5, r32 = lane & 31;

    const int nqt = (Tn + 31) >> 5;
    const int qt = qb * 8 + wave;
    const bool active = qt < nqt;                 // wave-uniform
    const int i = qt * 32 + r32;
    const int iq = i < Tn ? i : Tn - 1;
    v8 qf[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) qf[s] = *(const v8*)(base + (int64_t)iq * ld + 16 * s + 8 * hh);

    float m_run = NEG_INF, l_run = 0.f;
    f32x16 o[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;

    const int nkc = (Tn + HD_KC - 1) / HD_KC;
    for (int kc = 0; kc < nkc; ++kc) {
        if (kc) __syncthreads();
        // ---- stage the chunk: K rows, V^T, key mask (rows past the clip are zeros: their P is 0 and 0 * stale LDS could be NaN) ----
        for (int idx = tid; idx < HD_KC * CPR; idx += 512) {
            const int row = idx / CPR, c = idx - row * CPR;
            const int j = kc * HD_KC + row;
            uint4 kv = make_uint4(0, 0, 0, 0);
            v8 vv;
#pragma unroll
            for (int e = 0; e < 8; ++e) vv[e] = (T)0.0f;
            if (j < Tn) {
                kv = *(const uint4*)(base + (int64_t)j * ld + E + c * 8);
                vv = *(const v8*)(base + (int64_t)j * ld + 2 * E + c * 8);
            }
            *(uint4*)(Ks + row * KROW + c * 16) = kv;
#pragma unroll
            for (int e = 0; e < 8; ++e) Vt[(c * 8 + e) * HD_VLD + row] = vv[e];
        }
        if (tid < HD_KC) {
            const int j = kc * HD_KC + tid;
            bool ok = j < Tn;
            if (ok && key_pad) ok = key_pad[(int64_t)b * Tn + j] == 0;
            kadd[tid] = ok ? 0.f : NEG_INF;
        }
        __syncthreads();
        if (!active) continue;
        const int left = Tn - kc * HD_KC;
        const int nkt = left >= HD_KC ? HD_KC / 32 : (left + 31) >> 5;
        for (int kt = 0; kt < nkt; ++kt) {
            f32x16 S;
#pragma unroll
            for (int r = 0; r < 16; ++r) S[r] = 0.f;
            const int krow = kt * 32 + r32;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const v8 kf = *(const v8*)(Ks + krow * KROW + (hh + 2 * s) * 16);
                S = mfma32(kf, qf[s], S);
            }
            const int jb = kt * 32 + 4 * hh;
            const bool masked_tile = key_pad != nullptr || (kt * 32 + 32 > left);      // wave-uniform
            float sc[16];
            float mx = NEG_INF;
            if (masked_tile) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int jo = (r & 3) + 8 * (r >> 2);
                    sc[r] = __builtin_fmaf(S[r], sscale, kadd[jb + jo]);
                    mx = fmaxf(mx, sc[r]);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    sc[r] = S[r] * sscale;
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
                for (int d = 0; d < ND; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
            }
            v8 pf[2];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[s2][j] = (T)p[8 * s2 + j];      // p in [0, 1]
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int key0 = kt * 32 + 16 * s2 + 4 * hh;
#pragma unroll
                for (int d = 0; d < ND; ++d) {
                    const T* vr = Vt + (32 * d + r32) * HD_VLD + key0;
                    const v4 lo = *(const v4*)vr, hi = *(const v4*)(vr + 8);
                    v8 vf;
                    vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
                    vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
                    o[d] = mfma32(vf, pf[s2], o[d]);
                }
            }
        }
    }
    if (!active || i >= Tn) return;
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.f / l_tot;
    T* orow = out + ((int64_t)b * Tn + i) * E + h * D;
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            v4 a;
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] = Half<T>::from(o[d][4 * g + e] * inv);
            *(v4*)(orow + 32 * d + 8 * g + 4 * hh) = a;
        }
}

template <typename T, int D>
int launch_hd(const void* qkv, int B, int Tn, int H, const uint8_t* key_pad, void* out, int q_log2e, hipStream_t s) {
    AVX_ENSURE_LDS((attention_hd_kernel<T, D>), hd_lds<D>());      // per device, behind a mutex (api.cpp)
    const int nqb = (Tn + 255) / 256;
    const float sscale = (q_log2e ? 1.0f : 1.4426950408889634f) / sqrtf((float)D);
    hipLaunchKernelGGL((attention_hd_kernel<T, D>), dim3((unsigned)((int64_t)B * H * nqb)), dim3(512), hd_lds<D>(), s, (const T*)qkv, Tn, H, nqb, key_pad,
                       (T*)out, sscale);
    AVX_LAUNCH_CHECK();
    return AVEXHIP_OK;
}

template <typename T>
int launch_hd_any(const void* qkv, int B, int Tn, int H, int D, const uint8_t* key_pad, void* out, int q_log2e, hipStream_t s) {
    switch (D) {
    case 32: return launch_hd<T, 32>(qkv, B, Tn, H, key_pad, out, q_log2e, s);
    case 64: return launch_hd<T, 64>(qkv, B, Tn, H, key_pad, out, q_log2e, s);
    case 96: return launch_hd<T, 96>(qkv, B, Tn, H, key_pad, out, q_log2e, s);
    case 128: return launch_hd<T, 128>(qkv, B, Tn, H, key_pad, out, q_log2e, s);
    }
    avexhip_set_error("attention_hd: head width %d not built (32, 64, 96, 128)", D);
    return AVEXHIP_ERR_INVALID;
}

}  // namespace

namespace avx {

int attention_hd(const void* qkv, int B, int T, int H, int head_dim, const uint8_t* key_pad, void* out, int dtype, hipStream_t s, int q_log2e) {
    AVX_REQUIRE(qkv && out, "attention_hd: null buffer");
    AVX_REQUIRE(B > 0 && H > 0 && T > 0 && T <= 32768, "attention_hd: bad B=%d H=%d T=%d", B, H, T);
    AVX_REQUIRE((int64_t)B * H * ((T + 255) / 256) < (1ll << 31), "attention_hd: grid too large");
    if (dtype == AVEXHIP_F16) return launch_hd_any<_Float16>(qkv, B, T, H, head_dim, key_pad, out, q_log2e, s);
    if (dtype == AVEXHIP_BF16) return launch_hd_any<__bf16>(qkv, B, T, H, head_dim, key_pad, out, q_log2e, s);
    avexhip_set_error("attention_hd: unknown dtype %d", dtype);
    return AVEXHIP_ERR_INVALID;
}

}  // namespace avx

extern "C" int avexhip_attention_hd(const void* qkv, int B, int T, int H, int head_dim, const uint8_t* key_pad, void* out, int dtype, void* stream) {
    return avx::attention_hd(qkv, B, T, H, head_dim, key_pad, out, dtype, (hipStream_t)stream, 0);
}
