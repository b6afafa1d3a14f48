// Probe heads on the device: what the reference's online probes run after extract_embeddings(), so that hooked
// embeddings never leave HBM between the encoder and the logits.
//
//   layer mix      base_probes.py:197-206   out = sum_l softmax(layer_weights)_l * tap_l  (weights 1.0 each when there are none),
//                                           accumulated in list order with separate multiply and add like the reference's loop
//   dense          nn.Linear (+ ReLU / erf-GELU / Tanh, + residual): linear_probe.py:44-46,66; mlp_probe.py:51-73,91;
//                                           in_proj / out_proj / classifier of attention_probe.py:59-86,128-134
//   mha            the scaled-dot-product core of nn.MultiheadAttention(batch_first=True) as attention_probe.py:128 calls it
//                                           (self attention, key_padding_mask, no dropout in eval)
//
// Everything is fp32: the probes are trained and evaluated in fp32 by the reference and they are small next to the encoder
// (a linear probe is 0.03 % of one BEATs forward).  The dense kernel still uses the matrix core -- v_mfma_f32_32x32x2f32 runs exact
// fp32 FMA chains -- because the attention probe's in_proj over [B*T, E] rows is a real GEMM.
#include <math.h>

#include "common.h"

namespace {

constexpr int MIX_MAX = 16;
struct MixArgs {
    const float* tap[MIX_MAX];
    int L;
    const float* lw;      // raw layer weights [L] or NULL
};

__global__ __launch_bounds__(256) void layer_mix_kernel(MixArgs a, int64_t n, float* __restrict__ out) {
    float w[MIX_MAX];
    if (a.lw) {
        float mx = -__builtin_inff();
        for (int l = 0; l < a.L; ++l) mx = fmaxf(mx, a.lw[l]);
        float s = 0.f;
        for (int l = 0; l < a.L; ++l) { w[l] = expf(a.lw[l] - mx); s += w[l]; }
        for (int l = 0; l < a.L; ++l) w[l] = w[l] / s;
    } else {
        for (int l = 0; l < a.L; ++l) w[l] = 1.0f;
    }
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float acc = 0.f;
        for (int l = 0; l < a.L; ++l) acc = __fadd_rn(acc, __fmul_rn(w[l], a.tap[l][i]));   // out = out + w * emb, no contraction
        out[i] = acc;
    }
}

// out[m][n] = act(bias[n] + sum_k x[m][k] * w[n][k]) (+ resid[m][n]);  128 x 128 tile per 256-thread workgroup, K in steps of 16
// through LDS ([k][row] with a 4-word skew), each wave a 64 x 64 quadrant = 2 x 2 fp32 MFMA tiles.
constexpr int DT = 128, DK = 16, DLD = DT + 4;

__device__ __forceinline__ float probe_act(float v, int act) {
    if (act == 1) return fmaxf(v, 0.f);
    if (act == 2) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    if (act == 3) return tanhf(v);
    return v;
}

__global__ __launch_bounds__(256) void dense_f32_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ w,
                                                         int64_t ldw, const float* __restrict__ bias, const float* __restrict__ resid,
                                                         int64_t ldr, int M, int N, int K, int act, float* __restrict__ out, int64_t ldo) {
    __shared__ float xs[DK * DLD];
    __shared__ float ws[DK * DLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * DT, n0 = blockIdx.x * DT;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int fr = lane & 31, kh = lane >> 5;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const bool vec = (K % 4 == 0) && (ldx % 4 == 0) && (ldw % 4 == 0) && ((uintptr_t)x % 16 == 0) && ((uintptr_t)w % 16 == 0);
    for (int k0 = 0; k0 < K; k0 += DK) {
        // 128 rows x 16 k per operand: thread -> (row = tid / 4 + 64 * h, k4 = (tid % 4) * 4)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int row = (tid >> 2) + 64 * h, k4 = (tid & 3) * 4;
            float xv[4] = {0.f, 0.f, 0.f, 0.f}, wv[4] = {0.f, 0.f, 0.f, 0.f};
            const int gm = m0 + row, gn = n0 + row, gk = k0 + k4;
            if (gm < M) {
                if (vec && gk + 3 < K) {
                    const f32x4 v = *(const f32x4*)(x + (int64_t)gm * ldx + gk);
                    xv[0] = v[0]; xv[1] = v[1]; xv[2] = v[2]; xv[3] = v[3];
                } else {
                    for (int j = 0; j < 4; ++j) if (gk + j < K) xv[j] = x[(int64_t)gm * ldx + gk + j];
                }
            }
            if (gn < N) {
                if (vec && gk + 3 < K) {
                    const f32x4 v = *(const f32x4*)(w + (int64_t)gn * ldw + gk);
                    wv[0] = v[0]; wv[1] = v[1]; wv[2] = v[2]; wv[3] = v[3];
                } else {
                    for (int j = 0; j < 4; ++j) if (gk + j < K) wv[j] = w[(int64_t)gn * ldw + gk + j];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) { xs[(k4 + j) * DLD + row] = xv[j]; ws[(k4 + j) * DLD + row] = wv[j]; }
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < DK; kk += 2) {
            const float a0 = xs[(kk + kh) * DLD + wm + fr], a1 = xs[(kk + kh) * DLD + wm + 32 + fr];
            const float b0 = ws[(kk + kh) * DLD + wn + fr], b1 = ws[(kk + kh) * DLD + wn + 32 + fr];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int gn = n0 + wn + 32 * j + fr;
            if (gn >= N) continue;
            const float bv = bias ? bias[gn] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gm = m0 + wm + 32 * i + 8 * (r >> 2) + 4 * kh + (r & 3);
                if (gm >= M) continue;
                float v = probe_act(acc[i][j][r] + bv, act);
                if (resid) v += resid[(int64_t)gm * ldr + gn];
                out[(int64_t)gm * ldo + gn] = v;
            }
        }
}

// Self attention core for one (clip, head, 16-query block): q, k, v are the thirds of qkv rows [B*T, 3E], head h at columns
// h*hd .. h*hd+hd-1 of each third (nn.MultiheadAttention's in_proj layout).  Scores for the 16 queries over all T keys sit in
// LDS; thread (q = tid / 16, j = tid % 16) owns keys j, j+16, ... and, for the output, head columns j, j+16, ...
constexpr int MQ = 16, MHD_MAX = 128;

__global__ __launch_bounds__(256) void mha_f32_kernel(const float* __restrict__ qkv, int T, int E, int H, int hd,
                                                       const uint8_t* __restrict__ key_pad, float* __restrict__ out) {
    extern __shared__ float smem[];
    float* qs = smem;                       // [MQ][hd]
    float* sc = smem + MQ * MHD_MAX;        // [MQ][T]
    const int tid = threadIdx.x;
    const int qb = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int q0 = qb * MQ;
    const int64_t ld = 3 * (int64_t)E;
    const float* base = qkv + (int64_t)b * T * ld + (int64_t)h * hd;
    const float scale = 1.0f / sqrtf((float)hd);
    for (int i = tid; i < MQ * hd; i += 256) {
        const int q = i / hd, d = i % hd;
        qs[q * MHD_MAX + d] = q0 + q < T ? base[(int64_t)(q0 + q) * ld + d] * scale : 0.f;
    }
    __syncthreads();
    const int q = tid >> 4, j0 = tid & 15;
    float mx = -__builtin_inff();
    for (int j = j0; j < T; j += 16) {
        const float* kr = base + (int64_t)j * ld + E;
        float s = 0.f;
        for (int d = 0; d < hd; d += 4) {
            const f32x4 kv = *(const f32x4*)(kr + d);
            const f32x4 qv = *(const f32x4*)(qs + q * MHD_MAX + d);
            s += (qv[0] * kv[0] + qv[1] * kv[1]) + (qv[2] * kv[2] + qv[3] * kv[3]);
        }
        if (key_pad && key_pad[(int64_t)b * T + j]) s = -__builtin_inff();
        sc[q * T + j] = s;
        mx = fmaxf(mx, s);
    }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
    for (int j = j0; j < T; j += 16) {
        const float e = expf(sc[q * T + j] - mx);
        sc[q * T + j] = e;
        sum += e;
    }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) sum += __shfl_xor(sum, o);
    const float inv = 1.0f / sum;
    __syncthreads();   // (the 16 threads of one query are in one wave, but keep the LDS hand-over explicit)
    if (q0 + q >= T) return;
    const float* vb = base + 2 * E;
    for (int d = j0; d < hd; d += 16) {
        float o = 0.f;
        for (int j = 0; j < T; ++j) o = __builtin_fmaf(sc[q * T + j], vb[(int64_t)j * ld + d], o);
        out[((int64_t)b * T + q0 + q) * E + (int64_t)h * hd + d] = o * inv;
    }
}


// The same attention core on the fp32 matrix core for head dims 32 / 64 / 96 / 128.  One wave owns 32 queries; the four waves of a
// workgroup share 32-key tiles of K and V staged in LDS.  Scores are formed TRANSPOSED (keys x queries: A = K tile, B = Q^T from
// registers) so that each lane ends up with 16 scores of ONE query (column lane & 31); the running max / sum of the online softmax
// are then per-lane scalars with a single cross-half exchange, and the probabilities are, register for register, the B operand of
// the second product  O^T[d][q] += V^T[d][j] P^T[j][q]  (accumulator register r of half kh holds key 8*(r>>2) + 4*kh + (r&3), which
// is exactly the pair of keys MFMA step r contracts when the A operand reads V rows by that same formula).
template <int HD>
__global__ __launch_bounds__(256) void mha_mfma_kernel(const float* __restrict__ qkv, int T, int E, const uint8_t* __restrict__ key_pad,
                                                        float* __restrict__ out) {
    constexpr int LDK = HD + 4, LDV = HD + 8, HH = HD / 2, NDT = HD / 32;
    __shared__ float ks[32 * LDK];
    __shared__ float vs[32 * LDV];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, kh = lane >> 5;
    const int h = blockIdx.y, b = blockIdx.z;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const int64_t ld = 3 * (int64_t)E;
    const float* base = qkv + (int64_t)b * T * ld + (int64_t)h * HD;
    const float scale = 1.0f / sqrtf((float)HD);
    const bool active = q0 < T;
    float qf[HH];
    {
        const int q = q0 + fr < T ? q0 + fr : T - 1;
        const float* qr = base + (int64_t)q * ld + kh * HH;
#pragma unroll
        for (int i = 0; i < HH; i += 4) {
            const f32x4 v = *(const f32x4*)(qr + i);
            qf[i] = v[0] * scale; qf[i + 1] = v[1] * scale; qf[i + 2] = v[2] * scale; qf[i + 3] = v[3] * scale;
        }
    }
    f32x16 o[NDT];
#pragma unroll
    for (int t = 0; t < NDT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float m = -__builtin_inff(), l = 0.f;
    for (int j0 = 0; j0 < T; j0 += 32) {
        __syncthreads();
        for (int i = tid; i < 32 * (HD / 4); i += 256) {
            const int j = i / (HD / 4), d4 = (i % (HD / 4)) * 4;
            f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
            if (j0 + j < T) {
                kv = *(const f32x4*)(base + (int64_t)(j0 + j) * ld + E + d4);
                vv = *(const f32x4*)(base + (int64_t)(j0 + j) * ld + 2 * E + d4);
            }
            *(f32x4*)(ks + j * LDK + d4) = kv;
            *(f32x4*)(vs + j * LDV + d4) = vv;
        }
        __syncthreads();
        if (!active) continue;
        f32x16 sacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
        const float* kr = ks + fr * LDK + kh * HH;
#pragma unroll
        for (int i = 0; i < HH; i += 4) {
            const f32x4 kv = *(const f32x4*)(kr + i);
            sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kv[0], qf[i], sacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kv[1], qf[i + 1], sacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kv[2], qf[i + 2], sacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kv[3], qf[i + 3], sacc, 0, 0, 0);
        }
        // dead keys (past T, or padded) as a bit per accumulator register; branch-free selects below (element-wise conditional
        // stores into the accumulator vector made the compiler emit divergent copies of all 16 registers -- and wrong results)
        unsigned dead = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) dead |= (unsigned)(j0 + 8 * (r >> 2) + 4 * kh + (r & 3) >= T) << r;
        if (key_pad) {
            const uint8_t* kp = key_pad + (int64_t)b * T;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = j0 + 8 * (r >> 2) + 4 * kh + (r & 3);
                dead |= (unsigned)(kp[j < T ? j : T - 1] != 0) << r;
            }
        }
        float mloc = -__builtin_inff();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            sacc[r] = (dead >> r) & 1u ? -__builtin_inff() : sacc[r];
            mloc = fmaxf(mloc, sacc[r]);
        }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
        const float mn = fmaxf(m, mloc);
        const float shift = mn == -__builtin_inff() ? 0.f : mn;
        const float f = expf(m - shift);
        m = mn;
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sacc[r] = expf(sacc[r] - shift); psum += sacc[r]; }
        l = l * f + psum;
#pragma unroll
        for (int t = 0; t < NDT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[t][r] *= f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float* vr = vs + (8 * (r >> 2) + 4 * kh + (r & 3)) * LDV + fr;
#pragma unroll
            for (int t = 0; t < NDT; ++t) o[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[32 * t], sacc[r], o[t], 0, 0, 0);
        }
    }
    if (!active || q0 + fr >= T) return;
    l += __shfl_xor(l, 32);
    const float inv = 1.0f / l;
    float* orow = out + ((int64_t)b * T + q0 + fr) * E + (int64_t)h * HD;
#pragma unroll
    for (int t = 0; t < NDT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 v = {o[t][4 * g] * inv, o[t][4 * g + 1] * inv, o[t][4 * g + 2] * inv, o[t][4 * g + 3] * inv};
            *(f32x4*)(orow + 32 * t + 8 * g + 4 * kh) = v;
        }
}

}  // namespace

// torch.nn.functional.interpolate(x.transpose(1, 2), size=Tout, mode="linear", align_corners=False).transpose(1, 2) on [B, Tin, C]
// rows (base_probes.py:398-411: taps of different sequence lengths are brought to the shortest one): source position
// (t + 0.5) * Tin / Tout - 0.5 clamped at 0, the two neighbours weighted (1 - w) and w in that order.
__global__ __launch_bounds__(256) void seq_interp_linear_kernel(const float* __restrict__ in, int Tin, int C, int Tout, float scale, float* __restrict__ out) {
    const int t = blockIdx.x, b = blockIdx.y;
    float src = scale * ((float)t + 0.5f) - 0.5f;
    src = src < 0.f ? 0.f : src;
    const int i0 = (int)src;
    const int i1 = i0 + (i0 < Tin - 1 ? 1 : 0);
    const float w1 = src - (float)i0, w0 = 1.f - w1;
    const float* r0 = in + ((int64_t)b * Tin + i0) * C;
    const float* r1 = in + ((int64_t)b * Tin + i1) * C;
    float* o = out + ((int64_t)b * Tout + t) * C;
    for (int c = threadIdx.x; c < C; c += 256) o[c] = w0 * r0[c] + w1 * r1[c];
}

extern "C" int avexhip_seq_interp_linear(const float* in_dev, int B, int Tin, int C, int Tout, float* out_dev, void* stream) {
    AVX_REQUIRE(in_dev && out_dev, "seq_interp_linear: null argument");
    AVX_REQUIRE(B > 0 && Tin > 0 && Tout > 0 && C > 0 && B <= 65535, "seq_interp_linear: bad shape B=%d Tin=%d Tout=%d C=%d", B, Tin, Tout, C);
    seq_interp_linear_kernel<<<dim3(Tout, B), dim3(256), 0, (hipStream_t)stream>>>(in_dev, Tin, C, Tout, (float)Tin / (float)Tout, out_dev);
    AVX_LAUNCH_CHECK();
    return AVEXHIP_OK;
}

extern "C" int avexhip_layer_mix(const float* const* taps, int L, const float* layer_weights, int64_t n, float* out, void* stream) {
    AVX_REQUIRE(taps && out && L >= 1 && L <= MIX_MAX && n >= 0, "layer_mix: need 1..%d taps, got %d", MIX_MAX, L);
    if (n == 0) return AVEXHIP_OK;
    MixArgs a;
    for (int l = 0; l < MIX_MAX; ++l) a.tap[l] = l < L ? taps[l] : nullptr;
    for (int l = 0; l < L; ++l) AVX_REQUIRE(a.tap[l], "layer_mix: tap %d is NULL", l);
    a.L = L;
    a.lw = layer_weights;
    const int64_t blocks = (n + 255) / 256;
    layer_mix_kernel<<<dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, (hipStream_t)stream>>>(a, n, out);
    AVX_LAUNCH_CHECK();
    return AVEXHIP_OK;
}

extern "C" int avexhip_dense_f32(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, const float* resid,
                                 int64_t ldr, int M, int N, int K, int act, float* out, int64_t ldo, void* stream) {
    AVX_REQUIRE(x && w && out && M >= 0 && N >= 1 && K >= 1, "dense_f32: bad arguments (M %d N %d K %d)", M, N, K);
    AVX_REQUIRE(ldx >= K && ldw >= K && ldo >= N && (!resid || ldr >= N), "dense_f32: leading dimensions too small");
    AVX_REQUIRE(act >= 0 && act <= 3, "dense_f32: activation %d (0 none, 1 relu, 2 gelu, 3 tanh)", act);
    if (M == 0) return AVEXHIP_OK;
    const int gy = (M + DT - 1) / DT;
    AVX_REQUIRE(gy <= 65535, "dense_f32: M %d too large", M);
    dense_f32_kernel<<<dim3((N + DT - 1) / DT, gy), dim3(256), 0, (hipStream_t)stream>>>(x, ldx, w, ldw, bias, resid, ldr, M, N, K, act, out,
                                                                                      ldo);
    AVX_LAUNCH_CHECK();
    return AVEXHIP_OK;
}

extern "C" int avexhip_mha_f32(const float* qkv, int B, int T, int E, int H, const uint8_t* key_pad, float* out, void* stream) {
    AVX_REQUIRE(qkv && out && B >= 0 && T >= 1 && H >= 1 && E >= H && E % H == 0, "mha_f32: bad arguments");
    const int hd = E / H;
    AVX_REQUIRE(hd % 4 == 0 && hd <= MHD_MAX, "mha_f32: head dim %d (need a multiple of 4, <= %d)", hd, MHD_MAX);
    AVX_REQUIRE(T <= 2048 && B <= 65535 && H <= 65535, "mha_f32: T %d > 2048 keys", T);
    if (B == 0) return AVEXHIP_OK;
    if (hd == 32 || hd == 64 || hd == 96 || hd == 128) {
        const dim3 grid((T + 127) / 128, H, B);
        if (hd == 32) mha_mfma_kernel<32><<<grid, dim3(256), 0, (hipStream_t)stream>>>(qkv, T, E, key_pad, out);
        else if (hd == 64) mha_mfma_kernel<64><<<grid, dim3(256), 0, (hipStream_t)stream>>>(qkv, T, E, key_pad, out);
        else if (hd == 96) mha_mfma_kernel<96><<<grid, dim3(256), 0, (hipStream_t)stream>>>(qkv, T, E, key_pad, out);
        else mha_mfma_kernel<128><<<grid, dim3(256), 0, (hipStream_t)stream>>>(qkv, T, E, key_pad, out);
        AVX_LAUNCH_CHECK();
        return AVEXHIP_OK;
    }
    const size_t lds = sizeof(float) * ((size_t)MQ * MHD_MAX + (size_t)MQ * T);
    AVX_ENSURE_LDS(mha_f32_kernel, sizeof(float) * (MQ * MHD_MAX + MQ * 2048));
    mha_f32_kernel<<<dim3((T + MQ - 1) / MQ, H, B), dim3(256), lds, (hipStream_t)stream>>>(qkv, T, E, H, hd, key_pad, out);
    AVX_LAUNCH_CHECK();
    return AVEXHIP_OK;
}
