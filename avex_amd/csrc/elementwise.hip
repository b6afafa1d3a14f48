// HBM-bound helpers of the BEATs path: dtype casts, LayerNorm (fp32 stats, fp32 + half outputs),
// mean pooling over tokens, fbank -> patch-major half layout.
//
// Reference call sites: nn.LayerNorm at beats.py:275,353 (512-wide, patch features) and
// backbone.py:106,176-177 / :294,362 / :302,373 (768-wide); features.mean(dim=1) at README:80 and
// beats_model.py:275; Conv2d patch geometry at beats.py:263-269,350-352.
#include "common.h"

namespace {

template <typename T>
__global__ void cast_to_half_kernel(const float* __restrict__ in, T* __restrict__ out, int64_t n, unsigned int* __restrict__ ovf) {
    int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
    float ovf_mx = 0.f;      // range alarm of an f16 destination (values saturate at +-65504): a handle's weight upload refuses a checkpoint that does not fit
    for (; i + 3 < n; i += stride) {
        const f32x4 v = *(const f32x4*)(in + i);
        ovf_see4<T>(ovf_mx, v);
        typename Half<T>::v4 h;
        h[0] = Half<T>::from(v[0]); h[1] = Half<T>::from(v[1]);
        h[2] = Half<T>::from(v[2]); h[3] = Half<T>::from(v[3]);
        *(typename Half<T>::v4*)(out + i) = h;
    }
    // tail (n % 4) handled by the first threads of block 0
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t j = (n & ~(int64_t)3) + threadIdx.x;
        ovf_see<T>(ovf_mx, in[j], 0.f);
        out[j] = Half<T>::from(in[j]);
    }
    ovf_commit<T>(ovf, ovf_mx);
}

template <typename T>
__global__ void cast_to_f32_kernel(const T* __restrict__ in, float* __restrict__ out, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = (float)in[i];
}

// One wave per row; lanes hold float4 slices lane, lane+64, ... (C <= 1024).  Two-pass statistics in
// registers (mean, then centred variance) like torch's CPU LayerNorm.
template <typename T>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ in, const T* __restrict__ in_h,
                                                        int64_t ld_in, const float* __restrict__ w,
                                                        const float* __restrict__ b, float eps, int M,
                                                        int C, float* __restrict__ out_f32, int64_t ldo,
                                                        T* __restrict__ out_h, int64_t ldh) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nv = C >> 2;  // float4 per row
    f32x4 v[4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = lane + 64 * i;
        if (idx < nv) {
            if (in_h) {
                const typename Half<T>::v4 hv = *(const typename Half<T>::v4*)(in_h + (int64_t)row * ld_in + idx * 4);
                v[i] = (f32x4){(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]};
            } else {
                v[i] = *(const f32x4*)(in + (int64_t)row * ld_in + idx * 4);
            }
            s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        } else {
            v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = lane + 64 * i;
        if (idx < nv) {
            v[i] -= mean;
            q += (v[i][0] * v[i][0] + v[i][1] * v[i][1]) + (v[i][2] * v[i][2] + v[i][3] * v[i][3]);
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = lane + 64 * i;
        if (idx < nv) {
            const f32x4 ww = *(const f32x4*)(w + idx * 4);
            const f32x4 bb = *(const f32x4*)(b + idx * 4);
            const f32x4 y = v[i] * rstd * ww + bb;
            if (out_f32) *(f32x4*)(out_f32 + (int64_t)row * ldo + idx * 4) = y;
            if (out_h) {
                typename Half<T>::v4 h;
                h[0] = Half<T>::from(y[0]); h[1] = Half<T>::from(y[1]);
                h[2] = Half<T>::from(y[2]); h[3] = Half<T>::from(y[3]);
                *(typename Half<T>::v4*)(out_h + (int64_t)row * ldh + idx * 4) = h;
            }
        }
    }
}

// Operand-type residual stream: half in -> half out (+ optional fp32 out).  One 32-lane half-wave per
// row with 16-byte accesses (8 elements per lane per access, C/8 <= 96 chunks -> up to 3 per lane), so
// a wave instruction moves 2 x 512 contiguous bytes; the 8-byte-per-lane form of the generic kernel ran
// at 2.6 TB/s on these rows.
template <typename T>
__global__ __launch_bounds__(256) void layernorm_half_kernel(const T* __restrict__ in_h, int64_t ld_in,
                                                             const float* __restrict__ w, const float* __restrict__ b,
                                                             float eps, int M, int C, float* __restrict__ out_f32,
                                                             int64_t ldo, T* __restrict__ out_h, int64_t ldh) {
    typedef typename Half<T>::v8 v8;
    const int l32 = threadIdx.x & 31;
    const int row = blockIdx.x * 8 + (threadIdx.x >> 5);
    const bool live = row < M;
    const int nc = C >> 3;   // 16-byte chunks per row
    float v[3][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int c = l32 + 32 * i;
        if (live && c < nc) {
            const v8 hv = *(const v8*)(in_h + (int64_t)row * ld_in + c * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) { v[i][e] = (float)hv[e]; s += v[i][e]; }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
        }
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 32);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int c = l32 + 32 * i;
        if (c < nc) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { v[i][e] -= mean; q += v[i][e] * v[i][e]; }
        }
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) q += __shfl_xor(q, o, 32);
    const float rstd = 1.0f / sqrtf(q / (float)C + eps);
    if (!live) return;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int c = l32 + 32 * i;
        if (c < nc) {
            const f32x4 w0 = *(const f32x4*)(w + c * 8), w1 = *(const f32x4*)(w + c * 8 + 4);
            const f32x4 b0 = *(const f32x4*)(b + c * 8), b1 = *(const f32x4*)(b + c * 8 + 4);
            float y[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                y[e] = v[i][e] * rstd * w0[e] + b0[e];
                y[4 + e] = v[i][4 + e] * rstd * w1[e] + b1[e];
            }
            if (out_f32) {
                *(f32x4*)(out_f32 + (int64_t)row * ldo + c * 8) = (f32x4){y[0], y[1], y[2], y[3]};
                *(f32x4*)(out_f32 + (int64_t)row * ldo + c * 8 + 4) = (f32x4){y[4], y[5], y[6], y[7]};
            }
            if (out_h) {
                v8 h;
#pragma unroll
                for (int e = 0; e < 8; ++e) h[e] = Half<T>::from(y[e]);
                *(v8*)(out_h + (int64_t)row * ldh + c * 8) = h;
            }
        }
    }
}

// Final LayerNorm + mean over tokens in one pass (features.mean(dim=1) of the last LayerNorm's output, beats_model.py:275 / README:80), for
// callers that want the pooled embedding only: the fp32 feature tensor (390 MB at 256 clips) is neither written nor read back.
// One 1024-thread workgroup per clip; a half-wave owns a row (same per-row arithmetic as layernorm_half_kernel: two-pass statistics in
// registers), two rows in flight per half-wave; column sums of (v - mean) * rstd stay in registers, the 32 half-waves are combined through
// LDS in a fixed order (deterministic), weight and bias are applied once at the end:  mean_t(LN(x)_t) = w * mean_t((x_t - mu_t) rstd_t) + b.
template <typename T>
__global__ __launch_bounds__(1024) void layernorm_pool_kernel(const T* __restrict__ in_h, int64_t ld_in, const float* __restrict__ w,
                                                              const float* __restrict__ b, float eps, int Tn, int C,
                                                              float* __restrict__ out) {
    typedef typename Half<T>::v8 v8;
    __shared__ float part[16][768];
    const int l32 = threadIdx.x & 31, hw = threadIdx.x >> 5, wave = threadIdx.x >> 6;
    const int nc = C >> 3;
    const T* base = in_h + (int64_t)blockIdx.x * Tn * ld_in;
    float acc[3][8];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[i][e] = 0.f;
    for (int r0 = hw; r0 < Tn; r0 += 64) {
        float v[2][3][8];
        float s[2] = {0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int row = r0 + 32 * u;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int c = l32 + 32 * i;
                if (row < Tn && c < nc) {
                    const v8 hv = *(const v8*)(base + (int64_t)row * ld_in + c * 8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) { v[u][i][e] = (float)hv[e]; s[u] += v[u][i][e]; }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[u][i][e] = 0.f;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float su = s[u];
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) su += __shfl_xor(su, o, 32);
            const float mean = su / (float)C;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int c = l32 + 32 * i;
                if (c < nc) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { v[u][i][e] -= mean; q += v[u][i][e] * v[u][i][e]; }
                }
            }
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) q += __shfl_xor(q, o, 32);
            const float rstd = 1.0f / sqrtf(q / (float)C + eps);
            if (r0 + 32 * u < Tn) {
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[i][e] += v[u][i][e] * rstd;
            }
        }
    }
    // the two half-waves of a wave, then the 16 waves through LDS
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[i][e] += __shfl_xor(acc[i][e], 32, 64);
    if ((threadIdx.x & 63) < 32) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int c = l32 + 32 * i;
            if (c < nc) {
#pragma unroll
                for (int e = 0; e < 8; ++e) part[wave][c * 8 + e] = acc[i][e];
            }
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 1024) {
        float tot = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) tot += part[k][c];
        out[(int64_t)blockIdx.x * C + c] = tot / (float)Tn * w[c] + b[c];
    }
}

// in [B,T,C] -> out [B,C]; block (64 columns x 4 token phases), grid (C/64, B).
// With frame_pad: masked mean over non-padded tokens (beats_model.py:269-273).
__global__ __launch_bounds__(256) void mean_pool_kernel(const float* __restrict__ in, int T, int C,
                                                        const uint8_t* __restrict__ frame_pad,
                                                        float* __restrict__ out) {
    __shared__ float part[4][64];
    __shared__ int cnt[4];
    const int b = blockIdx.y;
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int ph = threadIdx.x >> 6;
    float s = 0.f;
    int n = 0;
    if (c < C) {
        for (int t = ph; t < T; t += 4) {
            const bool pad = frame_pad && frame_pad[(int64_t)b * T + t];
            if (!pad) {
                s += in[((int64_t)b * T + t) * C + c];
                ++n;
            }
        }
    }
    part[ph][threadIdx.x & 63] = s;
    if ((threadIdx.x & 63) == 0) cnt[ph] = n;
    __syncthreads();
    if (ph == 0 && c < C) {
        const float tot = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
        int nn = cnt[0] + cnt[1] + cnt[2] + cnt[3];
        if (nn < 1) nn = 1;
        out[(int64_t)b * C + c] = tot / (float)nn;
    }
}

// fbank [B, frames, n_mels] fp32 -> patches [B, nt*nf, P*P] half with token = tp*nf + fq,
// element = (frame % P) * P + (mel % P)   (Conv2d(1,D,P,stride=P) im2col, beats.py:349-352).
template <typename T>
__global__ void patchify_kernel(const float* __restrict__ fb, int frames, int n_mels, int P, int nt,
                                int nf, T* __restrict__ out, int64_t total) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int pp = P * P;
    const int e = (int)(i % pp);
    const int64_t tok = i / pp;
    const int fq = (int)(tok % nf);
    const int tp = (int)((tok / nf) % nt);
    const int64_t b = tok / ((int64_t)nf * nt);
    const int fr = tp * P + e / P, mel = fq * P + e % P;
    out[i] = Half<T>::from(fb[(b * frames + fr) * n_mels + mel]);
}

#ifdef AVEX_DIAG
// Debug aid: fills a static LDS array with a pattern and re-checks it for a while; any foreign write
// into this workgroup's LDS shows up in report[] = {mismatch count, first bad word index, value seen, block}.
__global__ __launch_bounds__(256) void lds_canary_kernel(int iters, unsigned* __restrict__ report) {
    __shared__ __attribute__((aligned(16))) unsigned words[6720];   // 26880 B, the fbank kernel's footprint
    uint4* w4 = (uint4*)words;
    uint2* w2 = (uint2*)words;
    auto want = [&](int i) { return 0xC0DE0000u ^ (unsigned)i ^ (blockIdx.x << 16); };
    for (int i = threadIdx.x; i < 1680; i += 256) w4[i] = make_uint4(want(4 * i), want(4 * i + 1), want(4 * i + 2), want(4 * i + 3));
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        // 128-bit reads + rewrites
        for (int i = threadIdx.x; i < 1680; i += 256) {
            const uint4 v = w4[i];
            if (v.x != want(4 * i) || v.y != want(4 * i + 1) || v.z != want(4 * i + 2) || v.w != want(4 * i + 3)) {
                if (atomicAdd(&report[0], 1u) == 0) { report[1] = (unsigned)(4 * i); report[2] = v.x; report[3] = blockIdx.x; }
            }
            w4[i] = make_uint4(want(4 * i), want(4 * i + 1), want(4 * i + 2), want(4 * i + 3));
        }
        __syncthreads();
        // paired 64-bit reads at a 64-element stride (ds_read2st64_b64-style) + rewrites
        for (int i = threadIdx.x; i < 1600; i += 256) {
            const uint2 a = w2[i], b = w2[i + 64];
            if (a.x != want(2 * i) || a.y != want(2 * i + 1) || b.x != want(2 * i + 128) || b.y != want(2 * i + 129)) {
                if (atomicAdd(&report[0], 1u) == 0) { report[1] = (unsigned)(2 * i); report[2] = a.x; report[3] = blockIdx.x | 0x80000000u; }
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 1600; i += 256) {
            w2[i] = make_uint2(want(2 * i), want(2 * i + 1));
            w2[i + 64] = make_uint2(want(2 * i + 128), want(2 * i + 129));
        }
        __syncthreads();
    }
}
#endif  // AVEX_DIAG

// EAT / Data2Vec-multi token assembly: row 0 of a clip is the class token, row 1 + t is patch t plus its fixed position; every
// row then goes through the encoder's first LayerNorm (`pre_norm`, the context encoder's norm with layer_norm_first = False).
// One wave per output row, C <= 1024, C % 4 == 0.
template <typename T>
__global__ __launch_bounds__(256) void token_embed_ln_kernel(const T* __restrict__ patches, const float* __restrict__ pos, const float* __restrict__ cls,
                                                             const float* __restrict__ w, const float* __restrict__ b, float eps, int Tp, int C, int64_t rows,
                                                             T* __restrict__ out_h, float* __restrict__ out_f) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int64_t clip = row / (Tp + 1);
    const int t = (int)(row - clip * (Tp + 1)) - 1;          // -1: class token
    float v[16];
    float s = 0.f;
    const int n = C / 64;                                    // values per lane (strided by 64): C = 768 -> 12
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        v[i] = 0.f;
        if (i < n) {
            const int c = lane + 64 * i;
            v[i] = t < 0 ? cls[c] : (float)patches[(clip * Tp + t) * C + c] + pos[(int64_t)t * C + c];
            s += v[i];
        }
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) if (i < n) { const float d = v[i] - mean; q = __builtin_fmaf(d, d, q); }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (i < n) {
            const int c = lane + 64 * i;
            const float y = (v[i] - mean) * rstd * w[c] + b[c];
            if (out_h) out_h[row * C + c] = Half<T>::from(y);
            if (out_f) out_f[row * C + c] = y;
        }
    }
}

// rows m with pad[m] != 0 become zero in the fp32 and/or operand-type copy of x (the encoder's `x[padding_mask] = 0`,
// backbone.py:169-170, on the path where no GEMM epilogue does it: embed_dim == encoder_embed_dim, no post_extract_proj)
__global__ __launch_bounds__(256) void zero_rows_kernel(float* __restrict__ x32, unsigned short* __restrict__ xh, int64_t ld32, int64_t ldh,
                                                        int C, const uint8_t* __restrict__ pad) {
    const int m = blockIdx.x;
    if (!pad[m]) return;
    for (int c = threadIdx.x; c < C; c += 256) {
        if (x32) x32[(int64_t)m * ld32 + c] = 0.f;
        if (xh) xh[(int64_t)m * ldh + c] = 0;          // +0.0 in f16 and in bf16
    }
}

// GLU_Linear's gate (modules.py:155-171, glu_type "swish"): out[m][f] = y[m][f] * swish(y[m][F + f]); 8 elements per thread
template <typename T>
__global__ __launch_bounds__(256) void glu_swish_kernel(const T* __restrict__ in, int64_t M, int F, T* __restrict__ out, unsigned int* __restrict__ ovf) {
    typedef typename Half<T>::v8 v8;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int f8 = F >> 3;
    float mx = 0.f;
    if (idx < M * f8) {
        const int64_t m = idx / f8;
        const int f = (int)(idx - m * f8) * 8;
        const v8 a = *(const v8*)(in + m * 2 * F + f), g = *(const v8*)(in + m * 2 * F + F + f);
        v8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float gv = (float)g[e];
            const float r = (float)a[e] * (gv * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * gv)));
            ovf_see<T>(mx, r, 0.f);
            o[e] = Half<T>::from(r);
        }
        *(v8*)(out + m * F + f) = o;
    }
    ovf_commit<T>(ovf, mx);
}

// s[n] = sum_k float(W[n][k]) of a half matrix (fp32 accumulate): the column-sum vector of a LayerNorm-folded weight
template <typename T>
__global__ __launch_bounds__(256) void row_sum_half_kernel(const T* __restrict__ w, int N, int K, float* __restrict__ out) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= N) return;
    float s = 0.f;
    for (int k = lane; k < K; k += 64) s += (float)w[(int64_t)row * K + k];
    s = wave_sum(s);
    if (lane == 0) out[row] = s;
}

// Folded LayerNorm (GemmArgs): per-row partial statistics [M][nseg][2] = (sum, sum of squares) of the 64-column segments of a raw
// tensor y, written by the epilogue that produced y  ->  one (rstd, -mu * rstd) pair per row, LN(y) = y * rstd + (-mu rstd) (times
// gamma, plus beta).  One thread per row, the segments added in order (bit-reproducible); 96 bytes read and 8 written per row.
__global__ __launch_bounds__(256) void ln_rowstats_kernel(const float* __restrict__ stats, int M, int nseg, float eps, float* __restrict__ rows) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const f32x4* src = (const f32x4*)(stats + (int64_t)m * nseg * 2);      // nseg is even (launcher)
    float s1 = 0.f, s2 = 0.f;
    for (int q = 0; q < (nseg >> 1); ++q) { const f32x4 v = src[q]; s1 += v[0] + v[2]; s2 += v[1] + v[3]; }
    const float inv = 1.0f / (float)(64 * nseg);
    const float mu = s1 * inv;
    const float var = fmaxf(__builtin_fmaf(-mu, mu, s2 * inv), 0.f);
    const float rstd = __builtin_amdgcn_rsqf(var + eps);
    ((float2*)rows)[m] = make_float2(rstd, -mu * rstd);
}

// GemmArgs::pool_part -> per-clip means: one thread per (clip, column); a clip's 64-row blocks are added in increasing order
__global__ __launch_bounds__(256) void pool_reduce_kernel(const float* __restrict__ part, int B, int T, int N, float* __restrict__ out, int64_t ldo, int mode) {
    const int n = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (n >= N) return;
    const int64_t r0 = (int64_t)b * T, r1 = r0 + T - 1;
    float acc = mode == 1 ? -__builtin_inff() : 0.f;
    for (int64_t rb = r0 >> 6; rb <= (r1 >> 6); ++rb) {
        const int64_t first_clip = (rb * 64) / T;       // the clip of the block's first row owns slot 0
        const float v = part[(rb * 2 + (first_clip == b ? 0 : 1)) * N + n];
        acc = mode == 1 ? fmaxf(acc, v) : acc + v;
    }
    out[(int64_t)b * ldo + n] = mode == 1 ? acc : acc * (1.0f / (float)T);
}

// fp32 [B, T, C] -> [B, C]: the maximum over a clip's rows (mode 2) or its first row (mode 3)
__global__ __launch_bounds__(256) void agg_pool_kernel(const float* __restrict__ in, int T, int C, int mode, float* __restrict__ out) {
    const int c = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (c >= C) return;
    const float* p = in + (int64_t)b * T * C + c;
    float v = p[0];
    if (mode == 2)
        for (int t = 1; t < T; ++t) v = fmaxf(v, p[(int64_t)t * C]);
    out[(int64_t)b * C + c] = v;
}


// out[h][r] = rel_table[lut[clamp(r - (T - 1), -maxd, maxd) + maxd]][h]: the [H, 2T - 1] Toeplitz rows of the relative position bias
// (compute_bias, backbone.py:475-492) from the resident bucket table.  `lut` holds the T5 bucket of every offset up to the distance where
// the bucket saturates, computed ON THE HOST by avexhip_rel_bucket (the function pinned bit for bit to the reference's buckets): the
// device never evaluates a logarithm, so the table equals the host-built one exactly.
__global__ __launch_bounds__(256) void bias_toeplitz_kernel(const float* __restrict__ rel_table, const int* __restrict__ lut, int maxd, int T, int H,
                                                           float* __restrict__ out) {
    const int W = 2 * T - 1;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= H * W) return;
    const int hh = idx / W, r = idx - hh * W;
    int d = r - (T - 1);
    d = d < -maxd ? -maxd : (d > maxd ? maxd : d);
    out[idx] = rel_table[(size_t)lut[d + maxd] * H + hh];
}

// column vectors of a residual-side LayerNorm fold: ga = alpha * gamma, bb = bias + alpha * beta (GemmArgs::lnr_prefolded)
__global__ __launch_bounds__(256) void lnr_fold_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ bias, float alpha,
                                                        int N, float* __restrict__ ga, float* __restrict__ bb) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    ga[n] = gamma[n] * alpha;
    bb[n] = __builtin_fmaf(alpha, beta[n], bias[n]);
}

}  // namespace

namespace avx {

int bias_toeplitz(const float* rel_table, const int* lut, int maxd, int T, int H, float* out, hipStream_t s) {
    const int n = H * (2 * T - 1);
    hipLaunchKernelGGL(bias_toeplitz_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, rel_table, lut, maxd, T, H, out);
    AVX_LAUNCH_CHECK();
    return AVEXHIP_OK;
}

int lnr_fold(const float* gamma, const float* beta, const float* bias, float alpha, int N, float* ga, float* bb, hipStream_t s) {
    AVX_REQUIRE(gamma && beta && bias && ga && bb && N > 0, "lnr_fold: bad arguments");
    hipLaunchKernelGGL(lnr_fold_kernel, dim3((N + 255) / 256), dim3(256), 0, s, gamma, beta, bias, alpha, N, ga, bb);
    AVX_LAUNCH_CHECK();
    return AVEXHIP_OK;
}

int ln_rowstats(const float* stats, int M, int nseg, float eps, float* rows, hipStream_t s) {
    AVX_REQUIRE(stats && rows && M > 0, "ln_rowstats: bad arguments");
    AVX_REQUIRE(nseg > 0 && nseg % 2 == 0, "ln_rowstats: nseg = %d (row width must be a multiple of 128)", nseg);
    hipLaunchKernelGGL(ln_rowstats_kernel, dim3((M + 255) / 256), dim3(256), 0, s, stats, M, nseg, eps, rows);
    AVX_LAUNCH_CHECK();
    return AVEXHIP_OK;
}

int row_sum_half(const void* w, int N, int K, float* out, int dtype, hipStream_t s) {
    AVX_REQUIRE(w && out && N > 0 && K > 0, "row_sum_half: bad arguments");
    if (dtype == AVEXHIP_BF16) hipLaunchKernelGGL(row_sum_half_kernel<__bf16>, dim3((N + 3) / 4), dim3(256), 0, s, (const __bf16*)w, N, K, out);
    else hipLaunchKernelGGL(row_sum_half_kernel<_Float16>, dim3((N + 3) / 4), dim3(256), 0, s, (const _Float16*)w, N, K, out);
    AVX_LAUNCH_CHECK();
    return AVEXHIP_OK;
}


int cast_to_half(const float* in, void* out, int64_t n, int dtype, hipStream_t s, unsigned int* ovf) {
    AVX_REQUIRE(in && out && n >= 0, "cast_to_half: bad arguments");
    if (n == 0) return AVEXHIP_OK;
    int64_t blocks = (n / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    if (dtype == AVEXHIP_F16)
        hipLaunchKernelGGL(cast_to_half_kernel<_Float16>, dim3((unsigned)blocks), dim3(256), 0, s, in, (_Float16*)out, n, ovf);
    else if (dtype == AVEXHIP_BF16)
        hipLaunchKernelGGL(cast_to_half_kernel<__bf16>, dim3((unsigned)blocks), dim3(256), 0, s, in, (__bf16*)out, n, ovf);
    else {
        avexhip_set_error("cast_to_half: unknown dtype %d", dtype);
        return AVEXHIP_ERR_INVALID;
    }
    AVX_LAUNCH_CHECK();
    return AVEXHIP_OK;
}

int cast_to_f32(const void* in, float* out, int64_t n, int dtype, hipStream_t s) {
    AVX_REQUIRE(in && out && n >= 0, "cast_to_f32: bad arguments");
    if (n == 0) return AVEXHIP_OK;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (dtype == AVEXHIP_F16)
        hipLaunchKernelGGL(cast_to_f32_kernel<_Float16>, dim3((unsigned)blocks), dim3(256), 0, s, (const _Float16*)in, out, n);
    else if (dtype == AVEXHIP_BF16)
        hipLaunchKernelGGL(cast_to_f32_kernel<__bf16>, dim3((unsigned)blocks), dim3(256), 0, s, (const __bf16*)in, out, n);
    else {
        avexhip_set_error("cast_to_f32: unknown dtype %d", dtype);
        return AVEXHIP_ERR_INVALID;
    }
    AVX_LAUNCH_CHECK();
    return AVEXHIP_OK;
}

int layernorm(const float* in, const void* in_half, int64_t ld_in, const float* w, const float* b, float eps, int M,
              int C, float* out_f32, int64_t ldo, void* out_half, int64_t ldh, int dtype, hipStream_t s) {
    AVX_REQUIRE((in != nullptr) != (in_half != nullptr) && w && b, "layernorm: exactly one of in / in_half, and weight/bias, must be given");
    AVX_REQUIRE(C > 0 && C % 4 == 0 && C <= 1024, "layernorm: C=%d must be a multiple of 4 and <= 1024", C);
    AVX_REQUIRE(ld_in % 4 == 0 && (!out_f32 || ldo % 4 == 0) && (!out_half || ldh % 4 == 0),
                "layernorm: leading dims must be multiples of 4");
    AVX_REQUIRE(out_f32 || out_half, "layernorm: no output");
    if (M <= 0) return AVEXHIP_OK;
    if (in_half && C % 8 == 0 && C <= 768 && ld_in % 8 == 0 && (!out_half || ldh % 8 == 0)) {
        const dim3 g8((M + 7) / 8);
        if (dtype == AVEXHIP_F16)
            hipLaunchKernelGGL(layernorm_half_kernel<_Float16>, g8, dim3(256), 0, s, (const _Float16*)in_half, ld_in, w, b, eps, M, C, out_f32, ldo, (_Float16*)out_half, ldh);
        else if (dtype == AVEXHIP_BF16)
            hipLaunchKernelGGL(layernorm_half_kernel<__bf16>, g8, dim3(256), 0, s, (const __bf16*)in_half, ld_in, w, b, eps, M, C, out_f32, ldo, (__bf16*)out_half, ldh);
        else {
            avexhip_set_error("layernorm: unknown dtype %d", dtype);
            return AVEXHIP_ERR_INVALID;
        }
        AVX_LAUNCH_CHECK();
        return AVEXHIP_OK;
    }
    const dim3 grid((M + 3) / 4);
    if (dtype == AVEXHIP_F16)
        hipLaunchKernelGGL(layernorm_kernel<_Float16>, grid, dim3(256), 0, s, in, (const _Float16*)in_half, ld_in, w, b, eps, M, C, out_f32, ldo, (_Float16*)out_half, ldh);
    else if (dtype == AVEXHIP_BF16)
        hipLaunchKernelGGL(layernorm_kernel<__bf16>, grid, dim3(256), 0, s, in, (const __bf16*)in_half, ld_in, w, b, eps, M, C, out_f32, ldo, (__bf16*)out_half, ldh);
    else {
        avexhip_set_error("layernorm: unknown dtype %d", dtype);
        return AVEXHIP_ERR_INVALID;
    }
    AVX_LAUNCH_CHECK();
    return AVEXHIP_OK;
}

int mean_pool(const float* in, int B, int T, int C, const uint8_t* frame_pad, float* out, hipStream_t s) {
    AVX_REQUIRE(in && out && B > 0 && T > 0 && C > 0, "mean_pool: bad arguments");
    hipLaunchKernelGGL(mean_pool_kernel, dim3((C + 63) / 64, B), dim3(256), 0, s, in, T, C, frame_pad, out);
    AVX_LAUNCH_CHECK();
    return AVEXHIP_OK;
}

// LayerNorm over C (C % 8 == 0, C <= 768) of half rows [B*T, C] followed by the mean over each clip's T rows -> out [B, C] fp32
int layernorm_pool(const void* in_half, int64_t ld_in, const float* w, const float* b, float eps, int B, int T, int C, float* out, int dtype,
                   hipStream_t s) {
    AVX_REQUIRE(in_half && w && b && out && B > 0 && T > 0, "layernorm_pool: bad arguments");
    AVX_REQUIRE(C % 8 == 0 && C <= 768 && ld_in % 8 == 0, "layernorm_pool: C = %d (need a multiple of 8, <= 768)", C);
    if (dtype == AVEXHIP_F16)
        hipLaunchKernelGGL(layernorm_pool_kernel<_Float16>, dim3(B), dim3(1024), 0, s, (const _Float16*)in_half, ld_in, w, b, eps, T, C, out);
    else if (dtype == AVEXHIP_BF16)
        hipLaunchKernelGGL(layernorm_pool_kernel<__bf16>, dim3(B), dim3(1024), 0, s, (const __bf16*)in_half, ld_in, w, b, eps, T, C, out);
    else {
        avexhip_set_error("layernorm_pool: unknown dtype %d", dtype);
        return AVEXHIP_ERR_INVALID;
    }
    AVX_LAUNCH_CHECK();
    return AVEXHIP_OK;
}

int token_embed_ln(const void* patches, const float* pos, const float* cls, const float* w, const float* b, float eps, int B, int Tp, int C,
                   void* out_half, float* out_f32, int dtype, hipStream_t s) {
    AVX_REQUIRE(patches && pos && cls && w && b && (out_half || out_f32), "token_embed_ln: null argument");
    AVX_REQUIRE(B > 0 && Tp > 0 && C % 64 == 0 && C <= 1024, "token_embed_ln: bad shape B=%d patches=%d C=%d", B, Tp, C);
    const int64_t rows = (int64_t)B * (Tp + 1);
    const dim3 grid((unsigned)((rows + 3) / 4));
    if (dtype == AVEXHIP_F16)
        hipLaunchKernelGGL(token_embed_ln_kernel<_Float16>, grid, dim3(256), 0, s, (const _Float16*)patches, pos, cls, w, b, eps, Tp, C, rows, (_Float16*)out_half, out_f32);
    else if (dtype == AVEXHIP_BF16)
        hipLaunchKernelGGL(token_embed_ln_kernel<__bf16>, grid, dim3(256), 0, s, (const __bf16*)patches, pos, cls, w, b, eps, Tp, C, rows, (__bf16*)out_half, out_f32);
    else {
        avexhip_set_error("token_embed_ln: unknown dtype %d", dtype);
        return AVEXHIP_ERR_INVALID;
    }
    AVX_LAUNCH_CHECK();
    return AVEXHIP_OK;
}

int pool_reduce(const float* part, int B, int T, int N, float* out, int64_t ldo, hipStream_t s, int mode) {
    AVX_REQUIRE(part && out && B > 0 && T >= 64 && N > 0 && (mode == 0 || mode == 1), "pool_reduce: bad arguments (B=%d T=%d N=%d mode=%d)", B, T, N, mode);
    hipLaunchKernelGGL(pool_reduce_kernel, dim3((N + 255) / 256, B), dim3(256), 0, s, part, B, T, N, out, ldo, mode);
    AVX_LAUNCH_CHECK();
    return AVEXHIP_OK;
}

int agg_pool(const float* in, int B, int T, int C, int mode, float* out, hipStream_t s) {
    if (mode == 1) return mean_pool(in, B, T, C, nullptr, out, s);
    AVX_REQUIRE(in && out && B > 0 && T > 0 && C > 0 && (mode == 2 || mode == 3), "agg_pool: bad arguments (mode %d)", mode);
    hipLaunchKernelGGL(agg_pool_kernel, dim3((C + 255) / 256, B), dim3(256), 0, s, in, T, C, mode, out);
    AVX_LAUNCH_CHECK();
    return AVEXHIP_OK;
}

int glu_swish(const void* in, int64_t M, int F, void* out, unsigned int* ovf, int dtype, hipStream_t s) {
    AVX_REQUIRE(in && out && M > 0 && F > 0 && F % 8 == 0, "glu_swish: bad arguments (M=%lld F=%d)", (long long)M, F);
    const dim3 grid((unsigned)((M * (F / 8) + 255) / 256));
    if (dtype == AVEXHIP_F16) hipLaunchKernelGGL(glu_swish_kernel<_Float16>, grid, dim3(256), 0, s, (const _Float16*)in, M, F, (_Float16*)out, ovf);
    else if (dtype == AVEXHIP_BF16) hipLaunchKernelGGL(glu_swish_kernel<__bf16>, grid, dim3(256), 0, s, (const __bf16*)in, M, F, (__bf16*)out, ovf);
    else {
        avexhip_set_error("glu_swish: unknown dtype %d", dtype);
        return AVEXHIP_ERR_INVALID;
    }
    AVX_LAUNCH_CHECK();
    return AVEXHIP_OK;
}

int zero_rows(float* x32, int64_t ld32, void* x_half, int64_t ldh, int M, int C, const uint8_t* pad, hipStream_t s) {
    if (!pad || M <= 0 || (!x32 && !x_half)) return AVEXHIP_OK;
    hipLaunchKernelGGL(zero_rows_kernel, dim3(M), dim3(256), 0, s, x32, (unsigned short*)x_half, ld32, ldh, C, pad);
    AVX_LAUNCH_CHECK();
    return AVEXHIP_OK;
}

int patchify(const float* fbank, int B, int frames, int n_mels, int patch, void* out_patch, int dtype,
             hipStream_t s) {
    AVX_REQUIRE(fbank && out_patch && patch > 0, "patchify: bad arguments");
    const int nt = frames / patch, nf = n_mels / patch;
    const int64_t total = (int64_t)B * nt * nf * patch * patch;
    if (total == 0) return AVEXHIP_OK;
    const dim3 grid((unsigned)((total + 255) / 256));
    if (dtype == AVEXHIP_F16)
        hipLaunchKernelGGL(patchify_kernel<_Float16>, grid, dim3(256), 0, s, fbank, frames, n_mels, patch, nt, nf, (_Float16*)out_patch, total);
    else if (dtype == AVEXHIP_BF16)
        hipLaunchKernelGGL(patchify_kernel<__bf16>, grid, dim3(256), 0, s, fbank, frames, n_mels, patch, nt, nf, (__bf16*)out_patch, total);
    else {
        avexhip_set_error("patchify: unknown dtype %d", dtype);
        return AVEXHIP_ERR_INVALID;
    }
    AVX_LAUNCH_CHECK();
    return AVEXHIP_OK;
}

}  // namespace avx

#ifdef AVEX_DIAG
extern "C" int avexhip_debug_lds_canary(int blocks, int iters, unsigned* report_dev, void* stream) {
    hipLaunchKernelGGL(lds_canary_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, report_dev);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
#endif
