// First layer of the wav2vec2 / AVES convolutional feature extractor on gfx950.
//
// Restates torchaudio.models.wav2vec2 ConvLayerBlock 0 as the reference instantiates it (avex/models/aves_model.py:25-33,86:
// extractor_mode "group_norm", conv_bias False): Conv1d(1, 512, kernel 10, stride 5) -> GroupNorm(512 groups, 512 channels,
// affine, eps 1e-5: per clip and channel over time) -> GELU (erf).  One input channel and 10 taps is no MFMA shape, and the
// normalisation needs every frame of the clip before any output.
//
// Round 5: the statistics no longer cost a pass over frames x channels.  The layer is LINEAR in the waveform, so a channel's sum
// and sum of squares over time follow from 65 moments of the clip that do not depend on the channel:
//     y_c[t] = sum_j w_c[j] x[5 t + j]      sum_t y_c = sum_j w_c[j] S_j            S_j  = sum_t x[5 t + j]             (10)
//                                           sum_t y_c^2 = sum_jk w_c[j] w_c[k] R_jk   R_jk = sum_t x[5 t + j] x[5 t + k]  (55, j <= k)
// wavconv0_moments_kernel accumulates S and R per 1 024-frame block in fp64 (the products of two floats are exact in a double; a
// filter that cancels most of its input -- a high-pass on a clip with a DC offset -- makes w' R w a difference of large numbers, which
// fp32 sums would not survive), wavconv0_finish_kernel adds the blocks in a fixed order and turns them into each channel's
// (scale, shift) = (g / sqrt(var + eps), b - mean scale), and the one pass that is left computes the convolution, normalises,
// applies GELU and writes the operand-type activations [clip][frame][channel] that the following conv layers consume as
// strided-row GEMMs.  Before: two passes of 20 multiply-adds per (frame, channel pair), 2.2 ms per 128 clips; the first is gone.
// Bit-reproducible (no atomics, fixed orders); nothing of size frames x channels is ever written in fp32.
#include "common.h"

namespace {

constexpr int WC_K = 10, WC_S = 5, WC_C = 512, WC_FR = 1024;   // taps, stride, channels, frames per workgroup
constexpr int WC_NM = WC_K + WC_K * (WC_K + 1) / 2;            // 65 moments per clip

// moments of one 1 024-frame block of one clip: part[(clip * nblk + block) * 65 + i]
__global__ __launch_bounds__(256) void wavconv0_moments_kernel(const float* __restrict__ wav, int64_t stride, int frames, double* __restrict__ part) {
    __shared__ float xs[WC_FR * WC_S + WC_K];
    __shared__ double red[4][WC_NM];
    const int b = blockIdx.y, f0 = blockIdx.x * WC_FR, tid = threadIdx.x;
    const int nf = (frames - f0) < WC_FR ? (frames - f0) : WC_FR;
    const float* src = wav + (int64_t)b * stride + (int64_t)f0 * WC_S;
    const int nsamp = nf > 0 ? nf * WC_S + (WC_K - WC_S) : 0;
    for (int i = tid; i < nsamp; i += 256) xs[i] = src[i];
    __syncthreads();
    double m[WC_NM];
#pragma unroll
    for (int i = 0; i < WC_NM; ++i) m[i] = 0.0;
    for (int t = tid; t < nf; t += 256) {
        double x[WC_K];
#pragma unroll
        for (int j = 0; j < WC_K; ++j) x[j] = (double)xs[t * WC_S + j];
        int idx = WC_K;
#pragma unroll
        for (int j = 0; j < WC_K; ++j) {
            m[j] += x[j];
#pragma unroll
            for (int k = j; k < WC_K; ++k) { m[idx] = __builtin_fma(x[j], x[k], m[idx]); ++idx; }
        }
    }
    // wave sums (xor butterflies: every lane ends with the same value), then the four waves in a fixed order
#pragma unroll
    for (int i = 0; i < WC_NM; ++i) {
        double v = m[i];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
        m[i] = v;
    }
    if ((tid & 63) == 0) {
#pragma unroll
        for (int i = 0; i < WC_NM; ++i) red[tid >> 6][i] = m[i];
    }
    __syncthreads();
    if (tid < WC_NM) part[((int64_t)b * gridDim.x + blockIdx.x) * WC_NM + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}

// one workgroup per clip, one thread per channel: (scale, shift) of the GroupNorm from the clip's moments
__global__ __launch_bounds__(WC_C) void wavconv0_finish_kernel(const double* __restrict__ part, int nblk, int frames, const float* __restrict__ w,
                                                               const float* __restrict__ gn_w, const float* __restrict__ gn_b, float eps,
                                                               float* __restrict__ scsh) {
    __shared__ double mom[WC_NM];
    const int b = blockIdx.x, c = threadIdx.x;
    if (c < WC_NM) {
        double s = 0.0;
        for (int k = 0; k < nblk; ++k) s += part[((int64_t)b * nblk + k) * WC_NM + c];
        mom[c] = s;
    }
    __syncthreads();
    double wc[WC_K];
#pragma unroll
    for (int j = 0; j < WC_K; ++j) wc[j] = (double)w[c * WC_K + j];
    double s1 = 0.0, s2 = 0.0;
    int idx = WC_K;
#pragma unroll
    for (int j = 0; j < WC_K; ++j) {
        s1 = __builtin_fma(wc[j], mom[j], s1);
#pragma unroll
        for (int k = j; k < WC_K; ++k) { s2 = __builtin_fma((j == k ? 1.0 : 2.0) * wc[j] * wc[k], mom[idx], s2); ++idx; }
    }
    const double inv = 1.0 / (double)frames;
    const double mean = s1 * inv;
    const double var = __builtin_fmax(s2 * inv - mean * mean, 0.0);      // biased, as GroupNorm
    const float m0 = (float)mean, v0 = (float)var;
    const float sc = gn_w[c] / sqrtf(v0 + eps);
    *(float2*)(scsh + ((int64_t)b * WC_C + c) * 2) = make_float2(sc, __builtin_fmaf(-m0, sc, gn_b[c]));
}

template <typename T>
__global__ __launch_bounds__(256) void wavconv0_kernel(const float* __restrict__ wav, int64_t stride, int frames,
                                                       const float* __restrict__ w, const float* __restrict__ scsh,
                                                       T* __restrict__ out, int frames_pad) {
    __shared__ float xs[WC_FR * WC_S + WC_K];
    const int b = blockIdx.y, f0 = blockIdx.x * WC_FR, tid = threadIdx.x;
    const int nf = (frames - f0) < WC_FR ? (frames - f0) : WC_FR;          // valid frames of this block (may be <= 0 in the padding)
    const float* src = wav + (int64_t)b * stride + (int64_t)f0 * WC_S;
    const int nsamp = nf > 0 ? nf * WC_S + (WC_K - WC_S) : 0;
    for (int i = tid; i < nsamp; i += 256) xs[i] = src[i];
    __syncthreads();
    const int c0 = 2 * tid;                                               // two adjacent channels per thread
    // packed multiply-adds over PAIRS OF TAPS (even taps in the low half, odd taps in the high half, the halves added at the end): the
    // sample pairs come out of LDS as register pairs and nothing has to be broadcast -- pairing the two CHANNELS instead needs x[j] in both
    // halves, and for the x that sits in the odd register of its pair that is v_pk_fma_f32 ... op_sel:[0,1,0], the form that reads wrong
    // values beside MFMA work on gfx950 (avex_amd/isa_lint.py)
    f32x2 wa[WC_K / 2], wb[WC_K / 2];
#pragma unroll
    for (int q = 0; q < WC_K / 2; ++q) {
        wa[q] = (f32x2){w[c0 * WC_K + 2 * q], w[c0 * WC_K + 2 * q + 1]};
        wb[q] = (f32x2){w[(c0 + 1) * WC_K + 2 * q], w[(c0 + 1) * WC_K + 2 * q + 1]};
    }
    const f32x4 ss = *(const f32x4*)(scsh + ((int64_t)b * WC_C + c0) * 2);
    const f32x2 sc = {ss[0], ss[2]}, sh = {ss[1], ss[3]};
    T* orow = out + ((int64_t)b * frames_pad + f0) * WC_C + c0;
    typedef T v2 __attribute__((ext_vector_type(2)));
    AVX_F16_SATURATE_ON();                                                // no MFMA in this kernel: the conversions saturate in hardware (NaN and inf pass)
    AVX_CLAMP_TOKEN(inva);
    T* op = orow;
    for (int t = 0; t < nf; ++t, op += WC_C) {
        const float* x = xs + t * WC_S;
        f32x2 ya = {0.f, 0.f}, yb = {0.f, 0.f};
#pragma unroll
        for (int q = 0; q < WC_K / 2; ++q) {
            const f32x2 xp = {x[2 * q], x[2 * q + 1]};
            ya = __builtin_elementwise_fma(wa[q], xp, ya);
            yb = __builtin_elementwise_fma(wb[q], xp, yb);
        }
        const f32x2 y = {ya[0] + ya[1], yb[0] + yb[1]};
        const f32x2 g = gelu_erf2_h(__builtin_elementwise_fma(y, sc, sh), inva);      // the output is rounded to the operand type: the GELU sized for it
        v2 h; h[0] = Half<T>::from_hw(g[0]); h[1] = Half<T>::from_hw(g[1]);
        *(v2*)op = h;
    }
    // rows of the padded layout past the last frame: zeros (never read by a valid output of the next layer)
    const int tend = (frames_pad - f0) < WC_FR ? (frames_pad - f0) : WC_FR;
    v2 z; z[0] = (T)0.0f; z[1] = (T)0.0f;
    for (int t = nf > 0 ? nf : 0; t < tend; ++t) *(v2*)(orow + (int64_t)t * WC_C) = z;
}

}  // namespace

// frames of the first layer for T samples: (T - 10) / 5 + 1
extern "C" int avexhip_wavconv0_frames(int64_t T) { return T < WC_K ? 0 : (int)((T - WC_K) / WC_S + 1); }

// floats of scratch avexhip_wavconv0 needs: the (scale, shift) pairs [B][512][2] and, behind them, the 65 fp64 moments of every 1 024-frame block
extern "C" int64_t avexhip_wavconv0_stats_floats(int B, int64_t T) {
    const int frames = avexhip_wavconv0_frames(T);
    const int64_t nb = B > 0 ? B : 0;
    return nb * WC_C * 2 + nb * ((frames + WC_FR - 1) / WC_FR) * WC_NM * 2;
}

extern "C" int avexhip_wavconv0(const float* wav_dev, int B, int64_t T, int64_t wav_stride, const float* w_dev,
                                const float* gn_w_dev, const float* gn_b_dev, float eps, float* stats_dev, void* out_dev,
                                int frames_pad, int dtype, void* stream) {
    AVX_REQUIRE(wav_dev && w_dev && gn_w_dev && gn_b_dev && stats_dev && out_dev, "wavconv0: null argument");
    const int frames = avexhip_wavconv0_frames(T);
    AVX_REQUIRE(B > 0 && frames > 0, "wavconv0: empty input (B=%d, T=%lld)", B, (long long)T);
    AVX_REQUIRE(frames_pad >= frames, "wavconv0: frames_pad=%d < frames=%d", frames_pad, frames);
    AVX_REQUIRE(dtype == AVEXHIP_BF16 || dtype == AVEXHIP_F16, "wavconv0: unknown dtype %d", dtype);
    AVX_REQUIRE(((uintptr_t)stats_dev & 7) == 0, "wavconv0: stats_dev must be 8-byte aligned (it holds fp64 moments)");
    if (wav_stride <= 0) wav_stride = T;
    hipStream_t s = (hipStream_t)stream;
    const int nblk = (frames + WC_FR - 1) / WC_FR;
    const dim3 g1(nblk, B), g2((frames_pad + WC_FR - 1) / WC_FR, B);
    float* scsh = stats_dev;
    double* part = (double*)(stats_dev + (int64_t)B * WC_C * 2);
    hipLaunchKernelGGL(wavconv0_moments_kernel, g1, dim3(256), 0, s, wav_dev, wav_stride, frames, part);
    hipLaunchKernelGGL(wavconv0_finish_kernel, dim3(B), dim3(WC_C), 0, s, (const double*)part, nblk, frames, w_dev, gn_w_dev, gn_b_dev, eps, scsh);
    if (dtype == AVEXHIP_BF16)
        hipLaunchKernelGGL((wavconv0_kernel<__bf16>), g2, dim3(256), 0, s, wav_dev, wav_stride, frames, w_dev, (const float*)scsh, (__bf16*)out_dev, frames_pad);
    else
        hipLaunchKernelGGL((wavconv0_kernel<_Float16>), g2, dim3(256), 0, s, wav_dev, wav_stride, frames, w_dev, (const float*)scsh, (_Float16*)out_dev, frames_pad);
    AVX_LAUNCH_CHECK();
    return AVEXHIP_OK;
}
