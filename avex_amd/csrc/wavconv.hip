// First layer of the wav2vec2 / AVES convolutional feature extractor on gfx950.
//
// Restates torchaudio.models.wav2vec2 ConvLayerBlock 0 as the reference instantiates it (avex/models/aves_model.py:25-33,86:
// extractor_mode "group_norm", conv_bias False): Conv1d(1, 512, kernel 10, stride 5) -> GroupNorm(512 groups, 512 channels,
// affine, eps 1e-5: per clip and channel over time) -> GELU (erf).  One input channel and 10 taps is no MFMA shape, and the
// normalisation needs every frame of the clip before any output: two passes over the waveform (640 KB per clip, L2-resident)
// that both compute the convolution in fp32 -- the first only writes each workgroup's partial sum / sum of squares per (clip,
// channel) (no atomics: the second pass adds the partials in a fixed order, so results are bit-reproducible), the second
// normalises, applies GELU and writes the operand-type activations [clip][frame][channel] that the following conv layers
// consume as strided-row GEMMs.  Nothing of size frames x channels is ever written in fp32.
#include "common.h"

namespace {

constexpr int WC_K = 10, WC_S = 5, WC_C = 512, WC_FR = 1024;   // taps, stride, channels, frames per workgroup

template <typename T, bool APPLY>
__global__ __launch_bounds__(256) void wavconv0_kernel(const float* __restrict__ wav, int64_t stride, int frames,
                                                       const float* __restrict__ w, float* __restrict__ stats,
                                                       const float* __restrict__ gn_w, const float* __restrict__ gn_b, float eps,
                                                       T* __restrict__ out, int frames_pad) {
    __shared__ float xs[WC_FR * WC_S + WC_K];
    const int b = blockIdx.y, f0 = blockIdx.x * WC_FR, tid = threadIdx.x;
    const int nf = (frames - f0) < WC_FR ? (frames - f0) : WC_FR;          // valid frames of this block (may be <= 0 in the padding)
    const float* src = wav + (int64_t)b * stride + (int64_t)f0 * WC_S;
    const int nsamp = nf > 0 ? nf * WC_S + (WC_K - WC_S) : 0;
    for (int i = tid; i < nsamp; i += 256) xs[i] = src[i];
    __syncthreads();
    const int c0 = 2 * tid;                                               // two adjacent channels per thread
    float w0[WC_K], w1[WC_K];
#pragma unroll
    for (int j = 0; j < WC_K; ++j) { w0[j] = w[c0 * WC_K + j]; w1[j] = w[(c0 + 1) * WC_K + j]; }
    float sc0 = 1.f, sh0 = 0.f, sc1 = 1.f, sh1 = 0.f;
    const int nblk = (frames + WC_FR - 1) / WC_FR;                          // partial statistics: [clip][block][channel][2]
    if (APPLY) {
        const float inv = 1.0f / (float)frames;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < nblk; ++k) acc += *(const f32x4*)(stats + (((int64_t)b * nblk + k) * WC_C + c0) * 2);
        const float m0 = acc[0] * inv, m1 = acc[2] * inv;
        const float v0 = fmaxf(__builtin_fmaf(-m0, m0, acc[1] * inv), 0.f), v1 = fmaxf(__builtin_fmaf(-m1, m1, acc[3] * inv), 0.f);
        sc0 = gn_w[c0] / sqrtf(v0 + eps); sh0 = __builtin_fmaf(-m0, sc0, gn_b[c0]);
        sc1 = gn_w[c0 + 1] / sqrtf(v1 + eps); sh1 = __builtin_fmaf(-m1, sc1, gn_b[c0 + 1]);
    }
    float s1a = 0.f, s2a = 0.f, s1b = 0.f, s2b = 0.f;
    typedef typename Half<T>::v4 v4;
    T* orow = APPLY ? out + ((int64_t)b * frames_pad + f0) * WC_C + c0 : nullptr;
    for (int t = 0; t < nf; ++t) {
        const float* x = xs + t * WC_S;
        float y0 = 0.f, y1 = 0.f;
#pragma unroll
        for (int j = 0; j < WC_K; ++j) { y0 = __builtin_fmaf(w0[j], x[j], y0); y1 = __builtin_fmaf(w1[j], x[j], y1); }
        if (APPLY) {
            const f32x2 g = gelu_erf2((f32x2){__builtin_fmaf(y0, sc0, sh0), __builtin_fmaf(y1, sc1, sh1)});
            typedef T v2 __attribute__((ext_vector_type(2)));
            v2 h; h[0] = Half<T>::from(g[0]); h[1] = Half<T>::from(g[1]);
            *(v2*)(orow + (int64_t)t * WC_C) = h;
        } else {
            s1a += y0; s2a = __builtin_fmaf(y0, y0, s2a); s1b += y1; s2b = __builtin_fmaf(y1, y1, s2b);
        }
    }
    if (APPLY) {
        // rows of the padded layout past the last frame: zeros (never read by a valid output of the next layer)
        const int tend = (frames_pad - f0) < WC_FR ? (frames_pad - f0) : WC_FR;
        typedef T v2 __attribute__((ext_vector_type(2)));
        v2 z; z[0] = (T)0.0f; z[1] = (T)0.0f;
        for (int t = nf > 0 ? nf : 0; t < tend; ++t) *(v2*)(orow + (int64_t)t * WC_C) = z;
    } else {
        *(f32x4*)(stats + (((int64_t)b * nblk + blockIdx.x) * WC_C + c0) * 2) = (f32x4){s1a, s2a, s1b, s2b};
    }
}

}  // namespace

// frames of the first layer for T samples: (T - 10) / 5 + 1
extern "C" int avexhip_wavconv0_frames(int64_t T) { return T < WC_K ? 0 : (int)((T - WC_K) / WC_S + 1); }

// floats of scratch avexhip_wavconv0 needs for its partial statistics: B * ceil(frames / 1024) * 512 * 2
extern "C" int64_t avexhip_wavconv0_stats_floats(int B, int64_t T) {
    const int frames = avexhip_wavconv0_frames(T);
    return (int64_t)(B > 0 ? B : 0) * ((frames + WC_FR - 1) / WC_FR) * WC_C * 2;
}

extern "C" int avexhip_wavconv0(const float* wav_dev, int B, int64_t T, int64_t wav_stride, const float* w_dev,
                                const float* gn_w_dev, const float* gn_b_dev, float eps, float* stats_dev, void* out_dev,
                                int frames_pad, int dtype, void* stream) {
    AVX_REQUIRE(wav_dev && w_dev && gn_w_dev && gn_b_dev && stats_dev && out_dev, "wavconv0: null argument");
    const int frames = avexhip_wavconv0_frames(T);
    AVX_REQUIRE(B > 0 && frames > 0, "wavconv0: empty input (B=%d, T=%lld)", B, (long long)T);
    AVX_REQUIRE(frames_pad >= frames, "wavconv0: frames_pad=%d < frames=%d", frames_pad, frames);
    if (wav_stride <= 0) wav_stride = T;
    hipStream_t s = (hipStream_t)stream;
    const dim3 g1((frames + WC_FR - 1) / WC_FR, B), g2((frames_pad + WC_FR - 1) / WC_FR, B);
    if (dtype == AVEXHIP_BF16) {
        hipLaunchKernelGGL((wavconv0_kernel<__bf16, false>), g1, dim3(256), 0, s, wav_dev, wav_stride, frames, w_dev, stats_dev, gn_w_dev, gn_b_dev, eps, (__bf16*)nullptr, frames_pad);
        hipLaunchKernelGGL((wavconv0_kernel<__bf16, true>), g2, dim3(256), 0, s, wav_dev, wav_stride, frames, w_dev, stats_dev, gn_w_dev, gn_b_dev, eps, (__bf16*)out_dev, frames_pad);
    } else if (dtype == AVEXHIP_F16) {
        hipLaunchKernelGGL((wavconv0_kernel<_Float16, false>), g1, dim3(256), 0, s, wav_dev, wav_stride, frames, w_dev, stats_dev, gn_w_dev, gn_b_dev, eps, (_Float16*)nullptr, frames_pad);
        hipLaunchKernelGGL((wavconv0_kernel<_Float16, true>), g2, dim3(256), 0, s, wav_dev, wav_stride, frames, w_dev, stats_dev, gn_w_dev, gn_b_dev, eps, (_Float16*)out_dev, frames_pad);
    } else {
        avexhip_set_error("wavconv0: unknown dtype %d", dtype);
        return AVEXHIP_ERR_INVALID;
    }
    AVX_LAUNCH_CHECK();
    return AVEXHIP_OK;
}
