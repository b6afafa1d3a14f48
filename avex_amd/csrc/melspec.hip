// STFT power / mel spectrogram frontend (the reference's AudioProcessor for "spectrogram" / "mel_spectrogram",
// avex/data/audio_utils.py:77-172; EfficientNet config: n_fft 800, hop 160, hann, 128 mels, center) on gfx950.
//
// n_fft = 800 is not a power of two, and the path is one pass over the waveform either way, so the transform is done as a
// DENSE fp32 product on the fp32 MFMA (v_mfma_f32_32x32x2f32: exact fp32 FMA chains, 1/16 of the f16 rate): 32 frames x
// [n_fft samples] times a precomputed [n_fft, 2 * HALF] matrix (window folded in; columns 0..n_freq-1 real part, HALF.. the
// imaginary part).  One 256-thread workgroup per 32 frames: the (reflect-padded) samples of those frames sit in LDS with a
// one-word skew per hop so that the 32 lanes reading sample n of 32 different frames hit 32 banks; the four waves own the
// real/imaginary column halves; |X|^2 is combined through LDS, the triangular mel bank is applied as CSR rows, log(x + 1e-6) is
// written [clip][bin][frame] (128-byte rows) and a per-clip min / max is kept with ordered-integer atomics for the second,
// trivial pass that applies (x - min) / (max - min + 1e-8) (audio_utils.py:166-172).
#include <math.h>

#include <vector>

#include "common.h"

namespace {

struct MelDev {
    int n_fft, hop, n_freq, half, nc, n_out, center, skew, use_mel, kfold, fold;
    const float* dft;        // [kfold][nc], kfold = n_fft/2 + 1 rounded up to even: the folded contraction (see the kernel)
    const int* mel_start;    // [n_out] CSR over frequency bins (mel) -- unused for plain spectrograms
    const int* mel_len;
    const int* mel_off;
    const float* mel_w;
};

__device__ __forceinline__ int ord_key(float v) {            // monotonic float -> int
    const int b = __builtin_bit_cast(int, v);
    return b >= 0 ? b : b ^ 0x7fffffff;
}
__device__ __forceinline__ float ord_val(int k) { return __builtin_bit_cast(float, k >= 0 ? k : k ^ 0x7fffffff); }

template <int NTW>   // 32-column tiles per wave (HALF = 64 * NTW)
__global__ __launch_bounds__(256) void melspec_kernel(MelDev md, const float* __restrict__ wav, int64_t T, int64_t stride, int frames,
                                                      float* __restrict__ out, int* __restrict__ minmax, int take_log) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, f0 = blockIdx.x * 32;
    const int nseg = md.hop * 31 + md.n_fft;                      // samples covered by the block's 32 frames
    float* xs = (float*)smem;                                      // skewed: phys(i) = i + (skew ? i / hop : 0)
    const int xs_words = nseg + (md.skew ? nseg / md.hop + 1 : 0);
    const int pw_ld = md.half + 1;
    float* pw = xs + ((xs_words + 3) & ~3);                        // [32][half + 1]
    const float* src = wav + (int64_t)b * stride;
    const int64_t start = (int64_t)f0 * md.hop - (md.center ? md.n_fft / 2 : 0);
    for (int i = tid; i < nseg; i += 256) {
        int64_t j = start + i;
        if (j < 0) j = -j;                                        // reflect padding of torch.stft(center=True)
        if (j >= T) j = 2 * (T - 1) - j;
        if (j < 0) j = 0;
        xs[i + (md.skew ? i / md.hop : 0)] = (j < T) ? src[j] : 0.f;
    }
    __syncthreads();
    const int fr = lane & 31, kh = lane >> 5;
    const int halfsel = wave >> 1, wpair = wave & 1;
    f32x16 acc[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const float* bcol = md.dft + halfsel * md.half + wpair * NTW * 32 + fr;
    const int abase = md.hop * fr;
    // Real input and a window symmetric about n_fft / 2 (periodic Hann / Hamming are): fold sample n with sample N - n,
    //   Re X[k] = sum_{n=0}^{N/2} w[n] (x[n] + x[N-n]) cos(2 pi k n / N),   Im X[k] = -sum w[n] (x[n] - x[N-n]) sin(2 pi k n / N)
    // (n = 0 and n = N/2 have no partner), which halves the contraction length.  The real-part waves form the sums, the
    // imaginary-part waves the differences, on the fly from the same LDS samples.
    const int nhalf = md.n_fft / 2;
    const float sgn = halfsel ? -1.f : 1.f;
    for (int ks = 0; ks < md.kfold / 2; ++ks) {
        const int n = 2 * ks + kh;
        float a = 0.f;
        if (!md.fold) {                                   // asymmetric window: plain contraction over all n_fft samples
            const int ai = abase + n;
            a = xs[ai + (md.skew ? ai / md.hop : 0)];
        } else if (n <= nhalf) {
            const int ai = abase + n;
            a = xs[ai + (md.skew ? ai / md.hop : 0)];
            if (n > 0 && n < nhalf) {
                const int aj = abase + md.n_fft - n;
                a = __builtin_fmaf(sgn, xs[aj + (md.skew ? aj / md.hop : 0)], a);
            }
        }
        const float* brow = bcol + (int64_t)n * md.nc;
#pragma unroll
        for (int t = 0; t < NTW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, brow[t * 32], acc[t], 0, 0, 0);
    }
    // |X|^2: real halves write, imaginary halves add
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
        if (halfsel == ph) {
#pragma unroll
            for (int t = 0; t < NTW; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int f = 8 * (r >> 2) + 4 * kh + (r & 3);
                    const int col = (wpair * NTW + t) * 32 + fr;
                    float* d = pw + f * pw_ld + col;
                    const float v = acc[t][r] * acc[t][r];
                    *d = ph == 0 ? v : *d + v;
                }
        }
        __syncthreads();
    }
    // mel (CSR) or plain bins, log, store [clip][bin][frame]; per-clip min / max
    const int f = tid & 31, grp = tid >> 5;
    float mn = __builtin_inff(), mx = -__builtin_inff();
    const bool fvalid = f0 + f < frames;
    for (int m = grp; m < md.n_out; m += 8) {
        float e;
        if (md.use_mel) {
            const int st = md.mel_start[m], len = md.mel_len[m], off = md.mel_off[m];
            e = 0.f;
            for (int q = 0; q < len; ++q) e = __builtin_fmaf(pw[f * pw_ld + st + q], md.mel_w[off + q], e);
        } else {
            e = pw[f * pw_ld + m];
        }
        const float y = take_log ? logf(e + 1e-6f) : e;
        if (fvalid) {
            out[((int64_t)b * md.n_out + m) * frames + f0 + f] = y;
            mn = fminf(mn, y); mx = fmaxf(mx, y);
        }
    }
    if (minmax) {
        mn = -wave_max(-mn); mx = wave_max(mx);
        if (lane == 0 && mx >= mn) { atomicMin(minmax + 2 * b, ord_key(mn)); atomicMax(minmax + 2 * b + 1, ord_key(mx)); }
    }
}

__global__ __launch_bounds__(256) void melspec_minmax_init(int* mm, int B) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < B) { mm[2 * i] = 0x7fffffff; mm[2 * i + 1] = (int)0x80000000; }
}

__global__ __launch_bounds__(256) void melspec_norm_kernel(float* __restrict__ x, int64_t per_clip, const int* __restrict__ mm) {
    const int b = blockIdx.y;
    const float mn = ord_val(mm[2 * b]), mx = ord_val(mm[2 * b + 1]);
    const float inv = 1.0f / (mx - mn + 1e-8f);
    float* p = x + (int64_t)b * per_clip;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per_clip; i += (int64_t)gridDim.x * 256) p[i] = (p[i] - mn) / (mx - mn + 1e-8f);
    (void)inv;
}

}  // namespace

struct avexhip_melspec_plan {
    avexhip_melspec_config cfg;
    MelDev dev;
    void* blob = nullptr;
    size_t lds = 0;
};

extern "C" avexhip_melspec_plan* avexhip_melspec_plan_create(const avexhip_melspec_config* cfg, const float* window, const float* mel_fb) {
    if (!cfg || !window) { avexhip_set_error("melspec_plan_create: null argument"); return nullptr; }
    const int N = cfg->n_fft, win = cfg->win_length, hop = cfg->hop_length, nf = N / 2 + 1;
    if (N < 64 || N > 1024 || (N & 1) || win <= 0 || win > N || hop <= 0 || cfg->n_mels < 0 || cfg->n_mels > 512) {
        avexhip_set_error("melspec_plan_create: unsupported n_fft=%d win=%d hop=%d n_mels=%d (even n_fft in 64..1024, win <= n_fft)", N, win, hop, cfg->n_mels);
        return nullptr;
    }
    if (cfg->n_mels > 0 && !mel_fb) { avexhip_set_error("melspec_plan_create: mel_fb missing"); return nullptr; }
    const int half = ((nf + 63) / 64) * 64, nc = 2 * half, ntw = half / 64;
    if (ntw != 3 && ntw != 4 && ntw != 5 && ntw != 7 && ntw != 9) {
        avexhip_set_error("melspec_plan_create: n_fft=%d (column tiling %d) is not instantiated (n_fft 256..384, 400..512, 514..640, 770..896, 1026.. are)", N, ntw);
        return nullptr;
    }
    std::vector<float> hwin(win), hmel;
    if (hipMemcpy(hwin.data(), window, sizeof(float) * win, hipMemcpyDefault) != hipSuccess) { avexhip_set_error("melspec_plan_create: cannot read window"); return nullptr; }
    const int n_out = cfg->n_mels > 0 ? cfg->n_mels : nf;
    std::vector<int> st(n_out, 0), ln(n_out, 0), of(n_out, 0);
    std::vector<float> packed;
    if (cfg->n_mels > 0) {
        hmel.resize((size_t)nf * cfg->n_mels);
        if (hipMemcpy(hmel.data(), mel_fb, sizeof(float) * hmel.size(), hipMemcpyDefault) != hipSuccess) { avexhip_set_error("melspec_plan_create: cannot read mel_fb"); return nullptr; }
        for (int m = 0; m < n_out; ++m) {
            int lo = -1, hi = -1;
            for (int k = 0; k < nf; ++k) if (hmel[(size_t)k * n_out + m] != 0.f) { if (lo < 0) lo = k; hi = k; }
            st[m] = lo < 0 ? 0 : lo; ln[m] = lo < 0 ? 0 : hi - lo + 1; of[m] = (int)packed.size();
            for (int k = 0; k < ln[m]; ++k) packed.push_back(hmel[(size_t)(lo + k) * n_out + m]);
        }
    }
    // DFT matrix with the (centre-padded, torch.stft) window folded in: X[k] = sum_n w[n] x[n] exp(-2 pi i k n / N)
    const int wl = (N - win) / 2;
    auto wat = [&](int n) { return (n >= wl && n < wl + win) ? hwin[n - wl] : 0.f; };
    bool sym = true;      // symmetric about n_fft / 2 (periodic Hann of any length; Hamming when win_length == n_fft)
    for (int n = 1; n < N / 2; ++n)
        if (fabsf(wat(n) - wat(N - n)) > 1e-6f * fmaxf(1.f, fabsf(wat(n)))) sym = false;
    const int kfold = sym ? (((N / 2 + 1) + 1) & ~1) : N;
    std::vector<float> dft((size_t)kfold * nc, 0.f);
    for (int n = 0; n < (sym ? N / 2 + 1 : N); ++n) {
        const float w = wat(n);
        for (int k = 0; k < nf; ++k) {
            const long long kn = ((long long)k * n) % N;             // exact argument reduction
            const double a = -2.0 * M_PI * (double)kn / (double)N;
            dft[(size_t)n * nc + k] = (float)((double)w * cos(a));
            dft[(size_t)n * nc + half + k] = (float)((double)w * sin(a));
        }
    }
    const size_t o_dft = 0, o_st = sizeof(float) * dft.size(), o_ln = o_st + sizeof(int) * n_out, o_of = o_ln + sizeof(int) * n_out,
                 o_w = o_of + sizeof(int) * n_out, total = o_w + sizeof(float) * (packed.size() + 1);
    std::vector<char> host(total, 0);
    memcpy(host.data() + o_dft, dft.data(), sizeof(float) * dft.size());
    memcpy(host.data() + o_st, st.data(), sizeof(int) * n_out);
    memcpy(host.data() + o_ln, ln.data(), sizeof(int) * n_out);
    memcpy(host.data() + o_of, of.data(), sizeof(int) * n_out);
    if (!packed.empty()) memcpy(host.data() + o_w, packed.data(), sizeof(float) * packed.size());
    void* d = nullptr;
    if (hipMalloc(&d, total) != hipSuccess || hipMemcpy(d, host.data(), total, hipMemcpyHostToDevice) != hipSuccess) {
        avexhip_set_error("melspec_plan_create: device allocation failed");
        if (d) (void)hipFree(d);
        return nullptr;
    }
    avexhip_melspec_plan* p = new avexhip_melspec_plan();
    p->cfg = *cfg; p->blob = d;
    char* base = (char*)d;
    MelDev& md = p->dev;
    md.n_fft = N; md.hop = hop; md.n_freq = nf; md.half = half; md.nc = nc; md.n_out = n_out; md.center = cfg->center ? 1 : 0;
    md.skew = (hop % 2 == 0) ? 1 : 0; md.use_mel = cfg->n_mels > 0 ? 1 : 0; md.kfold = kfold; md.fold = sym ? 1 : 0;
    md.dft = (const float*)(base + o_dft); md.mel_start = (const int*)(base + o_st); md.mel_len = (const int*)(base + o_ln);
    md.mel_off = (const int*)(base + o_of); md.mel_w = (const float*)(base + o_w);
    const int nseg = hop * 31 + N;
    const int xs_words = nseg + (md.skew ? nseg / hop + 1 : 0);
    p->lds = sizeof(float) * (((xs_words + 3) & ~3) + 32 * (size_t)(half + 1));
    if (p->lds > 160 * 1024) { avexhip_set_error("melspec_plan_create: hop=%d n_fft=%d need %zu bytes of LDS", hop, N, p->lds); (void)hipFree(d); delete p; return nullptr; }
    return p;
}

extern "C" void avexhip_melspec_plan_destroy(avexhip_melspec_plan* p) {
    if (!p) return;
    if (p->blob) (void)hipFree(p->blob);
    delete p;
}

extern "C" int avexhip_melspec_num_frames(const avexhip_melspec_plan* p, int64_t T) {
    if (!p) return 0;
    if (p->cfg.center) return (int)(1 + T / p->cfg.hop_length);
    return T < p->cfg.n_fft ? 0 : (int)(1 + (T - p->cfg.n_fft) / p->cfg.hop_length);
}

extern "C" int avexhip_melspec_num_bins(const avexhip_melspec_plan* p) { return p ? p->dev.n_out : 0; }

extern "C" int avexhip_melspec_forward(const avexhip_melspec_plan* p, const float* wav_dev, int B, int64_t T, int64_t wav_stride,
                                       float* out_dev, int* minmax_dev, void* stream) {
    AVX_REQUIRE(p && wav_dev && out_dev, "melspec_forward: null argument");
    AVX_REQUIRE(B > 0 && T > 0, "melspec_forward: empty input");
    AVX_REQUIRE(!p->cfg.center || T > p->cfg.n_fft / 2, "melspec_forward: reflect padding needs more than n_fft/2 = %d samples (got %lld)", p->cfg.n_fft / 2, (long long)T);
    AVX_REQUIRE(!p->cfg.normalize || minmax_dev, "melspec_forward: normalisation needs the [B, 2] int scratch");
    if (wav_stride <= 0) wav_stride = T;
    const int frames = avexhip_melspec_num_frames(p, T);
    AVX_REQUIRE(frames > 0, "melspec_forward: input shorter than one frame");
    hipStream_t s = (hipStream_t)stream;
    int* mm = p->cfg.normalize ? minmax_dev : nullptr;
    if (mm) hipLaunchKernelGGL(melspec_minmax_init, dim3((B + 255) / 256), dim3(256), 0, s, mm, B);
    const dim3 grid((frames + 31) / 32, B);
    const int ntw = p->dev.half / 64;
    const int take_log = p->cfg.normalize ? 1 : 0;
#define AVX_MEL_LAUNCH(NTW)                                                                                              \
    do {                                                                                                                 \
        AVX_ENSURE_LDS(melspec_kernel<NTW>, 160 * 1024);                                                                 \
        hipLaunchKernelGGL(melspec_kernel<NTW>, grid, dim3(256), p->lds, s, p->dev, wav_dev, T, wav_stride, frames, out_dev, mm, take_log); \
    } while (0)
    switch (ntw) {
        case 3: AVX_MEL_LAUNCH(3); break;
        case 4: AVX_MEL_LAUNCH(4); break;
        case 5: AVX_MEL_LAUNCH(5); break;
        case 7: AVX_MEL_LAUNCH(7); break;
        case 9: AVX_MEL_LAUNCH(9); break;
        default: avexhip_set_error("melspec_forward: column tiling %d not instantiated", ntw); return AVEXHIP_ERR_INVALID;
    }
#undef AVX_MEL_LAUNCH
    AVX_LAUNCH_CHECK();
    if (mm) {
        const int64_t per_clip = (int64_t)p->dev.n_out * frames;
        hipLaunchKernelGGL(melspec_norm_kernel, dim3(64, B), dim3(256), 0, s, out_dev, per_clip, mm);
        AVX_LAUNCH_CHECK();
    }
    return AVEXHIP_OK;
}
