// Audio ingest on the device: PCM samples as the file holds them -> mono float32 -> the model's sample rate.
//
// The reference decodes on the host (soundfile / torchaudio.load), averages channels (`noise_wav.mean(dim=0)`,
// avex/data/augmentations.py:269-271; `audio_stereo_to_mono(..., "average")`, birdset_train_splits.py:186) and resamples with
// `torchaudio.transforms.Resample(orig, new)` (augmentations.py:274-276) -- torchaudio's documented band-limited sinc interpolation
// (Hann-windowed sinc, lowpass_filter_width 6, rolloff 0.99; optionally Kaiser-windowed), restated here:
//   orig, new reduced by their gcd;  base = min(orig, new) * rolloff;  width = ceil(lowpass_width * orig / base)
//   kernel[p][j] = sinc(t) * window(t) * base / orig,   t = ((j - width) / orig - p / new) * base  clamped to +-lowpass_width
//   out[q * new + p] = sum_j kernel[p][j] * x[q * orig + j - width]   (zero outside the clip),   length ceil(new * T / orig)
// PARITY UNPINNED against torchaudio (absent from both machines); the checker is oracle/ingest_oracle.py.
#include <math.h>

#include <vector>

#include "common.h"

namespace {

// interleaved PCM [frames][channels] of the given sample format -> mono fp32 [frames] (mean over channels), normalised like
// soundfile / torchaudio do for float32 output: int16 / 32768, int24 / 2^23, int32 / 2^31, uint8 (x - 128) / 128
__global__ __launch_bounds__(256) void pcm_to_mono_kernel(const unsigned char* __restrict__ raw, int fmt, int channels, int64_t frames, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= frames) return;
    float acc = 0.f;
    for (int c = 0; c < channels; ++c) {
        const int64_t e = i * channels + c;
        float v;
        if (fmt == 16) v = (float)((const short*)raw)[e] * (1.0f / 32768.0f);
        else if (fmt == 32) v = (float)((const int*)raw)[e] * (1.0f / 2147483648.0f);
        else if (fmt == 24) {
            const unsigned char* p = raw + 3 * e;
            int s = (int)p[0] | ((int)p[1] << 8) | ((int)(signed char)p[2] << 16);
            v = (float)s * (1.0f / 8388608.0f);
        } else if (fmt == 8) v = ((float)raw[e] - 128.0f) * (1.0f / 128.0f);
        else if (fmt == 64) v = (float)((const double*)raw)[e];
        else v = ((const float*)raw)[e];                      // fmt == 0: float32
        acc += v;
    }
    out[i] = channels > 1 ? acc / (float)channels : acc;
}

// One workgroup = 256 consecutive output samples of one clip.  The input samples they touch are staged in LDS; the polyphase table
// [new][taps] is read through the caches (each output reads one row).
__global__ __launch_bounds__(256) void resample_kernel(const float* __restrict__ x, int64_t T, int64_t x_stride, const float* __restrict__ table, int orig, int newr,
                                                       int width, int taps, float* __restrict__ out, int64_t n_out, int64_t out_stride) {
    extern __shared__ float xs[];
    const int b = blockIdx.y;
    const int64_t i0 = (int64_t)blockIdx.x * 256;
    const int64_t q0 = i0 / newr, q1 = (i0 + 255) / newr;
    const int64_t lo = q0 * orig - width;                       // first input sample any of the 256 outputs reads
    const int span = (int)((q1 - q0) * orig) + taps;
    const float* src = x + (int64_t)b * x_stride;
    for (int s = threadIdx.x; s < span; s += 256) {
        const int64_t j = lo + s;
        xs[s] = (j >= 0 && j < T) ? src[j] : 0.f;
    }
    __syncthreads();
    const int64_t i = i0 + threadIdx.x;
    if (i >= n_out) return;
    const int64_t q = i / newr;
    const int p = (int)(i - q * newr);
    const float* row = table + (int64_t)p * taps;
    const float* xin = xs + (int)((q - q0) * orig);
    float acc = 0.f;
    for (int j = 0; j < taps; ++j) acc = __builtin_fmaf(row[j], xin[j], acc);
    out[(int64_t)b * out_stride + i] = acc;
}

// resampy's interpolating resampler (librosa.resample(..., res_type="kaiser_best"), birdset_train_splits.py:190-196): output sample t sits
// at input time t / ratio; its two filter wings walk a half-window table (64 zero crossings x 512 samples each, Kaiser-windowed sinc)
// in steps of scale * 512 entries, every weight linearly interpolated between neighbouring table entries.  One thread per output
// sample; table and input through the caches (the table is 128 KB, a wing touches every index_step-th entry of it).
__global__ __launch_bounds__(256) void resample_interp_kernel(const float* __restrict__ x, int64_t T, int64_t x_stride, const float* __restrict__ win,
                                                              const float* __restrict__ delta, int nwin, int num_table, int index_step, double scale,
                                                              double time_increment, int64_t n_res, float post, float* __restrict__ out, int64_t n_out,
                                                              int64_t out_stride) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n_out) return;
    const int b = blockIdx.y;
    float acc = 0.f;
    if (t < n_res) {                                            // librosa's fix_length zero-fills beyond resampy's int(T * ratio) samples
        const float* src = x + (int64_t)b * x_stride;
        const double time_register = (double)t * time_increment;
        const int64_t n = (int64_t)time_register;
        double frac = scale * (time_register - (double)n);
        double index_frac = frac * num_table;
        int offset = (int)index_frac;
        float eta = (float)(index_frac - offset);
        int64_t i_max = (nwin - offset) / index_step;
        i_max = i_max < n + 1 ? i_max : n + 1;
        for (int64_t i = 0; i < i_max; ++i) {
            const int k = offset + (int)i * index_step;
            acc = __builtin_fmaf(__builtin_fmaf(eta, delta[k], win[k]), src[n - i], acc);
        }
        frac = scale - frac;
        index_frac = frac * num_table;
        offset = (int)index_frac;
        eta = (float)(index_frac - offset);
        int64_t k_max = (nwin - offset) / index_step;
        k_max = k_max < T - n - 1 ? k_max : T - n - 1;
        for (int64_t k = 0; k < k_max; ++k) {
            const int q = offset + (int)k * index_step;
            acc = __builtin_fmaf(__builtin_fmaf(eta, delta[q], win[q]), src[n + k + 1], acc);
        }
    }
    out[(int64_t)b * out_stride + t] = acc * post;
}

int gcd_i(int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; }

}  // namespace

struct avexhip_resample_plan {
    int orig = 0, newr = 0, width = 0, taps = 0;
    float* table = nullptr;
    // interpolating (resampy) plans: table = [win | delta], each nwin floats
    bool interp = false;
    int nwin = 0, num_table = 0, index_step = 0;
    double ratio = 1.0, scale = 1.0;
    float post = 1.f;
};

extern "C" avexhip_resample_plan* avexhip_resample_plan_create(int orig_freq, int new_freq, int lowpass_filter_width, double rolloff, double kaiser_beta) {
    if (orig_freq <= 0 || new_freq <= 0 || lowpass_filter_width <= 0 || !(rolloff > 0.0 && rolloff <= 1.0)) {
        avexhip_set_error("resample_plan_create: bad arguments orig=%d new=%d width=%d rolloff=%g", orig_freq, new_freq, lowpass_filter_width, rolloff);
        return nullptr;
    }
    const int g = gcd_i(orig_freq, new_freq);
    const int orig = orig_freq / g, newr = new_freq / g;
    const double base = (double)(orig < newr ? orig : newr) * rolloff;
    const int width = (int)ceil((double)lowpass_filter_width * orig / base);
    const int taps = 2 * width + orig;
    if ((int64_t)newr * taps > (1 << 26) || taps + 256 / newr * orig + 2 * orig > 36000) {
        avexhip_set_error("resample_plan_create: %d -> %d Hz reduces to %d -> %d (%d taps): table or staging too large", orig_freq, new_freq, orig, newr, taps);
        return nullptr;
    }
    std::vector<float> tab((size_t)newr * taps);
    const double lpw = (double)lowpass_filter_width;
    auto bessel_i0 = [](double z) { double s = 1.0, t = 1.0; for (int k = 1; k < 64; ++k) { t *= (z / (2.0 * k)) * (z / (2.0 * k)); s += t; if (t < 1e-18 * s) break; } return s; };
    for (int p = 0; p < newr; ++p)
        for (int j = 0; j < taps; ++j) {
            double t = ((double)(j - width) / orig - (double)p / newr) * base;
            t = t < -lpw ? -lpw : (t > lpw ? lpw : t);
            double w;
            if (kaiser_beta > 0.0) { const double r = t / lpw; w = bessel_i0(kaiser_beta * sqrt(1.0 - r * r)) / bessel_i0(kaiser_beta); }
            else { const double c = cos(t * M_PI / lpw / 2.0); w = c * c; }
            const double a = t * M_PI;
            const double sinc = a == 0.0 ? 1.0 : sin(a) / a;
            tab[(size_t)p * taps + j] = (float)(sinc * w * base / orig);
        }
    avexhip_resample_plan* pl = new avexhip_resample_plan();
    pl->orig = orig; pl->newr = newr; pl->width = width; pl->taps = taps;
    if (hipMalloc((void**)&pl->table, sizeof(float) * tab.size()) != hipSuccess ||
        hipMemcpy(pl->table, tab.data(), sizeof(float) * tab.size(), hipMemcpyHostToDevice) != hipSuccess) {
        avexhip_set_error("resample_plan_create: device allocation failed");
        if (pl->table) (void)hipFree(pl->table);
        delete pl;
        return nullptr;
    }
    return pl;
}

extern "C" avexhip_resample_plan* avexhip_resample_interp_plan_create(int orig_freq, int new_freq, int num_zeros, int precision, double rolloff,
                                                                       double kaiser_beta, int scale_energy) {
    if (orig_freq <= 0 || new_freq <= 0 || num_zeros <= 0 || num_zeros > 256 || precision < 1 || precision > 12 || !(rolloff > 0.0 && rolloff <= 1.0) ||
        !(kaiser_beta >= 0.0)) {
        avexhip_set_error("resample_interp_plan_create: bad arguments orig=%d new=%d zeros=%d precision=%d rolloff=%g beta=%g", orig_freq, new_freq, num_zeros,
                          precision, rolloff, kaiser_beta);
        return nullptr;
    }
    // resampy.filters.sinc_window: half of a Kaiser-windowed sinc on num_zeros * 2^precision + 1 points
    const int num_bits = 1 << precision;
    const int n = num_bits * num_zeros;
    const double ratio = (double)new_freq / (double)orig_freq;
    auto bessel_i0 = [](double z) { double s = 1.0, t = 1.0; for (int k = 1; k < 256; ++k) { t *= (z / (2.0 * k)) * (z / (2.0 * k)); s += t; if (t < 1e-18 * s) break; } return s; };
    std::vector<double> w((size_t)n + 1);
    const double i0b = bessel_i0(kaiser_beta);
    for (int i = 0; i <= n; ++i) {
        const double a = rolloff * (double)num_zeros * (double)i / (double)n;       // rolloff * linspace(0, num_zeros, n + 1)
        const double sinc = a == 0.0 ? 1.0 : sin(M_PI * a) / (M_PI * a);
        const double r = (double)i / (double)n;                                      // kaiser(2 n + 1)[n + i]
        const double taper = bessel_i0(kaiser_beta * sqrt(1.0 - r * r)) / i0b;
        w[i] = rolloff * sinc * taper * (ratio < 1.0 ? ratio : 1.0);                 // resampy scales the window when it decimates
    }
    std::vector<float> tab(2 * ((size_t)n + 1));
    for (int i = 0; i <= n; ++i) {
        tab[i] = (float)w[i];
        tab[(size_t)n + 1 + i] = i < n ? (float)(w[i + 1] - w[i]) : 0.f;
    }
    avexhip_resample_plan* pl = new avexhip_resample_plan();
    pl->interp = true; pl->orig = orig_freq; pl->newr = new_freq; pl->nwin = n + 1; pl->num_table = num_bits;
    pl->ratio = ratio; pl->scale = ratio < 1.0 ? ratio : 1.0;
    pl->index_step = (int)(pl->scale * num_bits);
    pl->post = scale_energy ? (float)(1.0 / sqrt(ratio)) : 1.f;
    if (pl->index_step < 1 || hipMalloc((void**)&pl->table, sizeof(float) * tab.size()) != hipSuccess ||
        hipMemcpy(pl->table, tab.data(), sizeof(float) * tab.size(), hipMemcpyHostToDevice) != hipSuccess) {
        avexhip_set_error("resample_interp_plan_create: %d -> %d Hz: ratio too small for the table, or device allocation failed", orig_freq, new_freq);
        if (pl->table) (void)hipFree(pl->table);
        delete pl;
        return nullptr;
    }
    return pl;
}

extern "C" void avexhip_resample_plan_destroy(avexhip_resample_plan* p) {
    if (!p) return;
    if (p->table) (void)hipFree(p->table);
    delete p;
}

extern "C" int64_t avexhip_resample_out_length(const avexhip_resample_plan* p, int64_t T) {
    if (!p || T <= 0) return 0;
    if (p->interp) return (int64_t)ceil((double)T * p->ratio);      // librosa: n_samples = ceil(len * ratio), resampy's int(len * ratio) zero-padded up to it
    return ((int64_t)p->newr * T + p->orig - 1) / p->orig;
}

extern "C" int avexhip_resample_forward(const avexhip_resample_plan* p, const float* x_dev, int B, int64_t T, int64_t x_stride, float* out_dev,
                                        int64_t out_stride, void* stream) {
    AVX_REQUIRE(p && x_dev && out_dev, "resample_forward: null argument");
    AVX_REQUIRE(B > 0 && T > 0 && B <= 65535, "resample_forward: bad shape B=%d T=%lld", B, (long long)T);
    const int64_t n_out = avexhip_resample_out_length(p, T);
    if (x_stride <= 0) x_stride = T;
    if (out_stride <= 0) out_stride = n_out;
    AVX_REQUIRE(B == 1 || x_stride >= T, "resample_forward: x_stride %lld < %lld input samples (rows would overlap)", (long long)x_stride, (long long)T);
    AVX_REQUIRE(out_stride >= n_out, "resample_forward: out_stride %lld < %lld output samples", (long long)out_stride, (long long)n_out);
    if (p->interp) {
        const int64_t n_res = (int64_t)((double)T * p->ratio);
        hipLaunchKernelGGL(resample_interp_kernel, dim3((unsigned)((n_out + 255) / 256), B), dim3(256), 0, (hipStream_t)stream, x_dev, T, x_stride, p->table,
                           p->table + p->nwin, p->nwin, p->num_table, p->index_step, p->scale, 1.0 / p->ratio, n_res, p->post, out_dev, n_out, out_stride);
        AVX_LAUNCH_CHECK();
        return AVEXHIP_OK;
    }
    const int span_max = (256 / p->newr + 2) * p->orig + p->taps;
    const size_t lds = sizeof(float) * (size_t)span_max;
    AVX_ENSURE_LDS(resample_kernel, 160 * 1024);
    hipLaunchKernelGGL(resample_kernel, dim3((unsigned)((n_out + 255) / 256), B), dim3(256), lds, (hipStream_t)stream, x_dev, T, x_stride, p->table, p->orig,
                       p->newr, p->width, p->taps, out_dev, n_out, out_stride);
    AVX_LAUNCH_CHECK();
    return AVEXHIP_OK;
}

extern "C" int avexhip_pcm_to_mono_f32(const void* raw_dev, int sample_format, int channels, int64_t frames, float* out_dev, void* stream) {
    AVX_REQUIRE(raw_dev && out_dev, "pcm_to_mono_f32: null argument");
    AVX_REQUIRE(channels > 0 && channels <= 64 && frames > 0, "pcm_to_mono_f32: bad shape channels=%d frames=%lld", channels, (long long)frames);
    AVX_REQUIRE(sample_format == 0 || sample_format == 8 || sample_format == 16 || sample_format == 24 || sample_format == 32 || sample_format == 64,
                "pcm_to_mono_f32: sample_format %d (0 = float32, 64 = float64, 8 / 16 / 24 / 32 = integer PCM)", sample_format);
    hipLaunchKernelGGL(pcm_to_mono_kernel, dim3((unsigned)((frames + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const unsigned char*)raw_dev, sample_format,
                       channels, frames, out_dev);
    AVX_LAUNCH_CHECK();
    return AVEXHIP_OK;
}
