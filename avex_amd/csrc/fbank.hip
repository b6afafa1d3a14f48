// Fused kaldi-compatible log-mel filterbank (the BEATs frontend) for gfx950.
//
// Restates avex/models/beats/beats.py:120-163 (_BatchedFbank.forward) + :322-323 (x * 2**15 and the
// (x - mean) / (2 std) normalisation) as ONE kernel; nothing but the waveform is read from HBM and
// nothing but the final features is written (algorithmic traffic 640 KB in + 511 KB out per 10 s clip
// for the fp32 output, 254 KB for the half patch-major output consumed by the patch-embed GEMM).
//
// One wave64 handles a PAIR of frames packed as one complex sequence z[n] = a[n] + i b[n]
// (n < 512, zero beyond the 400-sample window), so a single 512-point complex FFT yields both real
// spectra.  512 = 8*8*8: three register-resident radix-8 passes with each lane holding 8 points,
// exchanged through a per-wave LDS buffer (row stride 72 complex to stay bank-conflict free):
//   pass 1  lane n2 holds x[64 n1 + n2], n1 = 0..7 (exactly how frames load coalesced from HBM)
//   pass 2  lane (k1, m2) takes Y[k1][8 m1 + m2], m1 = 0..7
//   pass 3  lane l = k1 + 8 j1 takes U[k1][j1][m2], m2 = 0..7 and ends with X[l + 64 j2]
// then |A|^2, |B|^2 by conjugate symmetry, sparse triangular mel (<= ~10 taps per bin for 128 bins,
// CSR built on the host from the reference's dense [257, n_mels] matrix), log, affine.
#include <math.h>

#include <vector>

#include "common.h"

namespace {

__device__ __forceinline__ float2 operator+(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 operator-(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }  // a * (-i)

// in-place 8-point DFT, natural order in and out: x[k] <- sum_n x[n] exp(-2 pi i n k / 8)
__device__ __forceinline__ void dft8(float2 (&x)[8]) {
    const float c = 0.70710678118654752440f;
    const float2 a0 = x[0] + x[4], a1 = x[0] - x[4], a2 = x[2] + x[6], a3 = mul_mi(x[2] - x[6]);
    const float2 b0 = x[1] + x[5], b1 = x[1] - x[5], b2 = x[3] + x[7], b3 = mul_mi(x[3] - x[7]);
    const float2 e0 = a0 + a2, e2 = a0 - a2, e1 = a1 + a3, e3 = a1 - a3;
    const float2 o0 = b0 + b2, o2 = b0 - b2, o1 = b1 + b3, o3 = b1 - b3;
    const float2 t1 = make_float2(c * (o1.x + o1.y), c * (o1.y - o1.x));   // o1 * W8^1
    const float2 t2 = mul_mi(o2);                                          // o2 * W8^2
    const float2 t3 = make_float2(c * (o3.y - o3.x), -c * (o3.x + o3.y));  // o3 * W8^3
    x[0] = e0 + o0; x[4] = e0 - o0;
    x[1] = e1 + t1; x[5] = e1 - t1;
    x[2] = e2 + t2; x[6] = e2 - t2;
    x[3] = e3 + t3; x[7] = e3 - t3;
}

constexpr int ZROW = 72;  // complex elements per LDS row (64 + 8 pad)
// Every exchange buffer below is private to one wave, and a wave's LDS operations execute in order: the passes need no
// workgroup barrier, only the compiler kept from moving LDS accesses across the exchange points.
#define AVX_WAVE_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// Diagnostic build only (scripts/debug/conc_probe3.hip compiles this file with -DAVX_FBANK_TAPS): every wave dumps its
// registers after each stage so a wrong result can be traced to the first stage that differs.  Never defined in the library.
#ifdef AVX_FBANK_TAPS
__device__ float2* g_fbank_taps;
#define AVX_TAP(stage) do { float2* t_ = g_fbank_taps + ((((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 8 + (stage)) * 64 + lane) * 8; \
                            _Pragma("unroll") for (int q_ = 0; q_ < 8; ++q_) t_[q_] = x[q_]; } while (0)
#else
#define AVX_TAP(stage) do { } while (0)
#endif

template <typename T>
__global__ __launch_bounds__(256) void fbank_kernel(avx::FbankDev fb, const float* __restrict__ wav,
                                                    int64_t stride, int frames,
                                                    float* __restrict__ out_f32, T* __restrict__ out_patch,
                                                    int P, const float* __restrict__ clip_offset, int out_frames) {
    __shared__ float2 zbuf[4][8 * ZROW];
    __shared__ float pw[4][2][264];
    __shared__ float2 tw[512];   // twiddle table staged once per block (pass 1 and 2 gather from it)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int f0 = (blockIdx.x * 4 + wave) * 2;
    const bool valid[2] = {f0 < frames, f0 + 1 < frames};
    float2* z = zbuf[wave];
    tw[threadIdx.x] = fb.twiddle[threadIdx.x];
    tw[threadIdx.x + 256] = fb.twiddle[threadIdx.x + 256];
    __syncthreads();

    // ---- load, DC removal, pre-emphasis, window (beats.py:136-151) ----------------------------
    float2 x[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) x[q] = make_float2(0.f, 0.f);
    // A frame that is exactly zero after DC removal (digital silence, zero padding, a constant) must come out at the log floor
    // in every bin, as the reference's per-frame FFT gives it.  Packed beside a loud partner its spectrum is the difference of two
    // large numbers and picks up the partner's rounding noise (-158 dB of the partner: e^-13.5 instead of the floor e^-15.9 next
    // to a full-scale impulse), so such a frame is flagged here and its power written as 0.
    bool live[2];
#pragma unroll
    for (int fr = 0; fr < 2; ++fr) {
        float cur[8], prev[8];
        float s = 0.f;
        // branch-free: every lane loads from an in-range address (index clamped to the window, frame clamped to the clip's last one)
        // and the values outside the window / of a frame that does not exist are zeroed by a select; a conditional load per element
        // put each of the 40 loads in its own basic block with its own wait
        const bool vfr = valid[fr];
        const int fsafe = vfr ? f0 + fr : (frames > 0 ? frames - 1 : 0);
        const float* src = wav + (int64_t)b * stride + (int64_t)fsafe * fb.hop;
        const float coff = clip_offset ? clip_offset[b] : 0.f;   // EAT: mono - mono.mean() (eat/audio_processor.py:107)
        float wn[8], cv[8], pvv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { cv[q] = 0.f; pvv[q] = 0.f; }
        if (frames > 0) {                                        // (block-uniform: a clip shorter than one window has nothing to read)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int n = lane + 64 * q;
                const int nc = n < fb.win ? n : fb.win - 1;
                cv[q] = src[nc];
                pvv[q] = src[nc > 0 ? nc - 1 : 0];
            }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int n = lane + 64 * q;
            const int nc = n < fb.win ? n : fb.win - 1;
            wn[q] = fb.window[nc];
            const bool in = vfr && n < fb.win;
            cur[q] = in ? (cv[q] - coff) * fb.input_scale : 0.f;
            prev[q] = in ? (pvv[q] - coff) * fb.input_scale : 0.f;
            s += cur[q];
        }
        const float mean = fb.remove_dc ? wave_sum(s) / (float)fb.win : 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int n = lane + 64 * q;
            float y = 0.f;
            if (vfr && n < fb.win) {
                const float c = cur[q] - mean, pv = prev[q] - mean;
                y = (c - fb.preemph * pv) * wn[q];
            }
            if (fr == 0) x[q].x = y; else x[q].y = y;
        }
        bool nz = false;
#pragma unroll
        for (int q = 0; q < 8; ++q) nz = nz || (fr == 0 ? x[q].x : x[q].y) != 0.f;
        live[fr] = __any(nz);
    }

    // ---- 512-point complex FFT: three radix-8 passes ------------------------------------------
    AVX_TAP(0);
#ifdef AVX_FBANK_TAPS
    {   // the first dft8 spelled out with a tap after its first level (tap slot 7 = a0..a3, b0..b3; slot 1 = result)
        const float c = 0.70710678118654752440f;
        float2 y[8];
        y[0] = x[0] + x[4]; y[1] = x[0] - x[4]; y[2] = x[2] + x[6]; y[3] = mul_mi(x[2] - x[6]);
        y[4] = x[1] + x[5]; y[5] = x[1] - x[5]; y[6] = x[3] + x[7]; y[7] = mul_mi(x[3] - x[7]);
        { float2* t_ = g_fbank_taps + ((((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 8 + 7) * 64 + lane) * 8;
          _Pragma("unroll") for (int q_ = 0; q_ < 8; ++q_) t_[q_] = y[q_]; }
        const float2 e0 = y[0] + y[2], e2 = y[0] - y[2], e1 = y[1] + y[3], e3 = y[1] - y[3];
        const float2 o0 = y[4] + y[6], o2 = y[4] - y[6], o1 = y[5] + y[7], o3 = y[5] - y[7];
        const float2 t1 = make_float2(c * (o1.x + o1.y), c * (o1.y - o1.x));
        const float2 t2 = mul_mi(o2);
        const float2 t3 = make_float2(c * (o3.y - o3.x), -c * (o3.x + o3.y));
        x[0] = e0 + o0; x[4] = e0 - o0; x[1] = e1 + t1; x[5] = e1 - t1; x[2] = e2 + t2; x[6] = e2 - t2; x[3] = e3 + t3; x[7] = e3 - t3;
    }
#else
    dft8(x);
#endif
    AVX_TAP(1);
#pragma unroll
    for (int k1 = 1; k1 < 8; ++k1) x[k1] = cmul(x[k1], tw[lane * k1]);
    AVX_TAP(2);
#pragma unroll
    for (int k1 = 0; k1 < 8; ++k1) z[k1 * ZROW + lane] = x[k1];
    AVX_WAVE_SYNC();
    {
        const int k1 = lane >> 3, m2 = lane & 7;
#pragma unroll
        for (int m1 = 0; m1 < 8; ++m1) x[m1] = z[k1 * ZROW + 8 * m1 + m2];
        AVX_TAP(3);
        dft8(x);
#pragma unroll
        for (int j1 = 1; j1 < 8; ++j1) x[j1] = cmul(x[j1], tw[8 * m2 * j1]);
        AVX_TAP(4);
        AVX_WAVE_SYNC();
#pragma unroll
        for (int j1 = 0; j1 < 8; ++j1) z[k1 * ZROW + j1 * 8 + m2] = x[j1];
    }
    AVX_WAVE_SYNC();
    {
        const int k1 = lane & 7, j1 = lane >> 3;
#pragma unroll
        for (int m2 = 0; m2 < 8; ++m2) x[m2] = z[k1 * ZROW + j1 * 8 + m2];
        AVX_TAP(5);
        dft8(x);
        AVX_TAP(6);
        AVX_WAVE_SYNC();
#pragma unroll
        for (int j2 = 0; j2 < 8; ++j2) z[lane + 64 * j2] = x[j2];  // natural order Z[k]
    }
    AVX_WAVE_SYNC();

    // ---- split the two real spectra, power (beats.py:154-155) ----------------------------------
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const int k = lane + 64 * q;
        if (k <= 256) {
            const float2 zk = z[k], zn = z[(512 - k) & 511];
            const float ar = 0.5f * (zk.x + zn.x), ai = 0.5f * (zk.y - zn.y);
            const float br = 0.5f * (zk.y + zn.y), bi = -0.5f * (zk.x - zn.x);
#ifndef FBANK_ABS_SQ
#define FBANK_ABS_SQ 1     // 1: |X|^2 as the reference forms it, abs() then ** 2 (beats.py:155); 0: re^2 + im^2 directly (differs in the last bit)
#endif
#if FBANK_ABS_SQ
            const float ma = sqrtf(ar * ar + ai * ai), mb = sqrtf(br * br + bi * bi);
            pw[wave][0][k] = live[0] ? ma * ma : 0.f;
            pw[wave][1][k] = live[1] ? mb * mb : 0.f;
#else
            pw[wave][0][k] = live[0] ? ar * ar + ai * ai : 0.f;
            pw[wave][1][k] = live[1] ? br * br + bi * bi : 0.f;
#endif
        }
    }
    AVX_WAVE_SYNC();

    // ---- mel, log, affine (beats.py:159-163,323) -----------------------------------------------
    const int nm = fb.n_mels;
    const int nt = out_frames / P, nf = nm / P;
    // one mel bin of BOTH frames per lane and step: the filter's taps and weights are read once for the pair (same per-frame summation
    // order as one bin at a time: bit-identical)
    for (int m = lane; m < nm; m += 64) {
        const int st = fb.mel_start[m], len = fb.mel_len[m], off = fb.mel_off[m];
        float e0 = 0.f, e1 = 0.f;
#pragma unroll 4
        for (int t = 0; t < len; ++t) {
            const float wt = fb.mel_w[off + t];
            e0 += pw[wave][0][st + t] * wt;
            e1 += pw[wave][1][st + t] * wt;
        }
#pragma unroll
        for (int fr = 0; fr < 2; ++fr) {
            const int f = f0 + fr;
            if (f >= out_frames) continue;
            float y;
            // (the floor as a compare + select, not fmaxf: fmaxf returns its non-NaN operand and would turn a NaN sample into the log floor -- a finite
            //  embedding for a poisoned clip; torch.max(mel_energies, eps) in the reference's kaldi fbank propagates the NaN)
            const float en = fr ? e1 : e0;
            if (f < frames) y = (logf(en < fb.log_floor ? fb.log_floor : en) - fb.norm_mean) / fb.norm_div;
            else y = (0.f - fb.norm_mean) / fb.norm_div;      // zero-padded log-mel rows, normalised like the rest (audio_processor.py:121-135)
            if (out_f32) out_f32[((int64_t)b * out_frames + f) * nm + m] = y;
            if (out_patch) {
                const int tp = f / P;
                if (tp < nt && m < nf * P) {
                    const int64_t tok = ((int64_t)b * nt + tp) * nf + m / P;
                    out_patch[(tok * P + (f % P)) * P + (m % P)] = Half<T>::from(y);
                }
            }
        }
    }
}

// mean[b] = mean(wav[b, :T]) (fp32; block tree sum): the EAT frontend's per-clip DC removal (eat/audio_processor.py:107)
__global__ __launch_bounds__(1024) void clip_mean_kernel(const float* __restrict__ wav, int64_t T, int64_t stride, float* __restrict__ mean) {
    __shared__ float red[16];
    const float* src = wav + (int64_t)blockIdx.x * stride;
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < T; i += 1024) s += src[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x < 64) {
        float v = threadIdx.x < 16 ? red[threadIdx.x] : 0.f;
        v = wave_sum(v);
        if (threadIdx.x == 0) mean[blockIdx.x] = v / (float)T;
    }
}

}  // namespace

namespace avx {

int fbank(const FbankDev& fb, const float* wav, int B, int64_t T, int64_t stride, int frames,
          float* out_f32, void* out_patch, int patch, int dtype, hipStream_t s, const float* clip_offset, int out_frames) {
    AVX_REQUIRE(wav && B > 0, "fbank: bad arguments");
    AVX_REQUIRE(fb.win > 0 && fb.win <= 512 && fb.hop > 0, "fbank: win_length=%d must be in 1..512", fb.win);
    AVX_REQUIRE(out_f32 || out_patch, "fbank: no output");
    AVX_REQUIRE(stride >= T, "fbank: wav_stride < T");
    if (out_frames <= 0) out_frames = frames;      // output rows per clip: rows >= frames are padding, rows >= out_frames are cut
    if (frames > out_frames) frames = out_frames;
    if (out_frames <= 0) return AVEXHIP_OK;
    if (patch <= 0) patch = 16;
    const dim3 grid(((out_frames + 1) / 2 + 3) / 4, B);
    if (dtype == AVEXHIP_BF16)
        hipLaunchKernelGGL(fbank_kernel<__bf16>, grid, dim3(256), 0, s, fb, wav, stride, frames, out_f32, (__bf16*)out_patch, patch, clip_offset, out_frames);
    else
        hipLaunchKernelGGL(fbank_kernel<_Float16>, grid, dim3(256), 0, s, fb, wav, stride, frames, out_f32, (_Float16*)out_patch, patch, clip_offset, out_frames);
    AVX_LAUNCH_CHECK();
    return AVEXHIP_OK;
}

}  // namespace avx

// ---------------------------------------------------------------------------------------------
// Plan (host): twiddles, window copy, CSR mel bank
// ---------------------------------------------------------------------------------------------
struct avexhip_fbank_plan {
    avexhip_fbank_config cfg;
    avx::FbankDev dev;
    void* d_blob = nullptr;
};

extern "C" avexhip_fbank_plan* avexhip_fbank_plan_create(const avexhip_fbank_config* cfg, const float* window,
                                                         const float* mel_fb) {
    if (!cfg || !window || !mel_fb) {
        avexhip_set_error("fbank_plan_create: null argument");
        return nullptr;
    }
    if (cfg->win_length <= 0 || cfg->win_length > 512 || cfg->hop_length <= 0 || cfg->n_mels <= 0 ||
        cfg->n_mels > 1024) {
        avexhip_set_error("fbank_plan_create: unsupported win_length=%d hop=%d n_mels=%d (n_fft is fixed at 512)",
                          cfg->win_length, cfg->hop_length, cfg->n_mels);
        return nullptr;
    }
    const int win = cfg->win_length, nm = cfg->n_mels, nb = 257;
    std::vector<float> hwin(win), hmel((size_t)nb * nm);
    if (hipMemcpy(hwin.data(), window, sizeof(float) * win, hipMemcpyDefault) != hipSuccess ||
        hipMemcpy(hmel.data(), mel_fb, sizeof(float) * nb * nm, hipMemcpyDefault) != hipSuccess) {
        avexhip_set_error("fbank_plan_create: cannot read window/mel_fb");
        return nullptr;
    }
    std::vector<int> start(nm), len(nm), off(nm);
    std::vector<float> packed;
    for (int m = 0; m < nm; ++m) {
        int lo = -1, hi = -1;
        for (int k = 0; k < nb; ++k)
            if (hmel[(size_t)k * nm + m] != 0.f) {
                if (lo < 0) lo = k;
                hi = k;
            }
        start[m] = lo < 0 ? 0 : lo;
        len[m] = lo < 0 ? 0 : hi - lo + 1;
        off[m] = (int)packed.size();
        for (int k = 0; k < len[m]; ++k) packed.push_back(hmel[(size_t)(lo + k) * nm + m]);
    }
    std::vector<float2> tw(512);
    for (int k = 0; k < 512; ++k) {
        const double a = -2.0 * M_PI * (double)k / 512.0;
        tw[k] = make_float2((float)cos(a), (float)sin(a));
    }
    // one device blob: window | twiddle | start | len | off | packed weights
    const size_t o_win = 0;
    const size_t o_tw = (o_win + sizeof(float) * win + 15) & ~(size_t)15;
    const size_t o_st = o_tw + sizeof(float2) * 512;
    const size_t o_len = o_st + sizeof(int) * nm;
    const size_t o_off = o_len + sizeof(int) * nm;
    const size_t o_w = o_off + sizeof(int) * nm;
    const size_t total = o_w + sizeof(float) * (packed.size() + 1);
    std::vector<char> host(total, 0);
    memcpy(host.data() + o_win, hwin.data(), sizeof(float) * win);
    memcpy(host.data() + o_tw, tw.data(), sizeof(float2) * 512);
    memcpy(host.data() + o_st, start.data(), sizeof(int) * nm);
    memcpy(host.data() + o_len, len.data(), sizeof(int) * nm);
    memcpy(host.data() + o_off, off.data(), sizeof(int) * nm);
    if (!packed.empty()) memcpy(host.data() + o_w, packed.data(), sizeof(float) * packed.size());
    void* d = nullptr;
    if (hipMalloc(&d, total) != hipSuccess || hipMemcpy(d, host.data(), total, hipMemcpyHostToDevice) != hipSuccess) {
        avexhip_set_error("fbank_plan_create: device allocation failed");
        if (d) (void)hipFree(d);
        return nullptr;
    }
    avexhip_fbank_plan* p = new avexhip_fbank_plan();
    p->cfg = *cfg;
    p->d_blob = d;
    char* base = (char*)d;
    p->dev.win = win; p->dev.hop = cfg->hop_length; p->dev.n_mels = nm;
    p->dev.input_scale = cfg->input_scale; p->dev.preemph = cfg->preemph; p->dev.log_floor = cfg->log_floor;
    p->dev.norm_mean = cfg->norm_mean; p->dev.norm_div = cfg->norm_div == 0.f ? 1.f : cfg->norm_div;
    p->dev.remove_dc = cfg->remove_dc;
    p->dev.window = (const float*)(base + o_win);
    p->dev.twiddle = (const float2*)(base + o_tw);
    p->dev.mel_start = (const int*)(base + o_st);
    p->dev.mel_len = (const int*)(base + o_len);
    p->dev.mel_off = (const int*)(base + o_off);
    p->dev.mel_w = (const float*)(base + o_w);
    return p;
}

extern "C" void avexhip_fbank_plan_destroy(avexhip_fbank_plan* plan) {
    if (!plan) return;
    if (plan->d_blob) (void)hipFree(plan->d_blob);
    delete plan;
}

extern "C" int avexhip_fbank_num_frames(const avexhip_fbank_plan* plan, int64_t T) {
    if (!plan) return 0;
    if (T < plan->cfg.win_length) return 0;
    return (int)(1 + (T - plan->cfg.win_length) / plan->cfg.hop_length);
}

extern "C" int avexhip_fbank_forward_patches(const avexhip_fbank_plan* plan, const float* wav_dev, int B, int64_t T, int64_t wav_stride,
                                             const float* clip_offset_dev, int out_frames, int patch, void* out_patch_dev, int dtype,
                                             void* stream) {
    AVX_REQUIRE(plan && wav_dev && out_patch_dev, "fbank_forward_patches: null argument");
    AVX_REQUIRE(B > 0 && T > 0 && patch > 0, "fbank_forward_patches: empty input B=%d T=%lld patch=%d", B, (long long)T, patch);
    AVX_REQUIRE(dtype == AVEXHIP_F16 || dtype == AVEXHIP_BF16, "fbank_forward_patches: unknown dtype %d", dtype);
    if (wav_stride <= 0) wav_stride = T;
    const int frames = avexhip_fbank_num_frames(plan, T);
    if (out_frames <= 0) out_frames = frames;
    AVX_REQUIRE(out_frames >= patch && plan->cfg.n_mels >= patch, "fbank_forward_patches: %d frames x %d bins hold no %d x %d patch", out_frames,
                plan->cfg.n_mels, patch, patch);
    return avx::fbank(plan->dev, wav_dev, B, T, wav_stride, frames, nullptr, out_patch_dev, patch, dtype, (hipStream_t)stream, clip_offset_dev,
                      out_frames);
}

// internal accessor for the encoder handle
const avx::FbankDev* avexhip_fbank_plan_dev(const avexhip_fbank_plan* plan) { return plan ? &plan->dev : nullptr; }

extern "C" int avexhip_fbank_forward(const avexhip_fbank_plan* plan, const float* wav_dev, int B, int64_t T,
                                     int64_t wav_stride, float* out_dev, void* stream) {
    AVX_REQUIRE(plan && wav_dev && out_dev, "fbank_forward: null argument");
    AVX_REQUIRE(B > 0 && T > 0, "fbank_forward: empty input B=%d T=%lld", B, (long long)T);
    if (wav_stride <= 0) wav_stride = T;
    const int frames = avexhip_fbank_num_frames(plan, T);
    return avx::fbank(plan->dev, wav_dev, B, T, wav_stride, frames, out_dev, nullptr, 16, AVEXHIP_F16,
                      (hipStream_t)stream);
}

extern "C" int avexhip_clip_mean(const float* wav_dev, int B, int64_t T, int64_t wav_stride, float* mean_dev, void* stream) {
    AVX_REQUIRE(wav_dev && mean_dev, "clip_mean: null argument");
    AVX_REQUIRE(B > 0 && T > 0, "clip_mean: empty input B=%d T=%lld", B, (long long)T);
    if (wav_stride <= 0) wav_stride = T;
    hipLaunchKernelGGL(clip_mean_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, wav_dev, T, wav_stride, mean_dev);
    AVX_LAUNCH_CHECK();
    return AVEXHIP_OK;
}

extern "C" int avexhip_fbank_forward_padded(const avexhip_fbank_plan* plan, const float* wav_dev, int B, int64_t T,
                                            int64_t wav_stride, const float* clip_offset_dev, int out_frames, float* out_dev,
                                            void* stream) {
    AVX_REQUIRE(plan && wav_dev && out_dev, "fbank_forward_padded: null argument");
    AVX_REQUIRE(B > 0 && T > 0 && out_frames > 0, "fbank_forward_padded: empty input B=%d T=%lld out_frames=%d", B, (long long)T, out_frames);
    if (wav_stride <= 0) wav_stride = T;
    const int frames = avexhip_fbank_num_frames(plan, T);
    return avx::fbank(plan->dev, wav_dev, B, T, wav_stride, frames, out_dev, nullptr, 16, AVEXHIP_F16, (hipStream_t)stream,
                      clip_offset_dev, out_frames);
}

