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
    // FFT path (n_fft = 2^a 3^b 5^c <= 2048): mixed-radix Stockham passes in LDS
    int n_pass, radix[12], fb;       // radices in pass order; frames per workgroup (32, or 16 above 1024 points)
    const float2* tw;        // [n_fft] exp(-2 pi i k / n_fft)
    const float* win;        // [n_fft] analysis window, centre-padded to n_fft like torch.stft
    const float* dft;        // [kfold][nc], kfold = n_fft/2 + 1 rounded up to even: the folded contraction (see the kernel)
    const int* mel_start;    // [n_out] CSR over frequency bins (mel) -- unused for plain spectrograms
    const int* mel_len;
    const int* mel_off;
    const float* mel_w;
    int mel_nnz;             // packed weights (floats); > 0 with mel_lds: the FFT kernel keeps the CSR bank in LDS
    int mel_lds;
};

__device__ __forceinline__ int ord_key(float v) {            // monotonic float -> int
    const int b = __builtin_bit_cast(int, v);
    return b >= 0 ? b : b ^ 0x7fffffff;
}
__device__ __forceinline__ float ord_val(int k) { return __builtin_bit_cast(float, k >= 0 ? k : k ^ 0x7fffffff); }

template <int FB>
__device__ __forceinline__ void melspec_epilogue(const MelDev& md, const float* pw, int pw_ld, int b, int f0, int frames, float* __restrict__ out,
                                                 int* __restrict__ minmax, int take_log, int tid, int nthreads);

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
    melspec_epilogue<32>(md, pw, pw_ld, b, f0, frames, out, minmax, take_log, tid, 256);
}

// mel (CSR) or plain bins, log, store [clip][bin][frame], per-clip min / max -- shared by the dense and the FFT kernel.
// pw: [FB][pw_ld] power spectra of the block's FB frames in LDS; thread (f = tid % FB, grp = tid / FB) walks bins grp, grp + G, ...
template <int FB>
__device__ __forceinline__ void melspec_epilogue(const MelDev& md, const float* pw, int pw_ld, int b, int f0, int frames, float* __restrict__ out,
                                                 int* __restrict__ minmax, int take_log, int tid, int nthreads) {
    const int f = tid % FB, grp = tid / FB, G = nthreads / FB;
    float mn = __builtin_inff(), mx = -__builtin_inff();
    const bool fvalid = f0 + f < frames;
    for (int m = grp; m < md.n_out; m += G) {
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
        if ((tid & 63) == 0 && mx >= mn) { atomicMin(minmax + 2 * b, ord_key(mn)); atomicMax(minmax + 2 * b + 1, ord_key(mx)); }
    }
}

__device__ __forceinline__ float2 c_add(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 c_sub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 c_mul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 c_mni(float2 a) { return make_float2(a.y, -a.x); }              // a * (-i)
__device__ __forceinline__ float2 c_scale(float2 a, float s) { return make_float2(a.x * s, a.y * s); }

// r-point forward DFTs (exp(-2 pi i / r)), in place
__device__ __forceinline__ void dft2(float2* v) { const float2 a = v[0], b = v[1]; v[0] = c_add(a, b); v[1] = c_sub(a, b); }
__device__ __forceinline__ void dft3(float2* v) {
    const float2 t = c_add(v[1], v[2]);
    const float2 m = make_float2(v[0].x - 0.5f * t.x, v[0].y - 0.5f * t.y);
    const float2 n = c_mni(c_scale(c_sub(v[1], v[2]), 0.86602540378443864676f));           // -i (sqrt3 / 2)(v1 - v2)
    v[0] = c_add(v[0], t); v[1] = c_add(m, n); v[2] = c_sub(m, n);
}
__device__ __forceinline__ void dft4(float2* v) {
    const float2 a = c_add(v[0], v[2]), b = c_sub(v[0], v[2]), c = c_add(v[1], v[3]), d = c_mni(c_sub(v[1], v[3]));
    v[0] = c_add(a, c); v[1] = c_add(b, d); v[2] = c_sub(a, c); v[3] = c_sub(b, d);
}
__device__ __forceinline__ void dft5(float2* v) {
    const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f, s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
    const float2 t1 = c_add(v[1], v[4]), t2 = c_add(v[2], v[3]), t3 = c_sub(v[1], v[4]), t4 = c_sub(v[2], v[3]);
    const float2 m1 = make_float2(v[0].x + c1 * t1.x + c2 * t2.x, v[0].y + c1 * t1.y + c2 * t2.y);
    const float2 m2 = make_float2(v[0].x + c2 * t1.x + c1 * t2.x, v[0].y + c2 * t1.y + c1 * t2.y);
    const float2 n1 = c_mni(make_float2(s1 * t3.x + s2 * t4.x, s1 * t3.y + s2 * t4.y));
    const float2 n2 = c_mni(make_float2(s2 * t3.x - s1 * t4.x, s2 * t3.y - s1 * t4.y));
    v[0] = c_add(v[0], c_add(t1, t2));
    v[1] = c_add(m1, n1); v[4] = c_sub(m1, n1);
    v[2] = c_add(m2, n2); v[3] = c_sub(m2, n2);
}

// One Stockham pass of radix R over the wave's N-point sequence: butterfly j reads in[j + t N/R], twists input t by
// exp(-2 pi i t k / (Ns R)) with k = j mod Ns, transforms, and writes out[(j - k) R + k + t Ns]; after the last pass the
// sequence is in natural order.
template <int R>
__device__ __forceinline__ void stockham_pass(const float2* __restrict__ in, float2* __restrict__ out, const float2* __restrict__ tw, int N, int Ns, int lane) {
    const int M = N / R, step = N / (Ns * R);
    const float inv_ns = 1.0f / (float)Ns;
    for (int j = lane; j < M; j += 64) {
        const int k = j - (int)(((float)j + 0.5f) * inv_ns) * Ns;      // j mod Ns (exact: j < 2^11, the quotient is never within 2^-12 of an integer)
        float2 v[R];
#pragma unroll
        for (int t = 0; t < R; ++t) {
            v[t] = in[j + t * M];
            if (t > 0 && Ns > 1) v[t] = c_mul(v[t], tw[t * k * step]);      // t k step < N
        }
        if (R == 2) dft2(v); else if (R == 3) dft3(v); else if (R == 4) dft4(v); else dft5(v);
        const int j0 = (j - k) * R + k;
#pragma unroll
        for (int t = 0; t < R; ++t) out[j0 + t * Ns] = v[t];
    }
}

// The same pass with N and Ns known at compile time (the EfficientNet frontend's 800 points): the butterflies of a lane are unrolled, so
// their LDS reads go out together instead of one wait per butterfly, and the index arithmetic is constants.  Same operations in the same
// order as stockham_pass: the same bits.
template <int R, int N, int Ns>
__device__ __forceinline__ void stockham_pass_ct(const float2* __restrict__ in, float2* __restrict__ out, const float2* __restrict__ tw, int lane) {
    constexpr int M = N / R, step = N / (Ns * R), NIT = (M + 63) / 64;
    float2 v[NIT][R];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int j = lane + 64 * it;
        if (M % 64 == 0 || j < M) {
#pragma unroll
            for (int t = 0; t < R; ++t) v[it][t] = in[j + t * M];
        }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int j = lane + 64 * it;
        if (M % 64 == 0 || j < M) {
            const int k = j % Ns;
            if (Ns > 1) {
#pragma unroll
                for (int t = 1; t < R; ++t) v[it][t] = c_mul(v[it][t], tw[t * k * step]);
            }
            if (R == 2) dft2(v[it]); else if (R == 3) dft3(v[it]); else if (R == 4) dft4(v[it]); else dft5(v[it]);
            const int j0 = (j - k) * R + k;
#pragma unroll
            for (int t = 0; t < R; ++t) out[j0 + t * Ns] = v[it][t];
        }
    }
}

// The same pass IN PLACE (round 5): a wave's LDS instructions execute in order and this form issues every read of the pass before its first
// write, so one N-point buffer per wave is enough -- the ping-pong partner was 25 KB of the workgroup's 79 KB and held the kernel to two
// workgroups (two waves per SIMD) per CU for a chain of dependent LDS round trips.  Same operations in the same order: the same bits.
template <int R, int N, int Ns>
__device__ __forceinline__ void stockham_pass_ip(float2* buf, const float2* __restrict__ tw, int lane) {
    constexpr int M = N / R, step = N / (Ns * R), NIT = (M + 63) / 64;
    float2 v[NIT][R];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int j = lane + 64 * it;
        if (M % 64 == 0 || j < M) {
#pragma unroll
            for (int t = 0; t < R; ++t) v[it][t] = buf[j + t * M];
        }
    }
    asm volatile("" ::: "memory");                       // (compiler) no read of the buffer moves below this line
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int j = lane + 64 * it;
        if (M % 64 == 0 || j < M) {
            const int k = j % Ns;
            if (Ns > 1) {
#pragma unroll
                for (int t = 1; t < R; ++t) v[it][t] = c_mul(v[it][t], tw[t * k * step]);
            }
            if (R == 2) dft2(v[it]); else if (R == 3) dft3(v[it]); else if (R == 4) dft4(v[it]); else dft5(v[it]);
            const int j0 = (j - k) * R + k;
#pragma unroll
            for (int t = 0; t < R; ++t) buf[j0 + t * Ns] = v[it][t];
        }
    }
}

// STFT power spectra by FFT: every wave transforms frame PAIRS packed as one complex sequence (x_a + i x_b), 4 pairs per wave,
// FB = 8 * waves frames per workgroup.  A wave finishes its two frames on its own -- power spectra into the idle half of its
// ping-pong buffer, mel (CSR) / log into a [bin][frame] staging tile -- so the only workgroup-wide step is the final coalesced store
// of that tile (128-byte rows of [clip][bin][frame]) with the per-clip min / max.  LDS: twiddles 8 N + 16 N per wave + the staging
// tile (n_out x (FB + 1) floats): 74 KB for the EfficientNet setting (800 points, 128 mels), two workgroups per CU -- with the run-time
// passes.  The compiled-in 800-point passes run IN PLACE (stockham_pass_ip, round 5): 8 N per wave, 49 + 5 KB with the mel bank, THREE
// workgroups (three waves per SIMD; the kernel takes 153 registers) per CU for what is a chain of dependent LDS round trips:
// 0.80 -> 0.61 ms per 256 clips, the same bits.
#ifndef STFT_KNOCK
#define STFT_KNOCK 0      // diagnostic builds: bit 0 no FFT passes, bit 1 no spectrum split, bit 2 no mel / log stage, bit 3 no sample loads (timing only, wrong results)
#endif
template <int FB, int NF = 0>      // NF: n_fft when its passes are compiled in (800: radices 5 5 4 4 2, the planner's order), 0 = any length at run time
__global__ __launch_bounds__(FB * 8) void stft_fft_kernel(MelDev md, const float* __restrict__ wav, int64_t T, int64_t stride, int frames,
                                                          float* __restrict__ out, int* __restrict__ minmax, int take_log) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NW = FB / 8, SLD = FB + 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = md.n_fft, nf = N / 2 + 1;
    const int b = blockIdx.y, f0 = blockIdx.x * FB;
    float2* tw = (float2*)smem;                                   // [N]
    constexpr bool INPL = NF == 800;                              // compiled-in passes run in place: one buffer per wave
    constexpr int NBUF = INPL ? 1 : 2;
    float2* buf = tw + N + (size_t)wave * NBUF * N;               // this wave's buffer(s): ping-pong for the run-time passes
    float* stage = (float*)(tw + N + (size_t)NW * NBUF * N);      // [n_out][FB + 1]
    for (int i = tid; i < ((STFT_KNOCK & 16) ? 0 : N); i += FB * 8) tw[i] = md.tw[i];
    // mel bank in LDS (planner: mel_lds): [n_out] start, [n_out] length, [n_out] offset, [nnz] weights
    int* lb = (int*)(stage + (size_t)md.n_out * SLD);
    const int* m_start = md.mel_start; const int* m_len = md.mel_len; const int* m_off = md.mel_off; const float* m_w = md.mel_w;
    if (md.mel_lds) {
        for (int i = tid; i < md.n_out; i += FB * 8) { lb[i] = md.mel_start[i]; lb[md.n_out + i] = md.mel_len[i]; lb[2 * md.n_out + i] = md.mel_off[i]; }
        float* lw = (float*)(lb + 3 * md.n_out);
        for (int i = tid; i < md.mel_nnz; i += FB * 8) lw[i] = md.mel_w[i];
        m_start = lb; m_len = lb + md.n_out; m_off = lb + 2 * md.n_out; m_w = lw;
    }
    __syncthreads();
    const float* src = wav + (int64_t)b * stride;
    for (int pr = 0; pr < 4; ++pr) {
        const int fl = wave * 8 + 2 * pr, fa = f0 + fl;           // frames fa, fa + 1 (block-local fl, fl + 1)
        float2* A = buf;
        float2* Bf = INPL ? buf : buf + N;
        // window, reflect padding (torch.stft center=True), pack
        const int64_t base = (int64_t)fa * md.hop - (md.center ? N / 2 : 0);
        bool nza = false, nzb = false;                            // an all-zero frame beside a loud partner: see fbank.hip
        // (all but the first and last few frame pairs of a clip lie inside it: plain loads at 32-bit offsets.  The reflecting form below
        // with its 64-bit index arithmetic per sample was 0.47 of the kernel's 1.30 ms per 256 clips)
        const bool interior = base >= 0 && base + md.hop + N <= T && fa + 1 < frames;
        if (interior && ((N | md.hop) & 3) == 0 && (((uintptr_t)(src + base) | (uintptr_t)md.win) & 15) == 0) {
            // ... four samples per lane and load when everything is 16-byte aligned (hop 160, 800 points, rows of 160 000 samples: always)
            // (requesting the NEXT pair's samples ahead of this pair's passes was measured: no gain, the partner waves cover the round trip)
            const float* s0 = src + base;
            const int hop = md.hop;
            for (int n = 4 * lane; n < ((STFT_KNOCK & 8) ? 0 : N); n += 256) {
                const f32x4 w = *(const f32x4*)(md.win + n), xa = *(const f32x4*)(s0 + n), xb = *(const f32x4*)(s0 + n + hop);
                const f32x4 za = xa * w, zb = xb * w;
                *(f32x4*)(A + n) = (f32x4){za[0], zb[0], za[1], zb[1]};
                *(f32x4*)(A + n + 2) = (f32x4){za[2], zb[2], za[3], zb[3]};
                nza = nza || za[0] != 0.f || za[1] != 0.f || za[2] != 0.f || za[3] != 0.f;
                nzb = nzb || zb[0] != 0.f || zb[1] != 0.f || zb[2] != 0.f || zb[3] != 0.f;
            }
        } else if (interior) {
            const float* s0 = src + base;
            const int hop = md.hop;
            for (int n = lane; n < N; n += 64) {
                const float w = md.win[n];
                const float2 z = make_float2(s0[n] * w, s0[n + hop] * w);
                A[n] = z;
                nza = nza || z.x != 0.f;
                nzb = nzb || z.y != 0.f;
            }
        } else
        for (int n = lane; n < N; n += 64) {
            float2 z;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                int64_t j = base + (int64_t)h * md.hop + n;
                if (j < 0) j = -j;
                if (j >= T) j = 2 * (T - 1) - j;
                if (j < 0) j = 0;
                const float x = (fa + h < frames && j < T) ? src[j] * md.win[n] : 0.f;
                if (h == 0) z.x = x; else z.y = x;
            }
            A[n] = z;
            nza = nza || z.x != 0.f;
            nzb = nzb || z.y != 0.f;
        }
        const bool live_a = __any(nza), live_b = __any(nzb);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // wave-private buffers: a wave's LDS operations execute in order
        if constexpr (NF == 800 && !(STFT_KNOCK & 1)) {
            stockham_pass_ip<5, 800, 1>(A, tw, lane);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            stockham_pass_ip<5, 800, 5>(A, tw, lane);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            stockham_pass_ip<4, 800, 25>(A, tw, lane);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            stockham_pass_ip<4, 800, 100>(A, tw, lane);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            stockham_pass_ip<2, 800, 400>(A, tw, lane);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        int Ns = 1;
        for (int p = 0; p < (NF ? 0 : md.n_pass); ++p) {
            const int R = md.radix[p];
            if (R == 5) stockham_pass<5>(A, Bf, tw, N, Ns, lane);
            else if (R == 4) stockham_pass<4>(A, Bf, tw, N, Ns, lane);
            else if (R == 3) stockham_pass<3>(A, Bf, tw, N, Ns, lane);
            else stockham_pass<2>(A, Bf, tw, N, Ns, lane);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            float2* t_ = A; A = Bf; Bf = t_;
            Ns *= R;
        }
        // split the two real spectra (conjugate symmetry); their power goes to the idle buffer (2 N floats >= 2 (N/2 + 2)) -- in place: over
        // the spectrum itself, every bin pair of the lane read before the first power is written (the same in-order argument as the passes)
        float* pa = (float*)Bf;
        float* pb = pa + nf + 1;
        if constexpr (INPL) {
            constexpr int NKI = (800 / 2 + 1 + 63) / 64;
            float2 zk[NKI], zn[NKI];
#pragma unroll
            for (int i = 0; i < NKI; ++i) {
                const int k = lane + 64 * i;
                if (k < nf && !(STFT_KNOCK & 2)) { zk[i] = A[k]; zn[i] = A[k == 0 ? 0 : N - k]; }
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int i = 0; i < NKI; ++i) {
                const int k = lane + 64 * i;
                if (k < nf && !(STFT_KNOCK & 2)) {
                    const float ar = 0.5f * (zk[i].x + zn[i].x), ai = 0.5f * (zk[i].y - zn[i].y);
                    const float br = 0.5f * (zk[i].y + zn[i].y), bi = -0.5f * (zk[i].x - zn[i].x);
                    pa[k] = live_a ? ar * ar + ai * ai : 0.f;
                    pb[k] = live_b ? br * br + bi * bi : 0.f;
                }
            }
        } else
        for (int k = lane; k < ((STFT_KNOCK & 2) ? 0 : nf); k += 64) {
            const float2 zk = A[k], zn = A[k == 0 ? 0 : N - k];
            const float ar = 0.5f * (zk.x + zn.x), ai = 0.5f * (zk.y - zn.y);
            const float br = 0.5f * (zk.y + zn.y), bi = -0.5f * (zk.x - zn.x);
            pa[k] = live_a ? ar * ar + ai * ai : 0.f;
            pb[k] = live_b ? br * br + bi * bi : 0.f;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // mel (CSR) or plain bins, log -> staging tile
        for (int m = lane; m < ((STFT_KNOCK & 4) ? 0 : md.n_out); m += 64) {
            float ea, eb;
            if (md.use_mel) {
                const int st = m_start[m], len = m_len[m], off = m_off[m];
                ea = 0.f; eb = 0.f;
                // four taps requested together (weights from L1 / L2, spectra from LDS) before their multiply-adds: one wait per four taps
                // instead of one per tap (the widest filters of 128 mels over 401 bins have 18).  Same order of additions.
                for (int q0 = 0; q0 < len; q0 += 4) {
                    float w[4], xa[4], xb[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const bool on = q0 + u < len;
                        w[u] = on ? m_w[off + q0 + u] : 0.f;
                        xa[u] = on ? pa[st + q0 + u] : 0.f;
                        xb[u] = on ? pb[st + q0 + u] : 0.f;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (q0 + u < len) { ea = __builtin_fmaf(xa[u], w[u], ea); eb = __builtin_fmaf(xb[u], w[u], eb); }
                    }
                }
            } else {
                ea = pa[m]; eb = pb[m];
            }
            stage[m * SLD + fl] = take_log ? logf(ea + 1e-6f) : ea;
            stage[m * SLD + fl + 1] = take_log ? logf(eb + 1e-6f) : eb;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __syncthreads();
    // coalesced store [clip][bin][frame] (FB consecutive frames per row) + per-clip min / max
    float mn = __builtin_inff(), mx = -__builtin_inff();
    const int f = tid % FB, fr_ok = f0 + f < frames;
#pragma unroll 4
    for (int m = tid / FB; m < ((STFT_KNOCK & 32) ? 0 : md.n_out); m += (FB * 8) / FB) {
        const float y = stage[m * SLD + f];
        if (fr_ok) {
            out[((int64_t)b * md.n_out + m) * frames + f0 + f] = y;
            mn = fminf(mn, y); mx = fmaxf(mx, y);
        }
    }
    if (minmax && !(STFT_KNOCK & 64)) {
        // one pair of atomics per workgroup (through the staging tile, which every wave has finished reading), not per wave: 65 k atomics
        // on 512 addresses were 44 us of the kernel
        mn = -wave_max(-mn); mx = wave_max(mx);
        __syncthreads();
        if (lane == 0) { stage[2 * wave] = mn; stage[2 * wave + 1] = mx; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < NW; ++w) { mn = fminf(mn, stage[2 * w]); mx = fmaxf(mx, stage[2 * w + 1]); }
            if (mx >= mn) { atomicMin(minmax + 2 * b, ord_key(mn)); atomicMax(minmax + 2 * b + 1, ord_key(mx)); }
        }
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
    size_t lds = 0;          // LDS bytes of the kernel the plan launches by default
    size_t lds_pp = 0;       // ... of the ping-pong (run-time passes) FFT kernel, when `inplace` selects the in-place one and an A/B run asks for the other
    bool inplace = false;    // FFT path: the compiled-in 800-point passes, one buffer per wave
    bool fft = false;        // mixed-radix FFT path (n_fft = 2^a 3^b 5^c <= 2048) instead of the dense fp32-MFMA product
};

extern "C" avexhip_melspec_plan* avexhip_melspec_plan_create(const avexhip_melspec_config* cfg, const float* window, const float* mel_fb) {
    if (!cfg || !window) { avexhip_set_error("melspec_plan_create: null argument"); return nullptr; }
    const int N = cfg->n_fft, win = cfg->win_length, hop = cfg->hop_length, nf = N / 2 + 1;
    if (N < 64 || N > 2048 || (N & 1) || win <= 0 || win > N || hop <= 0 || cfg->n_mels < 0 || cfg->n_mels > 512) {
        avexhip_set_error("melspec_plan_create: unsupported n_fft=%d win=%d hop=%d n_mels=%d (even n_fft in 64..2048, win <= n_fft)", N, win, hop, cfg->n_mels);
        return nullptr;
    }
    if (cfg->n_mels > 0 && !mel_fb) { avexhip_set_error("melspec_plan_create: mel_fb missing"); return nullptr; }
    // n_fft = 2^a 3^b 5^c: mixed-radix FFT in LDS (the reference's defaults: 2048 (AudioConfig), 800 (EfficientNet), 512, 1024, 400);
    // anything else (a factor 7, 11, ...) falls back to the dense fp32-MFMA product, which is built for n_fft <= 1024 only
    std::vector<int> radices;
    {
        int m = N;
        for (int pr : {5, 3}) while (m % pr == 0) { radices.push_back(pr); m /= pr; }
        while (m % 4 == 0) { radices.push_back(4); m /= 4; }
        while (m % 2 == 0) { radices.push_back(2); m /= 2; }
        if (m != 1 || radices.size() > 12) radices.clear();
    }
    const char* force_dense = getenv("AVEX_AMD_MELSPEC_DENSE");
    const bool use_fft = !radices.empty() && !(force_dense && atoi(force_dense) != 0 && N <= 1024);
    const int half = ((nf + 63) / 64) * 64, nc = 2 * half, ntw = half / 64;
    if (!use_fft && (N > 1024 || (ntw != 3 && ntw != 4 && ntw != 5 && ntw != 7 && ntw != 9))) {
        avexhip_set_error("melspec_plan_create: n_fft=%d has a prime factor above 5 and its dense column tiling %d is not instantiated "
                          "(dense path: n_fft 256..384, 400..512, 514..640, 770..896, 1026..1024)", N, ntw);
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
    std::vector<float> dft(use_fft ? 0 : (size_t)kfold * nc, 0.f);
    std::vector<float2> twid(use_fft ? N : 0);
    std::vector<float> wfull(use_fft ? N : 0);
    for (int k = 0; k < (int)twid.size(); ++k) {
        const double a = -2.0 * M_PI * (double)k / (double)N;
        twid[k] = make_float2((float)cos(a), (float)sin(a));
        wfull[k] = wat(k);
    }
    for (int n = 0; !use_fft && n < (sym ? N / 2 + 1 : N); ++n) {
        const float w = wat(n);
        for (int k = 0; k < nf; ++k) {
            const long long kn = ((long long)k * n) % N;             // exact argument reduction
            const double a = -2.0 * M_PI * (double)kn / (double)N;
            dft[(size_t)n * nc + k] = (float)((double)w * cos(a));
            dft[(size_t)n * nc + half + k] = (float)((double)w * sin(a));
        }
    }
    const size_t o_dft = 0, o_st = sizeof(float) * dft.size(), o_ln = o_st + sizeof(int) * n_out, o_of = o_ln + sizeof(int) * n_out,
                 o_w = o_of + sizeof(int) * n_out, o_tw = (o_w + sizeof(float) * (packed.size() + 1) + 15) & ~(size_t)15,
                 o_win = o_tw + sizeof(float2) * twid.size(), total = o_win + sizeof(float) * (wfull.size() + 1);
    std::vector<char> host(total, 0);
    if (!dft.empty()) memcpy(host.data() + o_dft, dft.data(), sizeof(float) * dft.size());
    if (!twid.empty()) { memcpy(host.data() + o_tw, twid.data(), sizeof(float2) * twid.size()); memcpy(host.data() + o_win, wfull.data(), sizeof(float) * wfull.size()); }
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
    p->fft = use_fft;
    md.n_pass = (int)radices.size();
    for (int i = 0; i < 12; ++i) md.radix[i] = i < md.n_pass ? radices[i] : 1;
    md.fb = (sizeof(float2) * ((size_t)N + 4 * 2 * (size_t)N) + sizeof(float) * (size_t)n_out * 33 <= 160 * 1024) ? 32 : 16;
    md.tw = (const float2*)(base + o_tw); md.win = (const float*)(base + o_win);
    const int nseg = hop * 31 + N;
    const int xs_words = nseg + (md.skew ? nseg / hop + 1 : 0);
    md.mel_nnz = (int)packed.size(); md.mel_lds = 0;
    if (use_fft) {
        p->inplace = md.fb == 32 && N == 800 && radices.size() == 5 && radices[0] == 5 && radices[1] == 5 && radices[2] == 4 && radices[3] == 4 && radices[4] == 2;
        p->lds_pp = sizeof(float2) * ((size_t)N + (size_t)(md.fb / 8) * 2 * N) + sizeof(float) * (size_t)n_out * (md.fb + 1);
        p->lds = p->inplace ? p->lds_pp - sizeof(float2) * (size_t)(md.fb / 8) * N : p->lds_pp;
        // the mel bank (CSR: start / length / offset per mel + packed weights, ~5 KB at 128 mels over 401 bins) beside them when the
        // workgroups per CU stay the same: its loads in the mel stage are then LDS reads instead of dependent trips to L1 / L2
        const size_t bank = sizeof(int) * 3 * (size_t)n_out + sizeof(float) * packed.size();
        if (md.use_mel && (160 * 1024) / (p->lds + bank) == (160 * 1024) / p->lds) { md.mel_lds = 1; p->lds += bank; p->lds_pp += bank; }
    }
    else p->lds = sizeof(float) * (((xs_words + 3) & ~3) + 32 * (size_t)(half + 1));
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
    const int take_log = p->cfg.normalize ? 1 : 0;
    if (p->fft) {
        const int FB = p->dev.fb;
        const dim3 gridf((frames + FB - 1) / FB, B);
        if (FB == 32) {
            static const bool generic = getenv("AVEX_AMD_STFT_GENERIC") && atoi(getenv("AVEX_AMD_STFT_GENERIC")) != 0;      // A/B: the run-time passes for 800 points too
            if (p->inplace && !generic) {
                AVX_ENSURE_LDS((stft_fft_kernel<32, 800>), 160 * 1024);
                hipLaunchKernelGGL((stft_fft_kernel<32, 800>), gridf, dim3(256), p->lds, s, p->dev, wav_dev, T, wav_stride, frames, out_dev, mm, take_log);
            } else {
                AVX_REQUIRE(p->lds_pp <= 160 * 1024, "melspec_forward: the run-time FFT passes need %zu bytes of LDS", p->lds_pp);
                AVX_ENSURE_LDS(stft_fft_kernel<32>, 160 * 1024);
                hipLaunchKernelGGL(stft_fft_kernel<32>, gridf, dim3(256), p->inplace ? p->lds_pp : p->lds, s, p->dev, wav_dev, T, wav_stride, frames, out_dev, mm, take_log);
            }
        } else {
            AVX_ENSURE_LDS(stft_fft_kernel<16>, 160 * 1024);
            hipLaunchKernelGGL(stft_fft_kernel<16>, gridf, dim3(128), p->lds, s, p->dev, wav_dev, T, wav_stride, frames, out_dev, mm, take_log);
        }
        AVX_LAUNCH_CHECK();
        if (mm) {
            const int64_t per_clip = (int64_t)p->dev.n_out * frames;
            hipLaunchKernelGGL(melspec_norm_kernel, dim3(64, B), dim3(256), 0, s, out_dev, per_clip, mm);
            AVX_LAUNCH_CHECK();
        }
        return AVEXHIP_OK;
    }
    const dim3 grid((frames + 31) / 32, B);
    const int ntw = p->dev.half / 64;
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
