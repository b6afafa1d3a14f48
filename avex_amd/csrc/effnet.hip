// EfficientNet-B0 building blocks that are not GEMMs (avex/models/efficientnet.py:55-66 -> torchvision efficientnet_b0),
// for gfx950.  Activations are NHWC in the operand type with the channel count padded to a multiple of 128 (padding channels
// stay exactly zero through every layer), so the 1x1 convolutions are plain avexhip_gemm calls over [B*H*W, C] rows with
// BatchNorm folded into weight and bias and SiLU / residual in the epilogue.  What remains is bandwidth-bound:
//   stem_conv      Conv2d(3 -> 32, 3x3, stride 2, pad 1) on the mel image whose three input channels are copies of one
//                  (efficientnet.py:133-135: x.repeat(1, 3, 1, 1)), i.e. a 1-channel convolution with channel-summed weights,
//                  + folded BatchNorm + SiLU, fp32 [B, H, W] in -> NHWC half out
//   dwconv         depthwise k x k (3 or 5), stride 1 or 2, "same" padding, folded BatchNorm + SiLU, and the
//                  squeeze-excitation average pool accumulated on the way out (one row of partial sums per workgroup, added in order)
//   se_fc          the two tiny fully connected layers of squeeze-excitation per clip: sigmoid(W2 silu(W1 mean + b1) + b2)
//   scale_channels x[b, p, c] *= s[b, c] (the excitation), in place
#include "common.h"

#ifndef MBCONV_MODE_TOGGLE
#define MBCONV_MODE_TOGGLE 1
#endif
namespace {

__device__ __forceinline__ float silu1(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }

// one thread: STEM_PIX consecutive output pixels of a row x 8 consecutive output channels.  With one pixel per thread the 18 weight loads of
// 16 bytes per thread (9.4 GB through the L1s per 256 clips) bound the kernel at 0.50 ms; four pixels share them.  Every output adds its
// taps in (ky, kx) order with zeros for the taps outside the image: the same values as skipping them.
constexpr int STEM_PIX = 4;
template <typename T>
__global__ __launch_bounds__(256) void stem_conv_kernel(const float* __restrict__ img, int H, int W, int Ho, int Wo,
                                                        const float* __restrict__ w /*[9][Cp]*/, const float* __restrict__ bias,
                                                        int Cp, T* __restrict__ out, float* __restrict__ raw /*[B,Ho,Wo,Cp] or null*/) {
    AVX_F16_SATURATE_ON();
    typedef typename Half<T>::v8 v8;
    constexpr int SP = STEM_PIX;
    const int cg = Cp >> 3;
    const int xg = (Wo + SP - 1) / SP;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (idx >= (int64_t)Ho * xg * cg) return;
    const int c8 = (int)(idx % cg) * 8;
    const int64_t t = idx / cg;
    const int oy = (int)(t / xg), ox0 = (int)(t - (int64_t)oy * xg) * SP;
    float acc[SP][8];
#pragma unroll
    for (int p = 0; p < SP; ++p)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[p][e] = 0.f;
    const float* src = img + (int64_t)b * H * W;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy * 2 + ky - 1;
        if (iy < 0 || iy >= H) continue;
        float xin[2 * SP + 1];
#pragma unroll
        for (int q = 0; q < 2 * SP + 1; ++q) {
            const int ix = ox0 * 2 - 1 + q;
            xin[q] = (ix >= 0 && ix < W) ? src[(int64_t)iy * W + ix] : 0.f;
        }
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const f32x4 w0 = *(const f32x4*)(w + (ky * 3 + kx) * Cp + c8), w1 = *(const f32x4*)(w + (ky * 3 + kx) * Cp + c8 + 4);
#pragma unroll
            for (int p = 0; p < SP; ++p) {
                const float x = xin[2 * p + kx];
#pragma unroll
                for (int e = 0; e < 4; ++e) { acc[p][e] = __builtin_fmaf(x, w0[e], acc[p][e]); acc[p][4 + e] = __builtin_fmaf(x, w1[e], acc[p][4 + e]); }
            }
        }
    }
    const f32x4 b0 = *(const f32x4*)(bias + c8), b1 = *(const f32x4*)(bias + c8 + 4);
#pragma unroll
    for (int p = 0; p < SP; ++p) {
        if (ox0 + p >= Wo) break;
        const int64_t o = (((int64_t)b * Ho + oy) * Wo + ox0 + p) * Cp + c8;
        v8 h;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float y = acc[p][e] + (e < 4 ? b0[e] : b1[e - 4]);
            if (raw) raw[o + e] = y;
            h[e] = Half<T>::from_hw(silu1(y));
        }
        *(v8*)(out + o) = h;
    }
}

#ifndef DW_PIX
#define DW_PIX 8      // output pixels along x per thread
#endif
#ifndef DW_PY
#define DW_PY 1       // output rows per thread
#endif
#ifndef DW_ROLL
#define DW_ROLL 0     // 1: the loop over input rows stays rolled (one row of loads live at a time: fewer registers, more waves per SIMD)
#endif
#ifndef DW_WAVES
#define DW_WAVES 0    // > 0: amdgpu_waves_per_eu for dwconv_kernel
#endif
// one thread: 8 channels, PIX consecutive output pixels along x of PY consecutive output rows; squeeze sums leave as per-workgroup partials.  The kernel is
// bound by its 16-byte loads from L2 (every input element is wanted by KS x KS outputs): a 1 x 4 tile makes 4.5 loads per output at 3 x 3
// (13.5 ms per 256 clips of EfficientNet-B0 at the time), 1 x 8 3.75 (12.8 ms); wider tiles lose to their registers (1 x 12: 13.2, 1 x 16:
// 14.1 ms), and two-row tiles, fewer loads still, do not pay either (on the final tree: 1 x 8 10.5 ms, 2 x 4 11.0, 2 x 6 10.6, 2 x 8 10.9).
// Every output adds its taps in (ky, kx) order whatever the tile: the tile shape does not change a bit of the result.
template <typename T, int KS, int ST>
__global__ __launch_bounds__(256)
#if DW_WAVES
__attribute__((amdgpu_waves_per_eu(DW_WAVES, DW_WAVES)))
#endif
void dwconv_kernel(const T* __restrict__ in, int H, int W, int Ho, int Wo, int Cp,
                                                     const float* __restrict__ w /*[KS*KS][Cp]*/, const float* __restrict__ bias,
                                                     T* __restrict__ out, float* __restrict__ part /*[B][gridDim.x][Cp]*/) {
    AVX_F16_SATURATE_ON();
    typedef typename Half<T>::v8 v8;
    constexpr int PAD = (KS - 1) / 2, PIX = DW_PIX, PY = DW_PY;
    constexpr int NCOL = (PIX - 1) * ST + KS, NROW = (PY - 1) * ST + KS;
    const int cg = Cp >> 3;
    const int xg = (Wo + PIX - 1) / PIX, yg = (Ho + PY - 1) / PY;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    const bool active = idx < (int64_t)yg * xg * cg;
    const int c8 = (int)(idx % cg) * 8;
    const int64_t t = idx / cg;
    const int oy0 = (int)(t / xg) * PY, ox0 = (int)(t - (int64_t)(t / xg) * xg) * PIX;
    float psum[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) psum[e] = 0.f;
    if (active) {
        float acc[PY][PIX][8];
#pragma unroll
        for (int py = 0; py < PY; ++py)
#pragma unroll
            for (int p = 0; p < PIX; ++p)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[py][p][e] = 0.f;
        const T* src = in + (int64_t)b * H * W * Cp + c8;
#if DW_ROLL
#pragma unroll 1
#else
#pragma unroll
#endif
        for (int r = 0; r < NROW; ++r) {
            const int iy = oy0 * ST - PAD + r;
            if (iy < 0 || iy >= H) continue;
            v8 col[NCOL];
#pragma unroll
            for (int q = 0; q < NCOL; ++q) {
                const int ix = ox0 * ST - PAD + q;
                if (ix >= 0 && ix < W) col[q] = *(const v8*)(src + ((int64_t)iy * W + ix) * Cp);
                else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) col[q][e] = (T)0.0f;
                }
            }
#pragma unroll
            for (int py = 0; py < PY; ++py) {
                const int ky = r - py * ST;                 // compile-time after unrolling
                if (ky < 0 || ky >= KS) continue;
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) {
                    const f32x4 w0 = *(const f32x4*)(w + (ky * KS + kx) * Cp + c8), w1 = *(const f32x4*)(w + (ky * KS + kx) * Cp + c8 + 4);
#pragma unroll
                    for (int p = 0; p < PIX; ++p) {
                        const v8 x = col[p * ST + kx];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            acc[py][p][e] = __builtin_fmaf((float)x[e], w0[e], acc[py][p][e]);
                            acc[py][p][4 + e] = __builtin_fmaf((float)x[4 + e], w1[e], acc[py][p][4 + e]);
                        }
                    }
                }
            }
        }
        const f32x4 b0 = *(const f32x4*)(bias + c8), b1 = *(const f32x4*)(bias + c8 + 4);
#pragma unroll
        for (int py = 0; py < PY; ++py) {
            const int oy = oy0 + py;
            if (oy >= Ho) break;
#pragma unroll
            for (int p = 0; p < PIX; ++p) {
                const int ox = ox0 + p;
                if (ox >= Wo) break;
                v8 h;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float y = silu1(acc[py][p][e] + (e < 4 ? b0[e] : b1[e - 4]));
                    h[e] = Half<T>::from_hw(y);
                    psum[e] += (float)h[e];                    // the pool sees what the next layer will read
                }
                *(v8*)(out + (((int64_t)b * Ho + oy) * Wo + ox) * Cp + c8) = h;
            }
        }
    }
    // squeeze: lanes of a wave hold different channel groups when cg < 64, the same group every cg lanes.  Each workgroup leaves ONE row of
    // partial sums (cg <= 256, so its 256 consecutive threads meet every channel group); pool_sum_kernel adds the rows in index order, so the
    // pool - and with it the whole network - repeats bit for bit from run to run (float atomics here made it depend on arrival order).
    if (part) {
        __shared__ float red[256 * 8];
#pragma unroll
        for (int e = 0; e < 8; ++e) red[threadIdx.x * 8 + e] = psum[e];
        __syncthreads();
        // thread j < min(256, cg) * 8 sums one (channel group, e) column over the block's threads with the same idx % cg
        const int ncol = (cg < 256 ? cg : 256) * 8;
        const int first = (int)(((int64_t)blockIdx.x * 256) % cg);      // channel group of thread 0
        for (int j = threadIdx.x; j < ncol; j += 256) {
            const int g = j >> 3, e = j & 7;                            // g: offset from thread 0's group
            float s = 0.f;
            for (int q = g; q < 256; q += cg) s += red[q * 8 + e];
            const int grp = (first + g) % cg;
            part[((int64_t)b * gridDim.x + blockIdx.x) * Cp + grp * 8 + e] = s;
        }
    }
}

// pool[b][c] = sum over the depthwise kernel's workgroups, in workgroup order
__global__ __launch_bounds__(256) void pool_sum_kernel(const float* __restrict__ part, int nblk, int Cp, float* __restrict__ pool) {
    const int b = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= Cp) return;
    const float* p = part + (int64_t)b * nblk * Cp + c;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int i = 0;
    for (; i + 4 <= nblk; i += 4) { s0 += p[(int64_t)i * Cp]; s1 += p[(int64_t)(i + 1) * Cp]; s2 += p[(int64_t)(i + 2) * Cp]; s3 += p[(int64_t)(i + 3) * Cp]; }
    for (; i < nblk; ++i) s0 += p[(int64_t)i * Cp];
    pool[(int64_t)b * Cp + c] = (s0 + s1) + (s2 + s3);
}

// s[b][c] = sigmoid(W2[c,:] . silu(W1 mean_b + b1) + b2[c]);   one workgroup of sixteen waves per clip (a wave per hidden unit at a time:
// with four waves the up to 48 dot products of a clip were a chain of twelve, 34 us per launch, 5 % of an EfficientNet-B0 step)
__global__ __launch_bounds__(1024) void se_fc_kernel(const float* __restrict__ pool, float inv_hw, int C, int Cp, int Cs,
                                                    const float* __restrict__ w1 /*[Cs][C]*/, const float* __restrict__ b1,
                                                    const float* __restrict__ w2 /*[C][Cs]*/, const float* __restrict__ b2,
                                                    float* __restrict__ scale /*[B][Cp]*/) {
    __shared__ float mean[2048];
    __shared__ float hid[512];
    const int b = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += 1024) mean[c] = pool[(int64_t)b * Cp + c] * inv_hw;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int j = wave; j < Cs; j += 16) {
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s = __builtin_fmaf(w1[(int64_t)j * C + c], mean[c], s);
        s = wave_sum(s);
        if (lane == 0) hid[j] = silu1(s + b1[j]);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < Cp; c += 1024) {
        float v = 0.f;
        if (c < C) {
            float s = b2[c];
            for (int j = 0; j < Cs; ++j) s = __builtin_fmaf(w2[(int64_t)c * Cs + j], hid[j], s);
            v = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * s));
        }
        scale[(int64_t)b * Cp + c] = v;
    }
}

// The same squeeze-excitation, taking the depthwise kernels' rows of partial sums directly (pool_sum_kernel's addition order: the same
// pool bit for bit) and the second layer's weights TRANSPOSED ([Cs][C]): thread c's loads of w2[c][j] were Cs floats apart from its
// neighbour's (a 128-byte line per lane and instruction; 32 us per launch at C = 1152, Cs = 48), w2t[j][c] is one line per 32 lanes.
__global__ __launch_bounds__(1024) void se_pool_fc_kernel(const float* __restrict__ part, int nblk, float inv_hw, int C, int Cp, int Cs,
                                                         const float* __restrict__ w1 /*[Cs][C]*/, const float* __restrict__ b1,
                                                         const float* __restrict__ w2t /*[Cs][C]*/, const float* __restrict__ b2,
                                                         float* __restrict__ scale /*[B][Cp]*/) {
    __shared__ float mean[2048];
    __shared__ float hid[512];
    const int b = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += 1024) {
        const float* p = part + (int64_t)b * nblk * Cp + c;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int i = 0;
        for (; i + 4 <= nblk; i += 4) { s0 += p[(int64_t)i * Cp]; s1 += p[(int64_t)(i + 1) * Cp]; s2 += p[(int64_t)(i + 2) * Cp]; s3 += p[(int64_t)(i + 3) * Cp]; }
        for (; i < nblk; ++i) s0 += p[(int64_t)i * Cp];
        mean[c] = ((s0 + s1) + (s2 + s3)) * inv_hw;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // both layers request their weights SIXTEEN AT A TIME before the first multiply-add of the batch: with run-time trip counts hipcc waits
    // for every load where it is used, and the 18 + 48 dependent round trips to L2 were the whole 32 us of a launch at C = 1152, Cs = 48.
    // The order of the additions is unchanged.
    for (int j = wave; j < Cs; j += 16) {
        float s = 0.f;
        const float* wr = w1 + (int64_t)j * C;
        for (int c0 = lane; c0 < C; c0 += 1024) {
            float w[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) w[u] = c0 + 64 * u < C ? wr[c0 + 64 * u] : 0.f;
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (c0 + 64 * u < C) s = __builtin_fmaf(w[u], mean[c0 + 64 * u], s);
        }
        s = wave_sum(s);
        if (lane == 0) hid[j] = silu1(s + b1[j]);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < Cp; c += 1024) {
        float v = 0.f;
        if (c < C) {
            float s = b2[c];
            for (int j0 = 0; j0 < Cs; j0 += 16) {
                float w[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) w[u] = j0 + u < Cs ? w2t[(int64_t)(j0 + u) * C + c] : 0.f;
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    if (j0 + u < Cs) s = __builtin_fmaf(w[u], hid[j0 + u], s);
            }
            v = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * s));
        }
        scale[(int64_t)b * Cp + c] = v;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void scale_channels_kernel(T* __restrict__ x, int64_t hw, int Cp, const float* __restrict__ scale) {
    typedef typename Half<T>::v8 v8;
    const int cg = Cp >> 3;
    const int b = blockIdx.y;
    const int64_t n = hw * cg;
    T* p = x + (int64_t)b * hw * Cp;
    const float* s = scale + (int64_t)b * Cp;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c8 = (int)(i % cg) * 8;
        v8 v = *(v8*)(p + i * 8);
        const f32x4 s0 = *(const f32x4*)(s + c8), s1 = *(const f32x4*)(s + c8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = Half<T>::from((float)v[e] * s0[e]); v[4 + e] = Half<T>::from((float)v[4 + e] * s1[e]); }
        *(v8*)(p + i * 8) = v;
    }
}


// ---------------------------------------------------------------------------------------------
// Fused front of an MBConv block: 1x1 expansion (+ folded BatchNorm + SiLU) and the depthwise convolution after it in ONE kernel, so the
// expanded tensor -- the largest tensor of the network by far at the first stages (96 channels at 64 x 501: 6 MB per clip, written once and
// read once) -- never exists in HBM.  A workgroup owns a TH x TW tile of OUTPUT pixels of one clip:
//   0. the tile's input pixels, halo included ((TH-1) ST + KS rows x (TW-1) ST + KS columns, KIN input channels each), go to LDS once;
//   per chunk of CC = 32 expanded channels:
//   1. expansion on the MFMA: W [32 x KIN] as the A operand, 16 pixels as B, so a lane ends with four consecutive channels of one pixel
//      (one 8-byte LDS write); bias + SiLU + rounding to the operand type exactly as the GEMM epilogue did; pixels outside the image are
//      written as ZERO (the depthwise convolution pads its INPUT, the expanded tensor, with zeros);
//   2. the depthwise taps from LDS in (ky, kx) order (the order dwconv_kernel adds them: same bits), bias + SiLU, 16-byte stores, and the
//      squeeze sums as one row of partials per workgroup (pool_sum_kernel adds the rows in order).
// The halo is expanded by every tile that needs it (1.2 - 1.5 x the SiLUs of the unfused expansion); what is saved is 2 x the expanded
// tensor of HBM traffic.  The kernel is bound by its vector instructions (two transcendentals per SiLU), not by memory.
// Global loads and the vmcnt counter: every parameter a chunk needs (expansion weight fragments, depthwise weights, biases) is requested
// at the top of the PREVIOUS chunk and consumed before that chunk's output stores are issued -- a wait placed after the stores would have
// to be vmcnt(0) (the stores are conditional, the compiler cannot count them) and would drain them: 1 - 2 us per chunk of ~1 us of work.
// ---------------------------------------------------------------------------------------------
// psum[e] += psum[e] of the lane 4 below, then of the lane 8 below, inside each 16-lane row (lanes without a source add 0): lanes 12 .. 15
// of a row end with the row's four channel-group sums.  The DPP operand sits inside the add (v_add_f32_dpp): from the builtin hipcc makes a
// v_mov_b32 0, a v_mov_b32_dpp and an add per step.  One asm block: the hazard recogniser does not look inside inline assembly -- a DPP read
// needs two wait states after the VALU write of its source; the eight independent chains give the second step seven instructions of
// distance, s_nop 1 covers whatever wrote the inputs.
static __device__ __forceinline__ void row_quarter_sums(float (&p)[8]) {
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %2, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %3, %3, %3 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %4, %4, %4 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %5, %5, %5 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %6, %6, %6 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %7, %7, %7 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %2, %2, %2 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %3, %3, %3 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %4, %4, %4 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %5, %5, %5 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %6, %6, %6 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %7, %7, %7 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 0"
        : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]));
}

typedef int a_i32x2m __attribute__((ext_vector_type(2)));
template <int N> struct a_icm { static constexpr int value = N; };
struct MbArgs {
    const void* in; int H, W, ld_in;
    const void* w_exp; int ldw; const float* b_exp;
    const float* w_dw; const float* b_dw;
    void* out; int Ho, Wo, cp_exp;
    float* part;                 // [B][gridDim.x][cp_exp] or NULL
    int tiles_x;
    unsigned int* ovf;
};

template <int KS, int ST, int KIN, int TH, int TW, int CC>
struct MbGeo {
    static constexpr int IH = (TH - 1) * ST + KS, IW = (TW - 1) * ST + KS, NPIX = IH * IW, NPG = (NPIX + 15) / 16, NPX = NPG * 16;
    static constexpr int ISTR = KIN > 0 ? KIN * 2 + 16 : 0;      // bytes per input pixel in LDS (16 bytes of padding: the 16 pixels of an MFMA fragment read meet 16 distinct bank groups)
    static constexpr int ESTR = CC * 2 + 16;       // bytes per expanded pixel
    static constexpr int NWD = KS * KS * CC;       // depthwise weights per chunk (floats), double-buffered
    static constexpr int LDS = NPX * (ISTR + ESTR) + 2 * NWD * 4 + 16 * CC * 4;
    static constexpr int WPE = (3 * LDS <= 160 * 1024 && KIN == 32) ? 3 : 2;      // waves per SIMD the kernel is compiled for (168 or 256 registers): the workgroups per CU the LDS allows; K = 64 needs the 256 (two sets of weight fragments in flight)
};

template <typename T, int KS, int ST, int KIN, int TH, int TW, int PIX, int CC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MbGeo<KS, ST, KIN, TH, TW, CC>::WPE, MbGeo<KS, ST, KIN, TH, TW, CC>::WPE))) void mbconv_kernel(const MbArgs p) {
    // f16 outputs saturate through MODE.FP16_OVFL (common.h): no clamp instructions.  With the bit set the MFMAs treat a NaN operand as 0
    // (scripts/micro/mfma_nan.hip), so it is set per pair of pixel groups, behind their MFMAs and in front of their conversions, and around the
    // depthwise stage (MBCONV_MODE_TOGGLE; 0 = round 4's form, the bit set once at the top: the A side of profiles/r05b_mbconv_mode.txt).
    if (!MBCONV_MODE_TOGGLE) AVX_F16_SATURATE_ON();
    typedef typename Half<T>::v8 v8;
    typedef typename Half<T>::v4 v4;
    typedef MbGeo<KS, ST, KIN, TH, TW, CC> G;
    constexpr int PAD = (KS - 1) / 2, IW = G::IW, NPIX = G::NPIX, NPG = G::NPG, NPX = G::NPX, ISTR = G::ISTR, ESTR = G::ESTR, NWD = G::NWD;
    constexpr int NCOL = (PIX - 1) * ST + KS, SXN = TW / PIX, NCG = CC / 8, NITEM = TH * SXN * NCG;
    constexpr int KST = KIN / 32, MT = CC / 16, NWR = (NWD + 255) / 256;
    static_assert(TW % PIX == 0 && NITEM % 256 == 0 && NCG == 4 && KIN % 32 == 0 && KIN > 0, "mbconv tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_in = smem;
    char* s_exp = smem + NPX * ISTR;
    float* s_w = (float*)(s_exp + NPX * ESTR);       // [2][NWD]
    float* s_red = s_w + 2 * NWD;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y;
    const int ty = blockIdx.x / p.tiles_x, tx = blockIdx.x - ty * p.tiles_x;
    const int oy0 = ty * TH, ox0 = tx * TW, iy0 = oy0 * ST - PAD, ix0 = ox0 * ST - PAD;
    const T* in = (const T*)p.in + (int64_t)b * p.H * p.W * p.ld_in;
    T* out = (T*)p.out + (int64_t)b * p.Ho * p.Wo * p.cp_exp;
    const int cp = p.cp_exp;
    const int lr = lane & 15, lq = lane >> 4, cg = tid & (NCG - 1);
    float ovf_mx = 0.f;

    // parameters of a chunk: expansion weight fragments + bias (this lane's MFMA operands), depthwise bias of the lane's channel group, and
    // the thread's share of the chunk's depthwise weights on their way to LDS
    struct Par { v8 wf[MT][KST]; f32x4 be[MT]; f32x4 bd0, bd1; float wd[NWR]; };
    auto request = [&](Par& q, int c) __attribute__((always_inline)) {
        const T* W = (const T*)p.w_exp;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int ks = 0; ks < KST; ++ks) q.wf[mt][ks] = *(const v8*)(W + (int64_t)(c + 16 * mt + lr) * p.ldw + ks * 32 + lq * 8);
            q.be[mt] = *(const f32x4*)(p.b_exp + c + 16 * mt + 4 * lq);
        }
        q.bd0 = *(const f32x4*)(p.b_dw + c + cg * 8);
        q.bd1 = *(const f32x4*)(p.b_dw + c + cg * 8 + 4);
#pragma unroll
        for (int r = 0; r < NWR; ++r) {
            const int i = tid + 256 * r;
            q.wd[r] = i < NWD ? p.w_dw[(int64_t)(i / CC) * cp + c + (i % CC)] : 0.f;
        }
    };
    auto stage_wd = [&](const Par& q, int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < NWR; ++r) {
            const int i = tid + 256 * r;
            if (i < NWD) s_w[slot * NWD + i] = q.wd[r];
        }
    };
    Par cur;
    request(cur, 0);
    {
        constexpr int NCH = KIN / 8;
        for (int i = tid; i < NPX * NCH; i += 256) {
            const int pix = i / NCH, ch = i - pix * NCH;
            const int iy = pix / IW, ix = pix - iy * IW;
            const int gy = iy0 + iy, gx = ix0 + ix;
            uint4 v = {0u, 0u, 0u, 0u};
            if (pix < NPIX && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) v = *(const uint4*)(in + ((int64_t)gy * p.W + gx) * p.ld_in + ch * 8);
            *(uint4*)(s_in + pix * ISTR + ch * 16) = v;
        }
    }
    stage_wd(cur, 0);
    // which of this lane's pixels (pixel group wave + 4 j, pixel lr of it) lie inside the image: the same for every chunk
    constexpr int NJ = (NPG + 3) / 4;
    static_assert(NJ <= 32, "mbconv: pixel groups per wave");
    unsigned inside_bits = 0u;
#pragma unroll 1
    for (int j = 0; j < NJ; ++j) {
        const int pix = (wave + 4 * j) * 16 + lr;
        const int iy = pix / IW, ix = pix - iy * IW;
        const int gy = iy0 + iy, gx = ix0 + ix;
        if (pix < NPIX && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) inside_bits |= 1u << j;
    }
    __syncthreads();                                     // the input tile and chunk 0's depthwise weights

    int slot = 0;
    for (int c = 0; c < cp; c += CC, slot ^= 1) {
        const bool more = c + CC < cp;
        Par nxt;
        if (more) request(nxt, c + CC);                  // in flight under the expansion below
        // NG (1 or 2) pixel groups at a time: their MFMAs first, then -- MODE.FP16_OVFL set -- bias, SiLU and the conversions of both (two
        // independent chains for the transcendentals), then the bit cleared again for the next groups' MFMAs
        auto expand_groups = [&](int j0, auto NGc) __attribute__((always_inline)) {
            constexpr int NG = decltype(NGc)::value;
            f32x4 accs[NG][MT];
#pragma unroll
            for (int u = 0; u < NG; ++u) {
                const int pix = (wave + 4 * (j0 + u)) * 16 + lr;
                v8 xf[KST];
#pragma unroll
                for (int ks = 0; ks < KST; ++ks) xf[ks] = *(const v8*)(s_in + pix * ISTR + ks * 64 + lq * 16);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < KST; ++ks) acc = mfma16(cur.wf[mt][ks], xf[ks], acc);
                    accs[u][mt] = acc;
                }
            }
            if (MBCONV_MODE_TOGGLE) AVX_F16_SAT_BEGIN();
#pragma unroll
            for (int u = 0; u < NG; ++u) {
                const int pix = (wave + 4 * (j0 + u)) * 16 + lr;
                const bool inside = (inside_bits >> (j0 + u)) & 1u;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const f32x4 v = silu4(accs[u][mt] + cur.be[mt]);
                    ovf_see4<T>(ovf_mx, v);
                    v4 h;
#pragma unroll
                    for (int e = 0; e < 4; ++e) h[e] = Half<T>::from_hw(v[e]);
                    a_i32x2m hb = __builtin_bit_cast(a_i32x2m, h);
                    hb[0] = inside ? hb[0] : 0; hb[1] = inside ? hb[1] : 0;
                    *(a_i32x2m*)(s_exp + pix * ESTR + (16 * mt + 4 * lq) * 2) = hb;
                }
            }
            if (MBCONV_MODE_TOGGLE) AVX_F16_SAT_END();
        };
        // every wave has NPG / 4 groups, the first NPG % 4 waves one more
#pragma unroll
        for (int j = 0; j + 1 < NPG / 4; j += 2) expand_groups(j, a_icm<2>{});
        if ((NPG / 4) % 2) expand_groups(NPG / 4 - 1, a_icm<1>{});
        if (NPG % 4 != 0 && wave < NPG % 4) expand_groups(NPG / 4, a_icm<1>{});
        const f32x4 bd0 = cur.bd0, bd1 = cur.bd1;
        if (more) {                                      // every load of this chunk is consumed HERE, before the taps' stores
            stage_wd(nxt, slot ^ 1);
            cur = nxt;
        }
        __syncthreads();
        if (MBCONV_MODE_TOGGLE) AVX_F16_SAT_BEGIN();      // the depthwise stage's output conversions; cleared at the end of the chunk
        float psum[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) psum[e] = 0.f;
#pragma unroll
        for (int it = tid; it < NITEM; it += 256) {
            const int t = it / NCG;
            const int oyl = t / SXN, sx = t - oyl * SXN;
            float acc[PIX][8];
#pragma unroll
            for (int q = 0; q < PIX; ++q)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[q][e] = 0.f;
            // one input row per trip of a ROLLED loop: unrolled, hipcc issues every LDS read of the item before its first tap (220 - 512
            // registers, spills at 5 x 5; scheduling barriers do not pin the taps, which are pure arithmetic)
            const char* rowp = s_exp + ((oyl * ST) * IW + sx * PIX * ST) * ESTR + cg * 16;
            const float* wrow = s_w + slot * NWD + cg * 8;
#pragma unroll 1
            for (int r = 0; r < KS; ++r) {
                v8 col[NCOL];
#pragma unroll
                for (int q = 0; q < NCOL; ++q) col[q] = *(const v8*)(rowp + q * ESTR);
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) {
                    const f32x4 w0 = *(const f32x4*)(wrow + kx * CC), w1 = *(const f32x4*)(wrow + kx * CC + 4);
#pragma unroll
                    for (int q = 0; q < PIX; ++q) {
                        const v8 x = col[q * ST + kx];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            acc[q][e] = __builtin_fmaf((float)x[e], w0[e], acc[q][e]);
                            acc[q][4 + e] = __builtin_fmaf((float)x[4 + e], w1[e], acc[q][4 + e]);
                        }
                    }
                }
                rowp += IW * ESTR;
                wrow += KS * CC;
            }
            const int oy = oy0 + oyl;
#pragma unroll
            for (int q = 0; q < PIX; ++q) {
                const int ox = ox0 + sx * PIX + q;
                if (oy < p.Ho && ox < p.Wo) {
                    const f32x4 y0 = silu4((f32x4){acc[q][0], acc[q][1], acc[q][2], acc[q][3]} + bd0);
                    const f32x4 y1 = silu4((f32x4){acc[q][4], acc[q][5], acc[q][6], acc[q][7]} + bd1);
                    v8 h;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { h[e] = Half<T>::from_hw(y0[e]); h[4 + e] = Half<T>::from_hw(y1[e]); }
#pragma unroll
                    for (int e = 0; e < 8; ++e) psum[e] += (float)h[e];
                    *(v8*)(out + ((int64_t)oy * p.Wo + ox) * cp + c + cg * 8) = h;
                }
            }
        }
        if (p.part) {
            // lanes with the same channel group are 4 apart: two DPP row shifts leave each 16-lane row's sums in its lanes 12 .. 15; the
            // sixteen rows of a workgroup are added in a fixed order below (no shuffles through the LDS crossbar: 32 ds_bpermute per chunk)
            row_quarter_sums(psum);
            if ((lane & 15) >= 12) {
                float* dst = s_red + (wave * 4 + (lane >> 4)) * CC + ((lane & 15) - 12) * 8;
                *(f32x4*)dst = (f32x4){psum[0], psum[1], psum[2], psum[3]};
                *(f32x4*)(dst + 4) = (f32x4){psum[4], psum[5], psum[6], psum[7]};
            }
        }
        __syncthreads();
        if (p.part && tid < CC) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) s += s_red[r * CC + tid];
            p.part[((int64_t)b * gridDim.x + blockIdx.x) * cp + c + tid] = s;
        }
        if (MBCONV_MODE_TOGGLE) AVX_F16_SAT_END();      // the next chunk's MFMAs run with MODE.FP16_OVFL clear
    }
    ovf_commit<T>(p.ovf, ovf_mx);
}

}  // namespace

extern "C" int avexhip_effnet_stem(const float* img_dev, int B, int H, int W, const float* w_dev, const float* bias_dev, int Cp,
                                   void* out_dev, float* raw_dev, int dtype, void* stream) {
    AVX_REQUIRE(img_dev && w_dev && bias_dev && out_dev, "effnet_stem: null argument");
    AVX_REQUIRE(B > 0 && H > 0 && W > 0 && Cp > 0 && Cp % 8 == 0, "effnet_stem: bad shape");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const int64_t n = (int64_t)Ho * ((Wo + STEM_PIX - 1) / STEM_PIX) * (Cp / 8);
    const dim3 grid((unsigned)((n + 255) / 256), B);
    if (dtype == AVEXHIP_BF16) hipLaunchKernelGGL(stem_conv_kernel<__bf16>, grid, dim3(256), 0, (hipStream_t)stream, img_dev, H, W, Ho, Wo, w_dev, bias_dev, Cp, (__bf16*)out_dev, raw_dev);
    else hipLaunchKernelGGL(stem_conv_kernel<_Float16>, grid, dim3(256), 0, (hipStream_t)stream, img_dev, H, W, Ho, Wo, w_dev, bias_dev, Cp, (_Float16*)out_dev, raw_dev);
    AVX_LAUNCH_CHECK();
    return AVEXHIP_OK;
}

static int64_t dw_blocks(int H, int W, int Cp, int k, int st) {
    const int pad = (k - 1) / 2;
    const int Ho = (H + 2 * pad - k) / st + 1, Wo = (W + 2 * pad - k) / st + 1;
    const int64_t n = (int64_t)((Ho + DW_PY - 1) / DW_PY) * ((Wo + DW_PIX - 1) / DW_PIX) * (Cp / 8);
    return (n + 255) / 256;
}

template <typename T>
static int dw_launch(const void* in, int B, int H, int W, int Cp, int k, int st, const float* w, const float* bias, void* out, float* pool,
                     float* part, hipStream_t s) {
    const int pad = (k - 1) / 2;
    const int Ho = (H + 2 * pad - k) / st + 1, Wo = (W + 2 * pad - k) / st + 1;
    const dim3 grid((unsigned)dw_blocks(H, W, Cp, k, st), B);
#define AVX_DW(KS, ST) hipLaunchKernelGGL((dwconv_kernel<T, KS, ST>), grid, dim3(256), 0, s, (const T*)in, H, W, Ho, Wo, Cp, w, bias, (T*)out, part)
    if (k == 3 && st == 1) AVX_DW(3, 1);
    else if (k == 3 && st == 2) AVX_DW(3, 2);
    else if (k == 5 && st == 1) AVX_DW(5, 1);
    else if (k == 5 && st == 2) AVX_DW(5, 2);
    else { avexhip_set_error("effnet_dwconv: kernel %d stride %d not built (3 or 5, stride 1 or 2)", k, st); return AVEXHIP_ERR_INVALID; }
#undef AVX_DW
    if (pool) hipLaunchKernelGGL(pool_sum_kernel, dim3((Cp + 255) / 256, B), dim3(256), 0, s, part, (int)grid.x, Cp, pool);
    AVX_LAUNCH_CHECK();
    return AVEXHIP_OK;
}

extern "C" size_t avexhip_effnet_dwconv_part_bytes(int B, int H, int W, int Cp, int k, int stride) {
    if (B <= 0 || H <= 0 || W <= 0 || Cp <= 0 || Cp % 8 || (k != 3 && k != 5) || (stride != 1 && stride != 2)) return 0;
    return sizeof(float) * (size_t)B * (size_t)dw_blocks(H, W, Cp, k, stride) * Cp;
}

extern "C" int avexhip_effnet_dwconv(const void* in_dev, int B, int H, int W, int Cp, int k, int stride, const float* w_dev,
                                     const float* bias_dev, void* out_dev, float* pool_dev, float* part_dev, size_t part_bytes, int dtype,
                                     void* stream) {
    AVX_REQUIRE(in_dev && w_dev && bias_dev && out_dev, "effnet_dwconv: null argument");
    AVX_REQUIRE(B > 0 && H > 0 && W > 0 && Cp > 0 && Cp % 8 == 0, "effnet_dwconv: bad shape");
    if (pool_dev) {
        AVX_REQUIRE(Cp <= 2048, "effnet_dwconv: the pooled form wants at most 2048 padded channels, got %d", Cp);
        AVX_REQUIRE(part_dev && part_bytes >= avexhip_effnet_dwconv_part_bytes(B, H, W, Cp, k, stride),
                    "effnet_dwconv: the pooled form wants avexhip_effnet_dwconv_part_bytes() = %zu bytes of scratch, got %zu",
                    avexhip_effnet_dwconv_part_bytes(B, H, W, Cp, k, stride), part_bytes);
    }
    if (!pool_dev) part_dev = nullptr;
    if (dtype == AVEXHIP_BF16) return dw_launch<__bf16>(in_dev, B, H, W, Cp, k, stride, w_dev, bias_dev, out_dev, pool_dev, part_dev, (hipStream_t)stream);
    return dw_launch<_Float16>(in_dev, B, H, W, Cp, k, stride, w_dev, bias_dev, out_dev, pool_dev, part_dev, (hipStream_t)stream);
}

extern "C" int avexhip_effnet_se(const float* pool_dev, int B, int64_t hw, int C, int Cp, int Cs, const float* w1_dev, const float* b1_dev,
                                 const float* w2_dev, const float* b2_dev, float* scale_dev, void* x_dev, int dtype, void* stream) {
    AVX_REQUIRE(pool_dev && w1_dev && b1_dev && w2_dev && b2_dev && scale_dev, "effnet_se: null argument");      // x_dev NULL: the scale vector only
    AVX_REQUIRE(B > 0 && hw > 0 && C > 0 && C <= 2048 && Cp >= C && Cp % 8 == 0 && Cs > 0 && Cs <= 512, "effnet_se: bad shape C=%d Cp=%d Cs=%d", C, Cp, Cs);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(se_fc_kernel, dim3(B), dim3(1024), 0, s, pool_dev, 1.0f / (float)hw, C, Cp, Cs, w1_dev, b1_dev, w2_dev, b2_dev, scale_dev);
    const dim3 grid(512, B);
    if (!x_dev) {}      // the caller applies the scale itself (GemmArgs::a_scale)
    else if (dtype == AVEXHIP_BF16) hipLaunchKernelGGL(scale_channels_kernel<__bf16>, grid, dim3(256), 0, s, (__bf16*)x_dev, hw, Cp, scale_dev);
    else hipLaunchKernelGGL(scale_channels_kernel<_Float16>, grid, dim3(256), 0, s, (_Float16*)x_dev, hw, Cp, scale_dev);
    AVX_LAUNCH_CHECK();
    return AVEXHIP_OK;
}

// fp32 rows [B * HW, ld] (NHWC, the layout every kernel of the stack works in) -> [B, C, HW] (the reference's NCHW), optionally undoing a
// folded BatchNorm on the way: out = (in - shift[c]) / scale[c].  A hook tap of the reference is the convolution's output BEFORE its
// BatchNorm (efficientnet.py:82-114: model.features.0.0, *.block.3.0, model.features.8.0), while the GEMM / stem epilogues hold
// conv * scale + shift.  32 x 32 tiles through LDS: coalesced on both sides.
namespace {
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ in, int64_t ld, int HW, int C, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, float* __restrict__ out) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int p = p0 + r, c = c0 + tx;
        float v = 0.f;
        if (p < HW && c < C) {
            v = in[((int64_t)b * HW + p) * ld + c];
            if (scale) v = (v - shift[c]) / scale[c];
        }
        tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, p = p0 + tx;
        if (c < C && p < HW) out[((int64_t)b * C + c) * HW + p] = tile[tx][r];
    }
}
}  // namespace

namespace {
// ---------------------------------------------------------------------------------------------
// Depthwise convolution through LDS (the blocks the fused kernel above does not take: no expansion, or more than 64 input channels).
// dwconv_kernel reads every input element KS times from L1 / L2 (once per input row of the outputs that want it) and is bound by those
// loads; here a workgroup owns a TH x TW tile of output pixels, walks the channels in chunks of 32, and each chunk's input tile (halo
// included) crosses L2 ONCE: requested into registers before the previous chunk's taps, written to the other LDS buffer after them --
// before that chunk's output stores, so the wait for the loads never has to drain stores (see mbconv_kernel) -- and one barrier per
// chunk.  Small feature maps (8 x 63, 4 x 32) have one or two tiles per clip: gridDim.z splits the chunks over several workgroups.
// Taps in (ky, kx) order, SiLU, rounding and the squeeze partials (one row per TILE) as in dwconv_kernel: the same output bits.
// (Computing the first block's input tile in here -- the stem convolution on the mel patch, so that the stem's output never reaches HBM --
// was built and measured: bit-identical, 693 us against 225 + 326 us for the two kernels; the halo's stem work outweighs the bytes.  Not kept.)
// ---------------------------------------------------------------------------------------------
struct DwArgs {
    const void* in; int H, W, Cp;
    const float* w_dw; const float* b_dw;
    void* out; int Ho, Wo;
    float* part;                 // [B][tiles][Cp] or NULL
    int tiles_x, chunks_per_wg;
};

// NSLOT 1 (round 5): a workgroup that owns ONE 32-channel chunk (the first block: 32 channels in all) has nothing to request ahead and
// no use for the second tile buffer (nor for the registers that hold a requested chunk: 114 instead of 190); with nothing pipelined
// inside a workgroup the other workgroups of the CU are all that covers its loads, and without the second buffer four fit instead of
// two: the first block's depthwise kernel 333 -> 259 us (three waves per SIMD) -> 230 us (four; five spill) per 256 clips, the same bits.
// (Workgroups that walk SEVERAL chunks keep the two-buffer form: the one-buffer loop was measured on them too, 8 - 9 % slower --
// 91 / 90 / 130 / 188 / 190 against 83 / 82 / 115 / 175 / 177 us for the five late depthwise launches.)
template <typename T, int KS, int ST, int TH, int TW, int PIX, int NSLOT = 2>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NSLOT == 1 ? (KS == 3 ? 4 : 3) : 2, NSLOT == 1 ? (KS == 3 ? 4 : 3) : 2))) void dwconv_lds_kernel(const DwArgs p) {
    AVX_F16_SATURATE_ON();
    typedef typename Half<T>::v8 v8;
    typedef MbGeo<KS, ST, 0, TH, TW, 32> G;
    constexpr int CC = 32, PAD = (KS - 1) / 2, IW = G::IW, NPIX = G::NPIX, NPX = G::NPX, ESTR = G::ESTR, NWD = G::NWD;
    constexpr int NCOL = (PIX - 1) * ST + KS, SXN = TW / PIX, NCG = 4, NITEM = TH * SXN * NCG;
    constexpr int NLD = (NPX * NCG + 255) / 256, NWR = (NWD + 255) / 256;
    static_assert(TW % PIX == 0 && NITEM == 256, "dwconv_lds tile: one item per thread");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_x = smem;                                    // [NSLOT][NPX * ESTR]
    float* s_w = (float*)(smem + NSLOT * NPX * ESTR);    // [NSLOT][NWD]
    float* s_red = s_w + NSLOT * NWD;                    // [NSLOT][16 * CC]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y;
    const int ty = blockIdx.x / p.tiles_x, tx = blockIdx.x - ty * p.tiles_x;
    const int oy0 = ty * TH, ox0 = tx * TW, iy0 = oy0 * ST - PAD, ix0 = ox0 * ST - PAD;
    const int cp = p.Cp;
    const T* in = (const T*)p.in + (int64_t)b * p.H * p.W * cp;
    T* out = (T*)p.out + (int64_t)b * p.Ho * p.Wo * cp;
    const int cg = tid & 3;
    const int c_begin = blockIdx.z * p.chunks_per_wg * CC;
    const int c_end = c_begin + p.chunks_per_wg * CC < cp ? c_begin + p.chunks_per_wg * CC : cp;

    // this thread's share of a chunk's input tile: element offsets (channel 0 of the chunk), -1 outside the image / tile
    int off[NLD];
#pragma unroll
    for (int r = 0; r < NLD; ++r) {
        const int i = tid + 256 * r;
        const int pix = i >> 2, ch = i & 3;
        const int iy = pix / IW, ix = pix - iy * IW;
        const int gy = iy0 + iy, gx = ix0 + ix;
        off[r] = (i < NPX * NCG && pix < NPIX && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) ? (gy * p.W + gx) * cp + ch * 8 : -1;
    }
    struct Par { uint4 x[NLD]; float wd[NWR]; f32x4 bd0, bd1; };
    auto request = [&](Par& q, int c) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < NLD; ++r) {
            q.x[r] = (uint4){0u, 0u, 0u, 0u};
            if (off[r] >= 0) q.x[r] = *(const uint4*)(in + off[r] + c);
        }
#pragma unroll
        for (int r = 0; r < NWR; ++r) {
            const int i = tid + 256 * r;
            q.wd[r] = i < NWD ? p.w_dw[(int64_t)(i / CC) * cp + c + (i % CC)] : 0.f;
        }
        q.bd0 = *(const f32x4*)(p.b_dw + c + cg * 8);
        q.bd1 = *(const f32x4*)(p.b_dw + c + cg * 8 + 4);
    };
    auto stage = [&](const Par& q, int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < NLD; ++r) {
            const int i = tid + 256 * r;
            if (i < NPX * NCG) *(uint4*)(s_x + slot * (NPX * ESTR) + (i >> 2) * ESTR + (i & 3) * 16) = q.x[r];
        }
#pragma unroll
        for (int r = 0; r < NWR; ++r) {
            const int i = tid + 256 * r;
            if (i < NWD) s_w[slot * NWD + i] = q.wd[r];
        }
    };
    Par nx;
    request(nx, c_begin);
    stage(nx, 0);
    f32x4 bd0 = nx.bd0, bd1 = nx.bd1;
    __syncthreads();
    const int t = tid >> 2;
    const int oyl = t / SXN, sx = t - oyl * SXN;
    const int oy = oy0 + oyl;
    int slot = 0;
    for (int c = c_begin; c < c_end; c += CC, slot ^= 1) {
        const bool more = NSLOT > 1 && c + CC < c_end;
        if (more) request(nx, c + CC);
        float acc[PIX][8];
#pragma unroll
        for (int q = 0; q < PIX; ++q)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[q][e] = 0.f;
        const char* rowp = s_x + slot * (NPX * ESTR) + ((oyl * ST) * IW + sx * PIX * ST) * ESTR + cg * 16;
        const float* wrow = s_w + slot * NWD + cg * 8;
#pragma unroll 1
        for (int r = 0; r < KS; ++r) {
            v8 col[NCOL];
#pragma unroll
            for (int q = 0; q < NCOL; ++q) col[q] = *(const v8*)(rowp + q * ESTR);
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
                const f32x4 w0 = *(const f32x4*)(wrow + kx * CC), w1 = *(const f32x4*)(wrow + kx * CC + 4);
#pragma unroll
                for (int q = 0; q < PIX; ++q) {
                    const v8 x = col[q * ST + kx];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[q][e] = __builtin_fmaf((float)x[e], w0[e], acc[q][e]);
                        acc[q][4 + e] = __builtin_fmaf((float)x[4 + e], w1[e], acc[q][4 + e]);
                    }
                }
            }
            rowp += IW * ESTR;
            wrow += KS * CC;
        }
        const f32x4 cb0 = bd0, cb1 = bd1;
        if (more) {                                      // the next chunk's tile: consumed before this chunk's stores
            stage(nx, slot ^ 1);
            bd0 = nx.bd0; bd1 = nx.bd1;
        }
        float psum[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) psum[e] = 0.f;
#pragma unroll
        for (int q = 0; q < PIX; ++q) {
            const int ox = ox0 + sx * PIX + q;
            if (oy < p.Ho && ox < p.Wo) {
                const f32x4 y0 = silu4((f32x4){acc[q][0], acc[q][1], acc[q][2], acc[q][3]} + cb0);
                const f32x4 y1 = silu4((f32x4){acc[q][4], acc[q][5], acc[q][6], acc[q][7]} + cb1);
                v8 h;
#pragma unroll
                for (int e = 0; e < 4; ++e) { h[e] = Half<T>::from_hw(y0[e]); h[4 + e] = Half<T>::from_hw(y1[e]); }
#pragma unroll
                for (int e = 0; e < 8; ++e) psum[e] += (float)h[e];
                *(v8*)(out + ((int64_t)oy * p.Wo + ox) * cp + c + cg * 8) = h;
            }
        }
        float* red = s_red + slot * (16 * CC);
        if (p.part) {
            row_quarter_sums(psum);
            if ((lane & 15) >= 12) {
                float* dst = red + (wave * 4 + (lane >> 4)) * CC + ((lane & 15) - 12) * 8;
                *(f32x4*)dst = (f32x4){psum[0], psum[1], psum[2], psum[3]};
                *(f32x4*)(dst + 4) = (f32x4){psum[4], psum[5], psum[6], psum[7]};
            }
        }
        __syncthreads();
        if (p.part && tid < CC) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) s += red[r * CC + tid];
            p.part[((int64_t)b * gridDim.x + blockIdx.x) * cp + c + tid] = s;
        }
    }
}

template <typename T, int KS, int ST>
int dwl_launch(const DwArgs& a0, int B, int64_t* n_tiles, hipStream_t s) {
    constexpr int TH = ST == 1 ? 8 : 4, TW = ST == 1 ? 32 : 16, PIX = ST == 1 ? 4 : 1;
    typedef MbGeo<KS, ST, 0, TH, TW, 32> G;
    constexpr int LDS = 2 * G::NPX * G::ESTR + 2 * G::NWD * 4 + 2 * 16 * 32 * 4;
    constexpr int LDS1 = G::NPX * G::ESTR + G::NWD * 4 + 16 * 32 * 4;
    DwArgs a = a0;
    a.tiles_x = (a.Wo + TW - 1) / TW;
    const int64_t tiles = (int64_t)a.tiles_x * ((a.Ho + TH - 1) / TH);
    if (n_tiles) { *n_tiles = tiles; return AVEXHIP_OK; }
    // enough workgroups for the chip (two per CU fit): split the channel chunks when the map has few tiles
    const int nchunk = a.Cp / 32;
    int64_t split = (1024 + tiles * B - 1) / (tiles * B);
    split = split < 1 ? 1 : (split > nchunk ? nchunk : split);
    a.chunks_per_wg = (int)((nchunk + split - 1) / split);
    const int gz = (nchunk + a.chunks_per_wg - 1) / a.chunks_per_wg;
    if (a.chunks_per_wg == 1 && ST == 1) {
        if constexpr (ST == 1) {
            AVX_ENSURE_LDS((dwconv_lds_kernel<T, KS, ST, TH, TW, PIX, 1>), LDS1);
            hipLaunchKernelGGL((dwconv_lds_kernel<T, KS, ST, TH, TW, PIX, 1>), dim3((unsigned)tiles, B, gz), dim3(256), LDS1, s, a);
        }
    } else {
        AVX_ENSURE_LDS((dwconv_lds_kernel<T, KS, ST, TH, TW, PIX>), LDS);
        hipLaunchKernelGGL((dwconv_lds_kernel<T, KS, ST, TH, TW, PIX>), dim3((unsigned)tiles, B, gz), dim3(256), LDS, s, a);
    }
    AVX_LAUNCH_CHECK();
    return AVEXHIP_OK;
}
template <typename T>
int dwl_dispatch(const DwArgs& a, int B, int k, int st, int64_t* n_tiles, hipStream_t s) {
    if (k == 3 && st == 1) return dwl_launch<T, 3, 1>(a, B, n_tiles, s);
    if (k == 3 && st == 2) return dwl_launch<T, 3, 2>(a, B, n_tiles, s);
    if (k == 5 && st == 1) return dwl_launch<T, 5, 1>(a, B, n_tiles, s);
    if (k == 5 && st == 2) return dwl_launch<T, 5, 2>(a, B, n_tiles, s);
    avexhip_set_error("dwconv_lds: kernel %d stride %d not built (3 or 5, stride 1 or 2)", k, st);
    return AVEXHIP_ERR_INVALID;
}
}  // namespace

namespace {
// tile of the fused kernel for (stride, KIN): output rows x columns per workgroup and pixels along x per thread
template <int ST, int KIN> struct MbTile;
template <int KIN> struct MbTile<2, KIN> { static constexpr int TH = 4, TW = 16, PIX = 1; };
template <> struct MbTile<1, 32> { static constexpr int TH = 8, TW = 32, PIX = 4; };
template <> struct MbTile<1, 64> { static constexpr int TH = 8, TW = 16, PIX = 2; };      // 144-byte input pixels: the narrower tile keeps two workgroups per CU

template <typename T, int KS, int ST, int KIN>
int mb_launch(const MbArgs& a0, int B, int64_t* n_tiles, hipStream_t s) {
    typedef MbTile<ST, KIN> Tl;
    typedef MbGeo<KS, ST, KIN, Tl::TH, Tl::TW, 32> G;
    MbArgs a = a0;
    a.tiles_x = (a.Wo + Tl::TW - 1) / Tl::TW;
    const int64_t tiles = (int64_t)a.tiles_x * ((a.Ho + Tl::TH - 1) / Tl::TH);
    if (n_tiles) { *n_tiles = tiles; return AVEXHIP_OK; }
    AVX_ENSURE_LDS((mbconv_kernel<T, KS, ST, KIN, Tl::TH, Tl::TW, Tl::PIX, 32>), G::LDS);
    hipLaunchKernelGGL((mbconv_kernel<T, KS, ST, KIN, Tl::TH, Tl::TW, Tl::PIX, 32>), dim3((unsigned)tiles, B), dim3(256), G::LDS, s, a);
    AVX_LAUNCH_CHECK();
    return AVEXHIP_OK;
}
template <typename T>
int mb_dispatch(const MbArgs& a, int B, int k, int st, int kin, int64_t* n_tiles, hipStream_t s) {
#define AVX_MB(KS, ST, KIN) if (k == KS && st == ST && kin == KIN) return mb_launch<T, KS, ST, KIN>(a, B, n_tiles, s)
    AVX_MB(3, 1, 32); AVX_MB(3, 2, 32); AVX_MB(5, 1, 32); AVX_MB(5, 2, 32);
    AVX_MB(3, 1, 64); AVX_MB(3, 2, 64); AVX_MB(5, 1, 64); AVX_MB(5, 2, 64);
#undef AVX_MB
    avexhip_set_error("mbconv_front: kernel %d stride %d K %d not built (k 3 | 5, stride 1 | 2, K 32 | 64)", k, st, kin);
    return AVEXHIP_ERR_INVALID;
}
}  // namespace

namespace avx {
// depthwise convolution that leaves its rows of squeeze partials in `part` WITHOUT reducing them (se_from_parts does); returns the rows per clip
int dwconv_parts(const void* in, int B, int H, int W, int Cp, int k, int stride, const float* w, const float* bias, void* out, float* part, size_t part_bytes,
                 int64_t* rows, int dtype, hipStream_t s) {
    AVX_REQUIRE(in && w && bias && out && part && rows && B > 0 && H > 0 && W > 0 && Cp > 0 && Cp % 8 == 0 && Cp <= 2048, "dwconv_parts: bad arguments");
    AVX_REQUIRE(part_bytes >= avexhip_effnet_dwconv_part_bytes(B, H, W, Cp, k, stride), "dwconv_parts: %zu bytes of scratch wanted, %zu given",
                avexhip_effnet_dwconv_part_bytes(B, H, W, Cp, k, stride), part_bytes);
    *rows = dw_blocks(H, W, Cp, k, stride);
    if (dtype == AVEXHIP_BF16) return dw_launch<__bf16>(in, B, H, W, Cp, k, stride, w, bias, out, nullptr, part, s);
    return dw_launch<_Float16>(in, B, H, W, Cp, k, stride, w, bias, out, nullptr, part, s);
}
// the LDS form of the depthwise convolution (Cp % 32 == 0); rows of partials per clip = its tiles
int64_t dwconv_lds_tiles(int H, int W, int k, int stride) {
    DwArgs a;
    memset(&a, 0, sizeof(a));
    const int pad = (k - 1) / 2;
    a.Ho = (H + 2 * pad - k) / stride + 1; a.Wo = (W + 2 * pad - k) / stride + 1;
    int64_t n = 0;
    if (dwl_dispatch<_Float16>(a, 1, k, stride, &n, nullptr) != AVEXHIP_OK) return 0;
    return n;
}
int dwconv_lds_parts(const void* in, int B, int H, int W, int Cp, int k, int stride, const float* w, const float* bias, void* out, float* part,
                     size_t part_bytes, int64_t* rows, int dtype, hipStream_t s) {
    AVX_REQUIRE(in && w && bias && out && rows && B > 0 && B <= 65535 && H > 0 && W > 0 && Cp > 0 && Cp % 32 == 0 && Cp <= 2048, "dwconv_lds_parts: bad arguments");
    AVX_REQUIRE((int64_t)H * W * Cp < (1ll << 31), "dwconv_lds_parts: a clip's feature map of %d x %d x %d elements does not fit 32-bit offsets", H, W, Cp);
    const int64_t tiles = dwconv_lds_tiles(H, W, k, stride);
    AVX_REQUIRE(tiles > 0, "dwconv_lds_parts: kernel %d stride %d not built", k, stride);
    AVX_REQUIRE(!part || part_bytes >= sizeof(float) * (size_t)B * tiles * Cp, "dwconv_lds_parts: %zu bytes of scratch wanted, %zu given",
                sizeof(float) * (size_t)B * tiles * Cp, part_bytes);
    *rows = tiles;
    DwArgs a;
    memset(&a, 0, sizeof(a));
    const int pad = (k - 1) / 2;
    a.in = in; a.H = H; a.W = W; a.Cp = Cp; a.w_dw = w; a.b_dw = bias; a.out = out; a.part = part;
    a.Ho = (H + 2 * pad - k) / stride + 1; a.Wo = (W + 2 * pad - k) / stride + 1;
    return dtype == AVEXHIP_BF16 ? dwl_dispatch<__bf16>(a, B, k, stride, nullptr, s) : dwl_dispatch<_Float16>(a, B, k, stride, nullptr, s);
}
// squeeze-excitation scale [B][Cp] from `rows` rows of partial sums per clip; w2t is the second layer TRANSPOSED [Cs][C]; x (or NULL) is rescaled in place
int se_from_parts(const float* part, int64_t rows, int B, int64_t hw, int C, int Cp, int Cs, const float* w1, const float* b1, const float* w2t,
                  const float* b2, float* scale, void* x, int dtype, hipStream_t s) {
    AVX_REQUIRE(part && w1 && b1 && w2t && b2 && scale && rows > 0 && B > 0 && hw > 0 && C > 0 && C <= 2048 && Cp >= C && Cp % 8 == 0 && Cs > 0 && Cs <= 512,
                "se_from_parts: bad shape C=%d Cp=%d Cs=%d", C, Cp, Cs);
    hipLaunchKernelGGL(se_pool_fc_kernel, dim3(B), dim3(1024), 0, s, part, (int)rows, 1.0f / (float)hw, C, Cp, Cs, w1, b1, w2t, b2, scale);
    const dim3 grid(512, B);
    if (!x) {}
    else if (dtype == AVEXHIP_BF16) hipLaunchKernelGGL(scale_channels_kernel<__bf16>, grid, dim3(256), 0, s, (__bf16*)x, hw, Cp, scale);
    else hipLaunchKernelGGL(scale_channels_kernel<_Float16>, grid, dim3(256), 0, s, (_Float16*)x, hw, Cp, scale);
    AVX_LAUNCH_CHECK();
    return AVEXHIP_OK;
}
// workgroups per clip of the fused kernel = rows of squeeze partials it leaves per clip
int64_t mbconv_front_tiles(int H, int W, int k, int stride, int kin) {
    MbArgs a;
    memset(&a, 0, sizeof(a));
    const int pad = (k - 1) / 2;
    a.Ho = (H + 2 * pad - k) / stride + 1; a.Wo = (W + 2 * pad - k) / stride + 1;
    int64_t n = 0;
    if (mb_dispatch<_Float16>(a, 1, k, stride, kin, &n, nullptr) != AVEXHIP_OK) return 0;
    return n;
}
// expansion (kin = 32 | 64 input channels read from rows of ld_in) + depthwise k x k
// + SiLU + squeeze partials; in [B, H, W, ld_in], out [B, Ho, Wo, cp_exp] (cp_exp % 32 == 0), pool [B, cp_exp] or NULL with
// part >= B * mbconv_front_tiles() * cp_exp floats
int mbconv_front(const void* in, int B, int H, int W, int ld_in, int kin, const void* w_exp, int ldw, const float* b_exp, int k, int stride,
                 const float* w_dw, const float* b_dw, int cp_exp, void* out, float* pool, float* part, size_t part_bytes, unsigned int* ovf,
                 int dtype, hipStream_t s) {
    AVX_REQUIRE(in && w_dw && b_dw && out && B > 0 && H > 0 && W > 0 && cp_exp > 0 && cp_exp % 32 == 0 && B <= 65535, "mbconv_front: bad arguments");
    AVX_REQUIRE(w_exp && b_exp && ld_in >= kin && ldw >= kin, "mbconv_front: expansion weights missing or rows shorter than K = %d", kin);
    MbArgs a;
    memset(&a, 0, sizeof(a));
    const int pad = (k - 1) / 2;
    a.in = in; a.H = H; a.W = W; a.ld_in = ld_in; a.w_exp = w_exp; a.ldw = ldw; a.b_exp = b_exp; a.w_dw = w_dw; a.b_dw = b_dw;
    a.out = out; a.Ho = (H + 2 * pad - k) / stride + 1; a.Wo = (W + 2 * pad - k) / stride + 1; a.cp_exp = cp_exp; a.ovf = ovf;
    const int64_t tiles = mbconv_front_tiles(H, W, k, stride, kin);
    AVX_REQUIRE(tiles > 0, "mbconv_front: kernel %d stride %d K %d not built", k, stride, kin);
    AVX_REQUIRE(!pool || part, "mbconv_front: the pooled form wants the scratch for its partial sums");
    if (part) {
        AVX_REQUIRE(part_bytes >= sizeof(float) * (size_t)B * tiles * cp_exp, "mbconv_front: %zu bytes of squeeze scratch wanted, %zu given",
                    sizeof(float) * (size_t)B * tiles * cp_exp, part_bytes);
        a.part = part;
    }
    const int rc = dtype == AVEXHIP_BF16 ? mb_dispatch<__bf16>(a, B, k, stride, kin, nullptr, s) : mb_dispatch<_Float16>(a, B, k, stride, kin, nullptr, s);
    if (rc != AVEXHIP_OK) return rc;
    if (pool) {
        hipLaunchKernelGGL(pool_sum_kernel, dim3((cp_exp + 255) / 256, B), dim3(256), 0, s, part, (int)tiles, cp_exp, pool);
        AVX_LAUNCH_CHECK();
    }
    return AVEXHIP_OK;
}
}  // namespace avx

namespace avx {
int nhwc_to_nchw(const float* in, int64_t ld, int B, int HW, int C, const float* scale, const float* shift, float* out, hipStream_t s) {
    AVX_REQUIRE(in && out && B > 0 && HW > 0 && C > 0 && ld >= C && B <= 65535, "nhwc_to_nchw: bad arguments");
    AVX_REQUIRE((scale == nullptr) == (shift == nullptr), "nhwc_to_nchw: scale and shift come together");
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((HW + 31) / 32, (C + 31) / 32, B), dim3(256), 0, s, in, ld, HW, C, scale, shift, out);
    AVX_LAUNCH_CHECK();
    return AVEXHIP_OK;
}
}  // namespace avx
