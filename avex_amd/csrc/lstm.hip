// LSTM layer for the LSTM probe (reference avex/models/probes/lstm_probe.py -> torch.nn.LSTM).
// Its own translation unit: the cell's sigmoid / tanh pairs make hipcc's SLP vectoriser emit v_pk_add_f32 with op_sel:[0,1], the packed form
// that is wrong beside matrix work on gfx950 (avex_amd/isa_lint.py); this file is built with -fno-slp-vectorize (avex_amd/build.py).
#include <hip/hip_runtime.h>
#include <math.h>

#include "common.h"

// One direction of one nn.LSTM layer (lstm_probe.py:61-68), the whole recurrence in one launch.  xg [B, T, 4H] holds the input half of the
// gates for every step (x W_ih^T + b_ih + b_hh: one dense product beforehand).  A workgroup owns LR clips and KS * H threads: thread (q, j) adds
// the quarter k in [q H / KS, (q + 1) H / KS) of h_{t-1} W_hh^T for hidden unit j's four gates (rows i, f, g, o of W_hh -- read TRANSPOSED,
// w_hhT [H][4H], so that the threads of a wave read consecutive addresses; h_{t-1} from LDS as broadcasts); the partial sums meet in LDS, and
// thread (q, j) finishes clip r = q (+ KS, ...) of the workgroup: adds the KS partials in order, applies the cell (its c stays in a register),
// writes h_t to LDS and to out[b][t][..].  fp32 throughout; W_hh (1 MB at H = 256) streams from L2 every step, which is what bounds a step:
// sixteen waves per CU keep four times the loads in flight of the one-thread-per-unit form (11 -> 6 ms per layer at 256 clips x 496 steps x 256 units).
constexpr int LR = 4;
template <int KS>
__global__ __launch_bounds__(1024) void lstm_layer_kernel(const float* __restrict__ xg, const float* __restrict__ w_hhT, int B, int T, int H, int reverse,
                                                           float* __restrict__ out, int64_t ldo) {
    extern __shared__ float lds[];                 // [LR][H] h_{t-1}, then [KS][LR][4][H] partial gate sums
    float* hs = lds;
    float* part = lds + LR * H;
    const int j = threadIdx.x % H, q = threadIdx.x / H;
    const int b0 = blockIdx.x * LR;
    const int kq = (H + KS - 1) / KS;                          // this thread's share of the contraction (the last share may be shorter)
    const int k0 = q * kq, k1 = k0 + kq < H ? k0 + kq : H;
    float c[(LR + KS - 1) / KS];
#pragma unroll
    for (int i = 0; i < (LR + KS - 1) / KS; ++i) c[i] = 0.f;
    for (int r = q; r < LR; r += KS) hs[r * H + j] = 0.f;
    __syncthreads();
    for (int s = 0; s < T; ++s) {
        const int t = reverse ? T - 1 - s : s;
        float acc[LR][4];
#pragma unroll
        for (int r = 0; r < LR; ++r) { acc[r][0] = 0.f; acc[r][1] = 0.f; acc[r][2] = 0.f; acc[r][3] = 0.f; }
        for (int k = k0; k < k1; ++k) {
            const float* wr = w_hhT + (int64_t)k * 4 * H + j;
            const float w0 = wr[0], w1 = wr[H], w2 = wr[2 * H], w3 = wr[3 * H];
#pragma unroll
            for (int r = 0; r < LR; ++r) {
                const float h = hs[r * H + k];
                acc[r][0] = __builtin_fmaf(h, w0, acc[r][0]); acc[r][1] = __builtin_fmaf(h, w1, acc[r][1]);
                acc[r][2] = __builtin_fmaf(h, w2, acc[r][2]); acc[r][3] = __builtin_fmaf(h, w3, acc[r][3]);
            }
        }
#pragma unroll
        for (int r = 0; r < LR; ++r)
#pragma unroll
            for (int g = 0; g < 4; ++g) part[((q * LR + r) * 4 + g) * H + j] = acc[r][g];
        __syncthreads();                           // partial sums complete; every thread has finished reading h_{t-1}
        int ci = 0;
        for (int r = q; r < LR; r += KS, ++ci) {
            const int b = b0 + r < B ? b0 + r : B - 1;
            const float* gx = xg + ((int64_t)b * T + t) * 4 * H + j;
            float gsum[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v = gx[g * H];
                for (int qq = 0; qq < KS; ++qq) v += part[((qq * LR + r) * 4 + g) * H + j];
                gsum[g] = v;
            }
            const float ig = 1.0f / (1.0f + __expf(-gsum[0])), fg = 1.0f / (1.0f + __expf(-gsum[1]));
            const float gg = tanhf(gsum[2]), og = 1.0f / (1.0f + __expf(-gsum[3]));
            c[ci] = fg * c[ci] + ig * gg;
            const float hv = og * tanhf(c[ci]);
            hs[r * H + j] = hv;
            if (b0 + r < B) out[((int64_t)(b0 + r) * T + t) * ldo + j] = hv;
        }
        __syncthreads();
    }
}

extern "C" int avexhip_lstm_layer(const float* xg, const float* w_hhT, int B, int T, int H, int reverse, float* out, int64_t ldo, void* stream) {
    AVX_REQUIRE(xg && w_hhT && out && B >= 0 && T >= 1, "lstm_layer: bad arguments");
    AVX_REQUIRE(H >= 1 && H <= 1024 && ldo >= H, "lstm_layer: hidden size %d (1 .. 1024)", H);      // any width: the reference's max(hidden, max_sequence_length / 4) gives 300 for its shipped configs
    if (B == 0) return AVEXHIP_OK;
    const dim3 grid((B + LR - 1) / LR);
    hipStream_t s = (hipStream_t)stream;
    if (H <= 256) {
        const size_t lds = sizeof(float) * ((size_t)LR * H + (size_t)4 * LR * 4 * H);
        AVX_ENSURE_LDS(lstm_layer_kernel<4>, 96 * 1024);
        lstm_layer_kernel<4><<<grid, dim3(4 * H), lds, s>>>(xg, w_hhT, B, T, H, reverse, out, ldo);
    } else if (H <= 341) {      // (the reference's shipped configs: 300 units)
        const size_t lds = sizeof(float) * ((size_t)LR * H + (size_t)3 * LR * 4 * H);
        AVX_ENSURE_LDS(lstm_layer_kernel<3>, 96 * 1024);
        lstm_layer_kernel<3><<<grid, dim3(3 * H), lds, s>>>(xg, w_hhT, B, T, H, reverse, out, ldo);
    } else if (H <= 512) {
        const size_t lds = sizeof(float) * ((size_t)LR * H + (size_t)2 * LR * 4 * H);
        AVX_ENSURE_LDS(lstm_layer_kernel<2>, 96 * 1024);
        lstm_layer_kernel<2><<<grid, dim3(2 * H), lds, s>>>(xg, w_hhT, B, T, H, reverse, out, ldo);
    } else {
        const size_t lds = sizeof(float) * ((size_t)LR * H + (size_t)1 * LR * 4 * H);
        AVX_ENSURE_LDS(lstm_layer_kernel<1>, 96 * 1024);
        lstm_layer_kernel<1><<<grid, dim3(H), lds, s>>>(xg, w_hhT, B, T, H, reverse, out, ldo);
    }
    AVX_LAUNCH_CHECK();
    return AVEXHIP_OK;
}
