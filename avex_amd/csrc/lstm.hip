// LSTM layer for the LSTM probe (reference avex/models/probes/lstm_probe.py -> torch.nn.LSTM).
// Its own translation unit: the cell's sigmoid / tanh pairs make hipcc's SLP vectoriser emit v_pk_add_f32 with op_sel:[0,1], the packed form
// that is wrong beside matrix work on gfx950 (avex_amd/isa_lint.py); this file is built with -fno-slp-vectorize (avex_amd/build.py).
#include <hip/hip_runtime.h>
#include <math.h>

#include "common.h"

// One direction of one nn.LSTM layer (lstm_probe.py:61-68), the whole recurrence in one launch.  xg [B, T, 4H] holds the input half of the
// gates for every step (x W_ih^T + b_ih + b_hh: one dense product beforehand); a workgroup owns LR clips and H threads, thread j = hidden
// unit j: per step it adds h_{t-1} W_hh^T for its four gates (rows i, f, g, o of W_hh -- read TRANSPOSED, w_hhT [H][4H], so that the
// threads of a wave read consecutive addresses; h_{t-1} from LDS as broadcasts), applies the cell, and writes h_t to LDS and to
// out[b][t][..].  fp32 throughout; W_hh (1 MB at H = 256) streams from L2 every step: ~4 us per step.
constexpr int LR = 4;
__global__ __launch_bounds__(1024) void lstm_layer_kernel(const float* __restrict__ xg, const float* __restrict__ w_hhT, int B, int T, int H, int reverse,
                                                           float* __restrict__ out, int64_t ldo) {
    extern __shared__ float hs[];                  // [LR][H] h_{t-1}
    const int j = threadIdx.x;
    const int b0 = blockIdx.x * LR;
    float c[LR], hv[LR];
#pragma unroll
    for (int r = 0; r < LR; ++r) { c[r] = 0.f; hv[r] = 0.f; hs[r * H + j] = 0.f; }
    __syncthreads();
    for (int s = 0; s < T; ++s) {
        const int t = reverse ? T - 1 - s : s;
        float acc[LR][4];
#pragma unroll
        for (int r = 0; r < LR; ++r) {
            const int b = b0 + r < B ? b0 + r : B - 1;
            const float* g = xg + ((int64_t)b * T + t) * 4 * H + j;
            acc[r][0] = g[0]; acc[r][1] = g[H]; acc[r][2] = g[2 * H]; acc[r][3] = g[3 * H];
        }
        for (int k = 0; k < H; ++k) {
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
        for (int r = 0; r < LR; ++r) {
            const float ig = 1.0f / (1.0f + __expf(-acc[r][0])), fg = 1.0f / (1.0f + __expf(-acc[r][1]));
            const float gg = tanhf(acc[r][2]), og = 1.0f / (1.0f + __expf(-acc[r][3]));
            c[r] = fg * c[r] + ig * gg;
            hv[r] = og * tanhf(c[r]);
        }
        __syncthreads();                           // every thread has finished reading h_{t-1}
#pragma unroll
        for (int r = 0; r < LR; ++r) {
            hs[r * H + j] = hv[r];
            if (b0 + r < B) out[((int64_t)(b0 + r) * T + t) * ldo + j] = hv[r];
        }
        __syncthreads();
    }
}

extern "C" int avexhip_lstm_layer(const float* xg, const float* w_hhT, int B, int T, int H, int reverse, float* out, int64_t ldo, void* stream) {
    AVX_REQUIRE(xg && w_hhT && out && B >= 0 && T >= 1, "lstm_layer: bad arguments");
    AVX_REQUIRE(H >= 64 && H <= 1024 && H % 64 == 0 && ldo >= H, "lstm_layer: hidden size %d (need a multiple of 64, <= 1024)", H);
    if (B == 0) return AVEXHIP_OK;
    lstm_layer_kernel<<<dim3((B + LR - 1) / LR), dim3(H), sizeof(float) * LR * H, (hipStream_t)stream>>>(xg, w_hhT, B, T, H, reverse, out, ldo);
    AVX_LAUNCH_CHECK();
    return AVEXHIP_OK;
}

