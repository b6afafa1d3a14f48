// Standalone harness of the variant-3 attention kernel (avex_amd/csrc/attention16.hip), for knock-out builds and in-kernel stamps
// without relinking the library:
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off [-DA3_KO=n] [-DA3_STAMPS=1] scripts/micro/att16_bench.hip -o scripts/micro/bin/att16_<tag>
//   ./att16_<tag> [B=256] [T=496] [iters=200]
// Random f16 q/k/v (unit variance), random bias row (0.3), the gate on.  Prints microseconds per launch at the board's steady state
// (the loop runs ~1 s before the timed part) and, with -DA3_STAMPS=1, the median shader-cycle stamps of one steady phase.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#include "../../avex_amd/csrc/attention16.hip"

void avexhip_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
namespace avx {
int ensure_max_dynamic_lds(const void* func, int bytes) { return hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess ? 0 : -1; }
int device_cu_count(int* n) { hipDeviceProp_t p; hipGetDeviceProperties(&p, 0); *n = p.multiProcessorCount; return 0; }
}

static unsigned long long rng = 88172645463325252ull;
static float urand() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (float)((rng >> 11) & 0xFFFFFF) / 16777216.0f; }
static float nrand() { float s = 0; for (int i = 0; i < 12; ++i) s += urand(); return s - 6.0f; }

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 256, T = argc > 2 ? atoi(argv[2]) : 496, iters = argc > 3 ? atoi(argv[3]) : 200, H = 12;
    const size_t nq = (size_t)B * T * 3 * H * 64;
    std::vector<_Float16> h(nq);
    for (size_t i = 0; i < nq; ++i) h[i] = (_Float16)nrand();
    std::vector<float> tab((size_t)H * (2 * T - 1)), gw(8 * 64), gb(8), ga(H, 1.0f);
    for (auto& v : tab) v = 0.3f * nrand();
    for (auto& v : gw) v = 0.1f * nrand();
    for (auto& v : gb) v = 0.1f * nrand();
    _Float16 *dq, *dout; float *dtab, *dgw, *dgb, *dga;
    hipMalloc(&dq, nq * 2); hipMalloc(&dout, (size_t)B * T * H * 64 * 2);
    hipMalloc(&dtab, tab.size() * 4); hipMalloc(&dgw, gw.size() * 4); hipMalloc(&dgb, 32); hipMalloc(&dga, H * 4);
    hipMemcpy(dq, h.data(), nq * 2, hipMemcpyHostToDevice);
    hipMemcpy(dtab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dgw, gw.data(), gw.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dgb, gb.data(), 32, hipMemcpyHostToDevice);
    hipMemcpy(dga, ga.data(), H * 4, hipMemcpyHostToDevice);
    int ncu = 256; avx::device_cu_count(&ncu);
    auto run = [&]() { return avx::attention16(dq, B, T, H, dtab, dgw, dgb, dga, nullptr, dout, AVEXHIP_F16, 0, ncu, 0); };
    if (run() != 0) return 1;
    hipDeviceSynchronize();
    for (int i = 0; i < 3 * iters; ++i) run();      // to the board's steady clock
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f, tot = 0;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0, 0);
        for (int i = 0; i < iters; ++i) run();
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = std::min(best, ms / iters); tot += ms / iters;
    }
    const double fl = 4.0 * B * T * (double)T * H * 64;
    printf("A3_KO=%d B=%d T=%d: mean %.1f us  min %.1f us  (%.0f TFLOP/s at the mean)\n", (int)A3_KO, B, T, tot / 5 * 1e3, best * 1e3, fl / (tot / 5 * 1e-3) / 1e12);
#if A3_STAMPS
    std::vector<unsigned long long> st(3 * 16 * 8 * 16 * 16);
    hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_a3_stamps), st.size() * 8);
    const char* names[15] = {"top", "barrier", "setup", "t0", "t1", "t2", "t3", "t4", "t5", "t6", "t7", "tiles_end", "end", "q_issued", "stored"};
    for (int ph = 6; ph <= 7; ++ph)
        for (int w = 0; w < 8; w += 4) {
            printf("phase %d wave %d: cycles since phase top (median over 16 blocks):", ph, w);
            for (int i = 1; i < 15; ++i) {
                std::vector<long long> v;
                for (int b = 0; b < 16; ++b) { const unsigned long long* d = &st[(((size_t)b * 8 + w) * 16 + ph) * 16]; if (d[i] && d[0]) v.push_back((long long)(d[i] - d[0])); }
                std::sort(v.begin(), v.end());
                printf(" %s=%lld", names[i], v.empty() ? -1 : v[v.size() / 2]);
            }
            printf("\n");
        }
#if A3_STAMPS == 2
    for (int tile = 0; tile < 2; ++tile)
        for (int w = 0; w < 8; w += 4) {
            printf("phase 6, key tile %d, wave %d: cycles per stage 0..7 + tail (median over 16 blocks):", tile ? 6 : 1, w);
            for (int i = 1; i < 10; ++i) {
                std::vector<long long> v;
                for (int b = 0; b < 16; ++b) { const unsigned long long* d = &st[(size_t)(1 + tile) * 16 * 8 * 16 * 16 + (((size_t)b * 8 + w) * 16 + 6) * 16]; if (d[i] && d[i - 1]) v.push_back((long long)(d[i] - d[i - 1])); }
                std::sort(v.begin(), v.end());
                printf(" %lld", v.empty() ? -1 : v[v.size() / 2]);
            }
            printf("\n");
        }
#endif
#endif
    return 0;
}
