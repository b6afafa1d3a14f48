// C-ABI consumer of the ENCODER HANDLE: no Python, no torch.  Reads a weight table and a waveform batch from a flat file written
// by tests/test_gpu_c_abi.py, builds an avexhip_beats handle, runs avexhip_beats_forward (features, pooled, two hook taps) and writes
// the results back for the test to compare with the same checkpoint run through the Python classes and with the CPU oracle.
//   file: int32 n_tensors, then per tensor { int32 name_len, name bytes, int64 numel, fp32 data }, then int32 B, int64 T, fp32 wav[B*T],
//         then the avexhip_beats_config as raw bytes
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "avexhip.h"

#define CK(x) do { if ((x) != hipSuccess) { std::printf("HIP error at line %d\n", __LINE__); return 2; } } while (0)
#define AK(x) do { int rc_ = (x); if (rc_ != AVEXHIP_OK) { std::printf("avexhip error %d: %s\n", rc_, avexhip_last_error()); return 3; } } while (0)

int main(int argc, char** argv) {
    if (argc < 3) { std::printf("usage: beats_consumer <in.bin> <out.bin>\n"); return 1; }
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 1;
    auto rd = [&](void* p, size_t n) { return std::fread(p, 1, n, f) == n; };
    int32_t nt = 0;
    if (!rd(&nt, 4)) return 1;
    std::vector<std::string> names(nt);
    std::vector<std::vector<float>> data(nt);
    for (int i = 0; i < nt; ++i) {
        int32_t nl; int64_t ne;
        if (!rd(&nl, 4)) return 1;
        names[i].resize(nl);
        if (!rd(&names[i][0], nl) || !rd(&ne, 8)) return 1;
        data[i].resize((size_t)ne);
        if (!rd(data[i].data(), sizeof(float) * (size_t)ne)) return 1;
    }
    int32_t B; int64_t T;
    if (!rd(&B, 4) || !rd(&T, 8)) return 1;
    std::vector<float> wav((size_t)B * T);
    avexhip_beats_config cfg;
    if (!rd(wav.data(), sizeof(float) * wav.size()) || !rd(&cfg, sizeof(cfg))) return 1;
    std::fclose(f);

    std::vector<avexhip_tensor> table(nt);
    for (int i = 0; i < nt; ++i) { table[i].name = names[i].c_str(); table[i].data = data[i].data(); table[i].numel = (int64_t)data[i].size(); }
    avexhip_beats* h = avexhip_beats_create(&cfg, table.data(), nt);
    if (!h) { std::printf("create failed: %s\n", avexhip_last_error()); return 3; }
    const int Tt = avexhip_beats_num_tokens(h, T), E = cfg.encoder_embed_dim, L = cfg.encoder_layers;
    const size_t ws_bytes = avexhip_beats_workspace_bytes(h, B, T);
    float *dwav, *dfeat, *dpool, *dh0, *dhl; void* ws;
    CK(hipMalloc(&dwav, sizeof(float) * wav.size())); CK(hipMalloc(&ws, ws_bytes));
    CK(hipMalloc(&dfeat, sizeof(float) * (size_t)B * Tt * E)); CK(hipMalloc(&dpool, sizeof(float) * (size_t)B * E));
    CK(hipMalloc(&dh0, sizeof(float) * (size_t)B * Tt * E)); CK(hipMalloc(&dhl, sizeof(float) * (size_t)B * Tt * E));
    CK(hipMemcpy(dwav, wav.data(), sizeof(float) * wav.size(), hipMemcpyHostToDevice));
    std::vector<float*> hooks(L + 1, nullptr);
    hooks[0] = dh0; hooks[L] = dhl;
    hipStream_t s;
    CK(hipStreamCreate(&s));
    AK(avexhip_beats_forward(h, dwav, B, T, T, nullptr, (1u << 0) | (1u << L), hooks.data(), 0, dfeat, dpool, ws, ws_bytes, s));
    // a too-small workspace and a bad hook mask must be refused, not crash
    if (avexhip_beats_forward(h, dwav, B, T, T, nullptr, 0, nullptr, 0, dfeat, dpool, ws, 16, s) != AVEXHIP_ERR_WORKSPACE) { std::printf("workspace check missing\n"); return 4; }
    if (avexhip_beats_forward(h, dwav, B, T, T, nullptr, 1u << (L + 1), hooks.data(), 0, dfeat, dpool, ws, ws_bytes, s) != AVEXHIP_ERR_INVALID) { std::printf("hook mask check missing\n"); return 4; }
    CK(hipStreamSynchronize(s));
    std::vector<float> feat((size_t)B * Tt * E), pool((size_t)B * E), h0(feat.size()), hl(feat.size());
    CK(hipMemcpy(feat.data(), dfeat, sizeof(float) * feat.size(), hipMemcpyDeviceToHost));
    CK(hipMemcpy(pool.data(), dpool, sizeof(float) * pool.size(), hipMemcpyDeviceToHost));
    CK(hipMemcpy(h0.data(), dh0, sizeof(float) * h0.size(), hipMemcpyDeviceToHost));
    CK(hipMemcpy(hl.data(), dhl, sizeof(float) * hl.size(), hipMemcpyDeviceToHost));
    avexhip_beats_destroy(h);
    FILE* o = std::fopen(argv[2], "wb");
    if (!o) return 1;
    int32_t hdr[3] = {B, Tt, E};
    std::fwrite(hdr, 4, 3, o);
    std::fwrite(feat.data(), sizeof(float), feat.size(), o); std::fwrite(pool.data(), sizeof(float), pool.size(), o);
    std::fwrite(h0.data(), sizeof(float), h0.size(), o); std::fwrite(hl.data(), sizeof(float), hl.size(), o);
    std::fclose(o);
    std::printf("BEATS CONSUMER OK B=%d tokens=%d E=%d\n", B, Tt, E);
    return 0;
}
