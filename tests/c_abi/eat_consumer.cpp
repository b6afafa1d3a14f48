// C-ABI consumer of the EAT ENCODER HANDLE (BASELINE config C3's model): no Python, no torch.  Reads a weight table and a waveform
// batch from a flat file written by tests/test_gpu_c_abi.py, builds an avexhip_eat handle, runs avexhip_eat_forward (features, class-token
// pooling, one attn.proj hook tap) and writes the results back for the test to compare with the Python wrapper on the same library and
// with the CPU oracle.
//   file: int32 n_tensors, then per tensor { int32 name_len, name bytes, int64 numel, fp32 data }, then int32 B, int64 T, fp32 wav[B*T],
//         then the avexhip_eat_config as raw bytes
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
    if (argc < 3) { std::printf("usage: eat_consumer <in.bin> <out.bin>\n"); return 1; }
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
    avexhip_eat_config cfg;
    if (!rd(wav.data(), sizeof(float) * wav.size()) || !rd(&cfg, sizeof(cfg))) return 1;
    std::fclose(f);

    std::vector<avexhip_tensor> table(nt);
    for (int i = 0; i < nt; ++i) { table[i].name = names[i].c_str(); table[i].data = data[i].data(); table[i].numel = (int64_t)data[i].size(); }
    avexhip_eat* h = avexhip_eat_create(&cfg, table.data(), nt);
    if (!h) { std::printf("create failed: %s\n", avexhip_last_error()); return 3; }
    const int Tt = avexhip_eat_num_tokens(h), E = cfg.embed_dim, L = cfg.depth;
    const size_t ws_bytes = avexhip_eat_workspace_bytes(h, B);
    float *dwav, *dfeat, *dpool, *dhook; void* ws;
    CK(hipMalloc(&dwav, sizeof(float) * wav.size())); CK(hipMalloc(&ws, ws_bytes));
    CK(hipMalloc(&dfeat, sizeof(float) * (size_t)B * Tt * E)); CK(hipMalloc(&dpool, sizeof(float) * (size_t)B * E));
    CK(hipMalloc(&dhook, sizeof(float) * (size_t)B * Tt * E));
    CK(hipMemcpy(dwav, wav.data(), sizeof(float) * wav.size(), hipMemcpyHostToDevice));
    std::vector<float*> hooks(L, nullptr);
    hooks[L - 1] = dhook;
    hipStream_t s;
    CK(hipStreamCreate(&s));
    AK(avexhip_eat_forward(h, dwav, B, T, T, nullptr, 1u << (L - 1), hooks.data(), 0, dfeat, dpool, 1, ws, ws_bytes, s));
    // refusals, not crashes: a too-small workspace, both inputs at once, pooling without a buffer
    if (avexhip_eat_forward(h, dwav, B, T, T, nullptr, 0, nullptr, 0, dfeat, nullptr, 0, ws, 16, s) != AVEXHIP_ERR_WORKSPACE) { std::printf("workspace check missing\n"); return 4; }
    if (avexhip_eat_forward(h, dwav, B, T, T, dfeat, 0, nullptr, 0, dfeat, nullptr, 0, ws, ws_bytes, s) == AVEXHIP_OK) { std::printf("wav + spec accepted\n"); return 4; }
    if (avexhip_eat_forward(h, dwav, B, T, T, nullptr, 0, nullptr, 0, dfeat, nullptr, 1, ws, ws_bytes, s) == AVEXHIP_OK) { std::printf("pooling without a buffer accepted\n"); return 4; }
    CK(hipStreamSynchronize(s));
    uint32_t ovf = 123;
    AK(avexhip_eat_overflow_count(h, &ovf, s, 1));
    if (ovf != 0) { std::printf("range alarm fired on O(1) activations: %u\n", ovf); return 5; }
    std::vector<float> feat((size_t)B * Tt * E), pool((size_t)B * E), hook((size_t)B * Tt * E);
    CK(hipMemcpy(feat.data(), dfeat, sizeof(float) * feat.size(), hipMemcpyDeviceToHost));
    CK(hipMemcpy(pool.data(), dpool, sizeof(float) * pool.size(), hipMemcpyDeviceToHost));
    CK(hipMemcpy(hook.data(), dhook, sizeof(float) * hook.size(), hipMemcpyDeviceToHost));
    FILE* o = std::fopen(argv[2], "wb");
    if (!o) return 1;
    const int32_t dims[3] = {B, Tt, E};
    std::fwrite(dims, 4, 3, o);
    std::fwrite(feat.data(), sizeof(float), feat.size(), o);
    std::fwrite(pool.data(), sizeof(float), pool.size(), o);
    std::fwrite(hook.data(), sizeof(float), hook.size(), o);
    std::fclose(o);
    avexhip_eat_destroy(h);
    std::printf("EAT CONSUMER OK\n");
    return 0;
}
