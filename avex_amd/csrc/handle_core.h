// What the encoder handles (avexhip_beats in api.cpp; avexhip_eat / avexhip_aves in encoders.cpp) share: the weight-table helpers,
// the per-layer parameter block, and the post-LN transformer layer loop on the GEMM / attention / LayerNorm kernels -- with the
// LayerNorms folded into the GEMM epilogues (GemmArgs) when the residual stream is kept in the operand type.
//
// The three encoders differ only in what surrounds the loop (frontend, positional scheme) and in four parameters of it:
//   alpha     DeepNorm residual scale (BEATs: (2 L)^(1/4), backbone.py:304-308; EAT, wav2vec2: 1)
//   eps       LayerNorm epsilon (BEATs / wav2vec2 1e-5, EAT 1e-6)
//   bias/gate relative-position bias table + gate (BEATs only)
//   hook site which raw GEMM output a layer's hook taps: fc2 (BEATs backbone.encoder.layers.{i}.fc2, AVES ...output_dense) or
//             the attention output projection (EAT backbone.model.blocks.{i}.attn.proj)
#pragma once
#include <math.h>
#include <stdlib.h>

#include <map>
#include <string>
#include <vector>

#include "common.h"

namespace avxh {

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

struct StageRec {
    std::string name;
    double flops;
    hipEvent_t e0, e1;
};

// state every handle has: operand type, device allocations to free, the f16 range alarm, per-stage profiling
struct HandleBase {
    int dtype = AVEXHIP_F16;
    const char* who = "create";            // prefix of error messages
    std::vector<void*> allocs;
    // range alarm of the f16 conversions (GemmArgs::ovf): device counter every GEMM of a forward adds to, mirrored to pinned host
    // memory by an asynchronous copy at the end of each forward (read without a synchronisation by *_overflow_count)
    unsigned int* d_ovf = nullptr;
    unsigned int* h_ovf = nullptr;
    // log2(e) of the attention's exp2 folded into W_q / b_q in fp32 when the weights are packed (avx::attention's q_log2e): Q is rounded to the
    // operand type once and the scores are scaled by the exact 1/8.  AVEX_AMD_Q_LOG2E=0 keeps plain Q (the kernel then scales by a constant
    // it rounds to the operand type: 0.18 % off in bf16)
    bool q_log2e = !(getenv("AVEX_AMD_Q_LOG2E") && atoi(getenv("AVEX_AMD_Q_LOG2E")) == 0);
    bool profiling = false;
    std::vector<StageRec> recs;
    std::vector<std::string> prof_names;
    std::vector<const char*> prof_name_ptrs;
    std::vector<float> prof_ms;
    std::vector<double> prof_flops;

    int init_alarm() {
        if (hipMalloc((void**)&d_ovf, sizeof(unsigned int)) != hipSuccess || hipMemset(d_ovf, 0, sizeof(unsigned int)) != hipSuccess ||
            hipHostMalloc((void**)&h_ovf, sizeof(unsigned int), hipHostMallocDefault) != hipSuccess) {
            avexhip_set_error("%s: cannot allocate the range-alarm counter", who);
            return AVEXHIP_ERR_HIP;
        }
        *h_ovf = 0;
        return AVEXHIP_OK;
    }
    int mirror_alarm(hipStream_t s) {
        if (d_ovf && h_ovf) AVX_HIP_CHECK(hipMemcpyAsync(h_ovf, d_ovf, sizeof(unsigned int), hipMemcpyDeviceToHost, s));
        return AVEXHIP_OK;
    }
    int overflow_count(uint32_t* events, hipStream_t s, int synchronize) {
        if (synchronize) AVX_HIP_CHECK(hipStreamSynchronize(s));
        *events = h_ovf ? *(volatile unsigned int*)h_ovf : 0u;
        return AVEXHIP_OK;
    }
    int overflow_reset(hipStream_t s) {
        AVX_HIP_CHECK(hipMemsetAsync(d_ovf, 0, sizeof(unsigned int), s));
        AVX_HIP_CHECK(hipMemcpyAsync(h_ovf, d_ovf, sizeof(unsigned int), hipMemcpyDeviceToHost, s));
        return AVEXHIP_OK;
    }
    // After the weights are packed and before any forward: the alarm counter holds what the UPLOAD clipped.  An f16 handle whose weights do
    // not fit the f16 range would compute with saturated weights, silently (the alarm of a forward only sees activations): refuse it.
    int weights_fit() {
        if (!d_ovf) return AVEXHIP_OK;
        unsigned int n = 0;
        AVX_HIP_CHECK(hipDeviceSynchronize());
        AVX_HIP_CHECK(hipMemcpy(&n, d_ovf, sizeof(n), hipMemcpyDeviceToHost));
        if (n != 0) {
            avexhip_set_error("%s: the checkpoint's weights do not fit the f16 range (+-65504): %u lane(s) clipped a value while packing them (a folded LayerNorm gain "
                              "or hidden_shift counts).  Use operand_dtype bf16 (fp32's exponent range).", who, n);
            return AVEXHIP_ERR_INVALID;
        }
        return AVEXHIP_OK;
    }
    virtual ~HandleBase() {
        for (void* p : allocs) (void)hipFree(p);
        if (d_ovf) (void)hipFree(d_ovf);
        if (h_ovf) (void)hipHostFree(h_ovf);
        for (auto& r : recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    }
};

// HIP events around every stage of a forward when the handle is in profiling mode
struct Prof {
    HandleBase* h;
    hipStream_t s;
    size_t next = 0;
    void begin(const char* name, double flops) {
        if (!h->profiling) return;
        if (next == h->recs.size()) {
            StageRec r;
            (void)hipEventCreate(&r.e0);
            (void)hipEventCreate(&r.e1);
            h->recs.push_back(r);
        }
        h->recs[next].name = name;
        h->recs[next].flops = flops;
        (void)hipEventRecord(h->recs[next].e0, s);
    }
    void end() {
        if (!h->profiling) return;
        (void)hipEventRecord(h->recs[next].e1, s);
        ++next;
    }
    // after the forward: aggregate by stage name (forces a stream synchronisation)
    int collect() {
        if (!h->profiling) return AVEXHIP_OK;
        AVX_HIP_CHECK(hipStreamSynchronize(s));
        std::map<std::string, std::pair<double, double>> agg;  // name -> (ms, flops)
        std::vector<std::string> order;
        for (size_t i = 0; i < next; ++i) {
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, h->recs[i].e0, h->recs[i].e1);
            if (!agg.count(h->recs[i].name)) order.push_back(h->recs[i].name);
            agg[h->recs[i].name].first += ms;
            agg[h->recs[i].name].second += h->recs[i].flops;
        }
        h->prof_names = order;
        h->prof_ms.clear(); h->prof_flops.clear(); h->prof_name_ptrs.clear();
        for (auto& nm : h->prof_names) {
            h->prof_ms.push_back((float)agg[nm].first);
            h->prof_flops.push_back(agg[nm].second);
        }
        for (auto& nm : h->prof_names) h->prof_name_ptrs.push_back(nm.c_str());
        return AVEXHIP_OK;
    }
};

// name -> fp32 tensor table handed to *_create; a key matches with or without the wrapper prefixes of the reference's state dicts
struct Table {
    const avexhip_tensor* t;
    int n;
    const char* strip1 = "backbone.";      // avex/models/beats_model.py, eat_hf.py: self.backbone
    const char* strip2 = nullptr;          // e.g. "model." (EAT: backbone.model.*, AVES: model.*)
    const avexhip_tensor* find(const std::string& name) const {
        for (int i = 0; i < n; ++i) {
            if (!t[i].name) continue;
            const char* nm = t[i].name;
            if (strip1 && strncmp(nm, strip1, strlen(strip1)) == 0) nm += strlen(strip1);
            if (strip2 && strncmp(nm, strip2, strlen(strip2)) == 0) nm += strlen(strip2);
            if (name == nm) return &t[i];
        }
        return nullptr;
    }
};

// copy an fp32 tensor to the device (source may be host or device memory)
inline int dev_f32(HandleBase* h, const Table& tb, const std::string& name, int64_t numel, float** out, bool required = true) {
    const avexhip_tensor* t = tb.find(name);
    if (!t) {
        if (!required) { *out = nullptr; return AVEXHIP_OK; }
        avexhip_set_error("%s: tensor '%s' missing from the weight table", h->who, name.c_str());
        return AVEXHIP_ERR_MISSING;
    }
    if (t->numel != numel || !t->data) {
        avexhip_set_error("%s: tensor '%s' has %lld elements, expected %lld", h->who, name.c_str(), (long long)t->numel, (long long)numel);
        return AVEXHIP_ERR_INVALID;
    }
    float* d = nullptr;
    AVX_HIP_CHECK(hipMalloc((void**)&d, sizeof(float) * (size_t)numel));
    h->allocs.push_back(d);
    AVX_HIP_CHECK(hipMemcpy(d, t->data, sizeof(float) * (size_t)numel, hipMemcpyDefault));
    *out = d;
    return AVEXHIP_OK;
}

// gather an fp32 tensor of the table into a host vector (appending)
inline int host_f32(HandleBase* h, const Table& tb, const std::string& name, int64_t numel, std::vector<float>& out) {
    const avexhip_tensor* t = tb.find(name);
    if (!t || t->numel != numel || !t->data) {
        avexhip_set_error("%s: tensor '%s' missing or mis-sized", h->who, name.c_str());
        return AVEXHIP_ERR_MISSING;
    }
    const size_t o = out.size();
    out.resize(o + (size_t)numel);
    AVX_HIP_CHECK(hipMemcpy(out.data() + o, t->data, sizeof(float) * (size_t)numel, hipMemcpyDefault));
    return AVEXHIP_OK;
}

// host fp32 values -> half at dst (device), via a temporary fp32 device staging buffer
inline int upload_half(HandleBase* h, const float* host_or_dev, int64_t numel, void* dst, const char* what) {
    float* tmp = nullptr;
    AVX_HIP_CHECK(hipMalloc((void**)&tmp, sizeof(float) * (size_t)numel));
    hipError_t e = hipMemcpy(tmp, host_or_dev, sizeof(float) * (size_t)numel, hipMemcpyDefault);
    int rc = AVEXHIP_OK;
    if (e != hipSuccess) {
        avexhip_set_error("%s: copy of '%s' failed: %s", h->who, what, hipGetErrorString(e));
        rc = AVEXHIP_ERR_HIP;
    } else {
        rc = avx::cast_to_half(tmp, dst, numel, h->dtype, nullptr, h->d_ovf);      // (weights_fit below reads the counter when the handle is complete)
        if (rc == AVEXHIP_OK && hipDeviceSynchronize() != hipSuccess) {
            avexhip_set_error("%s: cast of '%s' failed", h->who, what);
            rc = AVEXHIP_ERR_HIP;
        }
    }
    (void)hipFree(tmp);
    return rc;
}

// fp32 tensor of the table -> half copy at dst (device)
inline int dev_half_into(HandleBase* h, const Table& tb, const std::string& name, int64_t numel, void* dst) {
    const avexhip_tensor* t = tb.find(name);
    if (!t) {
        avexhip_set_error("%s: tensor '%s' missing from the weight table", h->who, name.c_str());
        return AVEXHIP_ERR_MISSING;
    }
    if (t->numel != numel || !t->data) {
        avexhip_set_error("%s: tensor '%s' has %lld elements, expected %lld", h->who, name.c_str(), (long long)t->numel, (long long)numel);
        return AVEXHIP_ERR_INVALID;
    }
    return upload_half(h, (const float*)t->data, numel, dst, name.c_str());
}

inline int dev_half(HandleBase* h, const Table& tb, const std::string& name, int64_t numel, void** out) {
    void* d = nullptr;
    AVX_HIP_CHECK(hipMalloc(&d, 2 * (size_t)numel));
    h->allocs.push_back(d);
    *out = d;
    return dev_half_into(h, tb, name, numel, d);
}

// W' = half(W * diag(gamma)), b' = b + W beta, s[n] = sum_k float(W'[n][k]) for a consumer of LayerNorm(y; gamma, beta)
// (W: [N, K] fp32 host rows gathered from the table by the caller)
inline int fold_ln(HandleBase* h, const std::vector<float>& W, const std::vector<float>& b, int N, int K, const float* gamma_dev,
                   const float* beta_dev, void** w_out, float** b_out, float** s_out) {
    std::vector<float> gamma(K), beta(K), Wg((size_t)N * K), bf(N);
    AVX_HIP_CHECK(hipMemcpy(gamma.data(), gamma_dev, sizeof(float) * K, hipMemcpyDefault));
    AVX_HIP_CHECK(hipMemcpy(beta.data(), beta_dev, sizeof(float) * K, hipMemcpyDefault));
    for (int n = 0; n < N; ++n) {
        double acc = b[n];
        const float* wr = &W[(size_t)n * K];
        float* wo = &Wg[(size_t)n * K];
        for (int k = 0; k < K; ++k) { wo[k] = wr[k] * gamma[k]; acc += (double)wr[k] * (double)beta[k]; }
        bf[n] = (float)acc;
    }
    float* tmp = nullptr;
    AVX_HIP_CHECK(hipMalloc((void**)&tmp, sizeof(float) * (size_t)N * K));
    void* wd = nullptr; float* bd = nullptr; float* sd = nullptr;
    int rc = AVEXHIP_OK;
    if (hipMalloc(&wd, 2 * (size_t)N * K) != hipSuccess || hipMalloc((void**)&bd, sizeof(float) * N) != hipSuccess ||
        hipMalloc((void**)&sd, sizeof(float) * N) != hipSuccess) {
        avexhip_set_error("%s: device allocation for folded weights failed", h->who);
        rc = AVEXHIP_ERR_HIP;
    }
    if (wd) h->allocs.push_back(wd);
    if (bd) h->allocs.push_back(bd);
    if (sd) h->allocs.push_back(sd);
    if (rc == AVEXHIP_OK && (hipMemcpy(tmp, Wg.data(), sizeof(float) * (size_t)N * K, hipMemcpyHostToDevice) != hipSuccess ||
                             hipMemcpy(bd, bf.data(), sizeof(float) * N, hipMemcpyHostToDevice) != hipSuccess)) {
        avexhip_set_error("%s: upload of folded weights failed", h->who);
        rc = AVEXHIP_ERR_HIP;
    }
    if (rc == AVEXHIP_OK) rc = avx::cast_to_half(tmp, wd, (int64_t)N * K, h->dtype, nullptr, h->d_ovf);
    if (rc == AVEXHIP_OK) rc = avx::row_sum_half(wd, N, K, sd, h->dtype, nullptr);
    if (rc == AVEXHIP_OK && hipDeviceSynchronize() != hipSuccess) { avexhip_set_error("%s: folding failed", h->who); rc = AVEXHIP_ERR_HIP; }
    (void)hipFree(tmp);
    *w_out = wd; *b_out = bd; *s_out = sd;
    return rc;
}

// ---------------------------------------------------------------------------------------------
// One post-LN transformer layer's parameters on the device
// ---------------------------------------------------------------------------------------------
struct Layer {
    void* w_qkv = nullptr; float* b_qkv = nullptr;
    void* w_o = nullptr;   float* b_o = nullptr;
    float* grep_w = nullptr; float* grep_b = nullptr; float* grep_a = nullptr;
    float* ln1_w = nullptr; float* ln1_b = nullptr;
    void* w_fc1 = nullptr; float* b_fc1 = nullptr;
    void* w_fc2 = nullptr; float* b_fc2 = nullptr;
    float* ln2_w = nullptr; float* ln2_b = nullptr;
    // LayerNorm-folded copies (see GemmArgs): fc1 consumes LN1 of this layer, QKV consumes LN2 of the previous layer
    void* w_fc1_f = nullptr; float* b_fc1_f = nullptr; float* s_fc1 = nullptr;
    void* w_qkv_f = nullptr; float* b_qkv_f = nullptr; float* s_qkv = nullptr;
    // residual-side folds (GemmArgs::lnr_prefolded): fc2 adds alpha * LN1(y1) of this layer, out_proj alpha * LN2(y2) of the previous layer;
    // ga = alpha * gamma, bb = bias + alpha * beta
    float* ga_fc2 = nullptr; float* bb_fc2 = nullptr;
    float* ga_o = nullptr; float* bb_o = nullptr;
};

// parameter names of layer i under a family's naming scheme ("%d" = layer index)
struct LayerNames {
    const char* qkv_fused;        // "blocks.%d.attn.qkv" ([3E, E] in one tensor), or NULL when q / k / v are separate:
    const char* q; const char* k; const char* v;
    const char* out_proj; const char* ln1; const char* fc1; const char* fc2; const char* ln2;
    const char* grep_linear; const char* grep_a;      // BEATs' gated relative position bias, or NULL
};

inline std::string fmt_name(const char* pattern, int i, const char* suffix) {
    char buf[256];
    snprintf(buf, sizeof(buf), pattern, i);
    return std::string(buf) + suffix;
}

struct CoreCfg {
    int E = 0, F = 0, H = 0, L = 0;   // F = 0: attention-only blocks, x = LN1(x + attn(x)) (the attention probe's layers, attention_probe.py:127-130)
    int head_dim = 64;                // 64: attention.hip (relative bias, gate, streaming); 32 / 96 / 128: attention_hd.hip (plain softmax attention)
    float alpha = 1.f, eps = 1e-5f;
    int hook_site = 0;            // 0: fc2's raw output, 1: the attention output projection's raw output
    bool fast = false;            // residual stream / pre-LN sums in the operand type
    bool fold = false;            // fast mode: the LayerNorms between the GEMMs folded into their epilogues (post-LN, GELU / SiLU FFN only)
    int fold_min_rows = 0;        // ... for chunks of at least this many token rows (default 4096; AVEX_AMD_LN_FOLD=1: 0 = always, so that a
                                  // clip's result never depends on the batch it came in)
    int act = 1;                  // GemmArgs::gelu code of the FFN activation (0 none, 1 erf GELU, 2 SiLU, 3 ReLU, 4 tanh GELU, 5 tanh)
    bool glu = false;             // fc1 is the reference's GLU_Linear(E, F, "swish"): one Linear to 2F, then value * swish(gate) (backbone.py:296-297)
    int hidden_shift = 0;         // > 0: fc1's hidden activations are stored x 2^-shift and fc2's weights packed x 2^shift (exact in the fp32 accumulation):
                                  // the f16 range ladder's rung for hidden activations beyond 65504 (the reference is fp32, backbone.py:365-370)
    bool pre_ln = false;          // pre-LN blocks (backbone.py:328-348): x += attn(LN1 x); x += ffn(LN2 x); `final_ln` after the stack (:146-147)
    bool batch_invariant = false; // residual_dtype bit 1 (AVEXHIP_RESIDUAL_BATCH_INVARIANT): a clip's outputs must not depend on the batch it arrives in --
                                  // fold at every size, no split-K, no LayerNorm inside a split-K epilogue, one final LayerNorm + pool path
    const float* final_ln_w = nullptr; const float* final_ln_b = nullptr;
};

// AVEX_AMD_LN_FOLD, read when a handle is created.  Unset / "auto": the encoder's LayerNorms are folded into the GEMMs around them (dims
// permitting) for chunks of >= 4096 token rows; smaller chunks run LayerNorm kernels.  Below that size the fold's 256-tile streaming kernel
// has a handful of tiles for 256 CUs and the 128-tile kernel (split-K for fc2) is quicker: one 10 s clip 2.17 -> 1.40 ms, four clips
// 2.22 -> 1.66, eight 2.28 -> 2.00; from sixteen clips on (7 936 rows) the fold is level or ahead (profiles/r03r_midsize.txt).
// "1": fold whatever the size -- a clip's embedding is then bit-identical whether it came alone or in a batch of 256 (with "auto" the two
// differ in their last bits, both inside the parity bar).  "0": never fold.
// residual_dtype of the handle configs: bit 0 = residual stream in the operand type, bit 1 = batch-invariant results
inline bool cfg_fast(int residual_dtype) { return (residual_dtype & 1) != 0; }
inline bool cfg_batch_invariant(int residual_dtype) {
    const char* e = getenv("AVEX_AMD_BATCH_INVARIANT");      // the environment can only turn it ON (for callers that cannot reach the config)
    return (residual_dtype & 2) != 0 || (e && atoi(e) != 0);
}
inline void fold_policy(bool fast, int E, int F, bool* fold, int* min_rows, bool batch_invariant = false) {
    const char* e = getenv("AVEX_AMD_LN_FOLD");
    const bool is_auto = !e || e[0] == 'a' || e[0] == 'A';
    *fold = fast && E % 256 == 0 && F % 256 == 0 && !(e && !is_auto && atoi(e) == 0);
    int rows = 4096;
    if (e && is_auto) { const char* c = strchr(e, ':'); if (c && atoi(c + 1) > 0) rows = atoi(c + 1); }      // "auto:4096": another threshold (experiments)
    *min_rows = is_auto && !batch_invariant ? rows : 0;
}

// upload layer i (and, with the fold, its LayerNorm-folded copies; layer i - 1 must have been built)
inline int build_layer(HandleBase* h, const Table& tb, const LayerNames& nm, const CoreCfg& c, std::vector<Layer>& layers, int i) {
    const int E = c.E, F = c.F, H = c.H;
    Layer& ly = layers[i];
    int rc;
#define RC(x) do { rc = (x); if (rc != AVEXHIP_OK) return rc; } while (0)
    AVX_HIP_CHECK(hipMalloc(&ly.w_qkv, 2 * (size_t)3 * E * E));
    h->allocs.push_back(ly.w_qkv);
    AVX_HIP_CHECK(hipMalloc((void**)&ly.b_qkv, sizeof(float) * 3 * E));
    h->allocs.push_back(ly.b_qkv);
    std::vector<float> Wqkv, bqkv;          // host copies: rows [0, E) are W_q (scaled by log2(e) here, in fp32), kept for the LayerNorm fold
    if (nm.qkv_fused) {
        RC(host_f32(h, tb, fmt_name(nm.qkv_fused, i, ".weight"), (int64_t)3 * E * E, Wqkv));
        RC(host_f32(h, tb, fmt_name(nm.qkv_fused, i, ".bias"), 3 * E, bqkv));
    } else {
        const char* part[3] = {nm.q, nm.k, nm.v};
        for (int j = 0; j < 3; ++j) {
            RC(host_f32(h, tb, fmt_name(part[j], i, ".weight"), (int64_t)E * E, Wqkv));
            RC(host_f32(h, tb, fmt_name(part[j], i, ".bias"), E, bqkv));
        }
    }
    if (h->q_log2e) {
        const float l2e = 1.4426950408889634f;
        for (size_t j = 0; j < (size_t)E * E; ++j) Wqkv[j] *= l2e;
        for (int j = 0; j < E; ++j) bqkv[j] *= l2e;
    }
    RC(upload_half(h, Wqkv.data(), (int64_t)3 * E * E, ly.w_qkv, "qkv.weight"));
    AVX_HIP_CHECK(hipMemcpy(ly.b_qkv, bqkv.data(), sizeof(float) * 3 * E, hipMemcpyHostToDevice));
    RC(dev_half(h, tb, fmt_name(nm.out_proj, i, ".weight"), (int64_t)E * E, &ly.w_o));
    RC(dev_f32(h, tb, fmt_name(nm.out_proj, i, ".bias"), E, &ly.b_o));
    if (nm.grep_linear) {
        RC(dev_f32(h, tb, fmt_name(nm.grep_linear, i, ".weight"), 8 * (E / H), &ly.grep_w));
        RC(dev_f32(h, tb, fmt_name(nm.grep_linear, i, ".bias"), 8, &ly.grep_b));
        RC(dev_f32(h, tb, fmt_name(nm.grep_a, i, ""), H, &ly.grep_a));
    }
    RC(dev_f32(h, tb, fmt_name(nm.ln1, i, ".weight"), E, &ly.ln1_w));
    RC(dev_f32(h, tb, fmt_name(nm.ln1, i, ".bias"), E, &ly.ln1_b));
    if (F == 0) return AVEXHIP_OK;         // attention-only block
    const int F1 = c.glu ? 2 * F : F;      // GLU_Linear keeps its Linear(E, 2F) under ".linear"
    RC(dev_half(h, tb, fmt_name(nm.fc1, i, c.glu ? ".linear.weight" : ".weight"), (int64_t)F1 * E, &ly.w_fc1));
    RC(dev_f32(h, tb, fmt_name(nm.fc1, i, c.glu ? ".linear.bias" : ".bias"), F1, &ly.b_fc1));
    if (c.hidden_shift > 0) {
        // fc2's weights x 2^shift: the power of two is exact in fp32 and in the operand type unless a weight leaves its range
        std::vector<float> W2;
        RC(host_f32(h, tb, fmt_name(nm.fc2, i, ".weight"), (int64_t)E * F, W2));
        const float sc = ldexpf(1.0f, c.hidden_shift);
        float mx = 0.f;
        for (float& v : W2) { v *= sc; mx = fabsf(v) > mx ? fabsf(v) : mx; }
        if (h->dtype == AVEXHIP_F16 && !(mx <= 65504.0f)) {
            avexhip_set_error("%s: hidden_shift %d takes layer %d's fc2 weights out of the f16 range (max |w| 2^shift = %g)", h->who, c.hidden_shift, i, (double)mx);
            return AVEXHIP_ERR_INVALID;
        }
        AVX_HIP_CHECK(hipMalloc(&ly.w_fc2, 2 * (size_t)E * F));
        h->allocs.push_back(ly.w_fc2);
        RC(upload_half(h, W2.data(), (int64_t)E * F, ly.w_fc2, "fc2.weight x 2^shift"));
    } else {
        RC(dev_half(h, tb, fmt_name(nm.fc2, i, ".weight"), (int64_t)E * F, &ly.w_fc2));
    }
    RC(dev_f32(h, tb, fmt_name(nm.fc2, i, ".bias"), E, &ly.b_fc2));
    RC(dev_f32(h, tb, fmt_name(nm.ln2, i, ".weight"), E, &ly.ln2_w));
    RC(dev_f32(h, tb, fmt_name(nm.ln2, i, ".bias"), E, &ly.ln2_b));
    if (c.fold) {
        auto two = [&](float** ga, float** bb) -> int {
            AVX_HIP_CHECK(hipMalloc((void**)ga, sizeof(float) * 2 * (size_t)E));
            h->allocs.push_back(*ga);
            *bb = *ga + E;
            return AVEXHIP_OK;
        };
        RC(two(&ly.ga_fc2, &ly.bb_fc2));
        RC(avx::lnr_fold(ly.ln1_w, ly.ln1_b, ly.b_fc2, c.alpha, E, ly.ga_fc2, ly.bb_fc2, nullptr));
        std::vector<float> Wh, bh;
        RC(host_f32(h, tb, fmt_name(nm.fc1, i, ".weight"), (int64_t)F * E, Wh));
        RC(host_f32(h, tb, fmt_name(nm.fc1, i, ".bias"), F, bh));
        RC(fold_ln(h, Wh, bh, F, E, ly.ln1_w, ly.ln1_b, &ly.w_fc1_f, &ly.b_fc1_f, &ly.s_fc1));
        if (i > 0) {   // QKV and out_proj of layer i read LN2 of layer i - 1
            const Layer& prev = layers[i - 1];
            RC(two(&ly.ga_o, &ly.bb_o));
            RC(avx::lnr_fold(prev.ln2_w, prev.ln2_b, ly.b_o, c.alpha, E, ly.ga_o, ly.bb_o, nullptr));
            RC(fold_ln(h, Wqkv, bqkv, 3 * E, E, prev.ln2_w, prev.ln2_b, &ly.w_qkv_f, &ly.b_qkv_f, &ly.s_qkv));
        }
    }
#undef RC
    return AVEXHIP_OK;
}

// ---------------------------------------------------------------------------------------------
// The layer loop's buffers inside a caller-provided workspace, and the loop itself
// ---------------------------------------------------------------------------------------------
struct CoreWs {
    float* x; char* xh; float* pre; char* preh; char* qkv; char* ah; char* hh; float* raw;
    char* hh2;                // GLU: fc1's [M, 2F] output before the gate
    float* pool;              // mean-pooled hook taps: per-block column sums [ceil(M / 64)][2][E] (GemmArgs::pool_part)
    float* splitk; size_t splitk_bytes;      // split-K partials of the few-row products (GemmArgs::splitk_ws): 8 x min(M, 1024) x E floats
    float* st1; float* st2;   // folded LayerNorm: per-row partial statistics [M][E/64][2] of y1 (preh) and y2 (xh)
    float* r1; float* r2;     // ... reduced to (rstd, -mu rstd) per row by avx::ln_rowstats
};

// `take(bytes)` hands out consecutive aligned pieces of the workspace (NULL base: sizes only)
template <typename Take>
inline CoreWs carve_core(const CoreCfg& c, size_t M, Take&& take) {
    CoreWs w;
    w.x = (float*)take(M * c.E * 4);
    w.xh = (char*)take(M * c.E * 2);
    w.pre = (float*)take(c.fast ? 256 : M * c.E * 4);
    w.preh = (char*)take(c.fast ? M * c.E * 2 : 256);
    w.qkv = (char*)take(M * 3 * c.E * 2);
    w.ah = (char*)take(M * c.E * 2);
    w.hh = (char*)take(M * c.F * 2);
    w.hh2 = (char*)take(c.glu ? M * c.F * 4 : 256);
    w.pool = (float*)take(((M + 63) / 64) * 2 * (size_t)c.E * 4);
    w.splitk_bytes = (8 * M < 16384 ? 8 * M : 16384) * (size_t)c.E * 4;      // 8 splits up to 2 048 rows, 2 up to 8 192
    w.splitk = (float*)take(w.splitk_bytes);
    w.raw = (float*)take(M * c.E * 4);
    w.st1 = (float*)take(c.fold ? M * (c.E / 64) * 8 : 256);
    w.st2 = (float*)take(c.fold ? M * (c.E / 64) * 8 : 256);
    w.r1 = (float*)take(c.fold ? (M + 256) * 8 : 256);
    w.r2 = (float*)take(c.fold ? (M + 256) * 8 : 256);
    return w;
}

struct CoreIo {
    int Bc = 0, Tt = 0;
    size_t c0 = 0;                        // first clip of this chunk within the caller's batch (offset of the outputs)
    const float* bias_tab = nullptr;      // [H, 2 Tt - 1] or NULL
    const uint8_t* pad = nullptr;         // [Bc, Tt] key padding or NULL
    uint32_t hook_mask = 0;               // bit (hook_bit0 + i) selects layer i
    int hook_bit0 = 0;
    float* const* hook_out = nullptr;     // indexed like the mask's bits
    int hook_pooled = 0;
    float* features_out = nullptr;        // caller's [B, Tt, E] or NULL
    float* pooled_out = nullptr;          // caller's [B, E] (mean over the Tt tokens) or NULL
    float* final_f32 = nullptr;           // out: where the fp32 features of this chunk were written (caller's buffer or w.x), NULL if nowhere
};

// A hooked GEMM: the raw tap goes to the caller's [B, T, E] buffer, or -- mean-pooled taps of clips with >= 64 tokens on the 256-tile
// kernel -- never exists: the epilogue leaves per-block column sums and tap_finish reduces them to [B, E].  Shorter clips / narrow
// outputs write the tap to scratch and pool it from there.
struct Tap {
    bool hooked = false, fused = false;
    float* out = nullptr;      // caller's buffer for this chunk
};
inline Tap tap_begin(const CoreCfg& c, const CoreWs& w, const CoreIo& io, int layer, avx::GemmArgs& g) {
    Tap t;
    t.hooked = (io.hook_mask >> (io.hook_bit0 + layer)) & 1u;
    if (!t.hooked) return t;
    const size_t per_clip = io.hook_pooled ? (size_t)c.E : (size_t)io.Tt * c.E;
    t.out = io.hook_out[io.hook_bit0 + layer] + io.c0 * per_clip;
    static const bool no_fuse = getenv("AVEX_AMD_POOL_FUSE") && atoi(getenv("AVEX_AMD_POOL_FUSE")) == 0;
    // io.hook_pooled: 1 mean, 2 max, 3 first token over a clip's rows (extract_embeddings' aggregations, beats_model.py:403-417)
    // ... where that kernel is the one the product takes anyway (a folded LayerNorm, or enough tiles): a pooled tap must not change which
    // kernel computes the layer, or its max / first-token values would differ in their last bits from the same reduction of the full tap
    const bool streams = g.ln_rows || g.lnr_y || g.stats_out || avx::gemm_streams(g.M, g.N);
    // batch-invariant handles do not fuse the MEAN: its partial sums are grouped by 64-row blocks of the whole batch, so a clip's mean would
    // depend on where the clip starts (maxima and first rows are exact and stay fused)
    t.fused = io.hook_pooled && io.Tt >= 64 && c.E % 256 == 0 && g.K >= 128 && streams && !no_fuse && !(c.batch_invariant && io.hook_pooled == 1);
    if (t.fused) {
        g.pool_T = io.Tt; g.pool_mode = io.hook_pooled - 1;
        g.pool_part = io.hook_pooled == 3 ? t.out : w.pool;      // the first rows go straight to the caller's [B, E]
    } else { g.out_raw = io.hook_pooled ? w.raw : t.out; g.ldraw = c.E; }
    return t;
}
inline int tap_finish(const Tap& t, const CoreCfg& c, const CoreWs& w, const CoreIo& io, hipStream_t cs) {
    if (!t.hooked || !io.hook_pooled) return AVEXHIP_OK;
    if (t.fused) return io.hook_pooled == 3 ? AVEXHIP_OK : avx::pool_reduce(w.pool, io.Bc, io.Tt, c.E, t.out, c.E, cs, io.hook_pooled - 1);
    return avx::agg_pool(w.raw, io.Bc, io.Tt, c.E, io.hook_pooled, t.out, cs);
}

// batch-invariant handles: every layer product in the 256-tile streaming kernel whatever its row count (the 128-tile kernel's residual
// epilogue multiplies and adds where the streaming kernel's fuses, and its split-K adds in another order)
inline void pin_kernel(const CoreCfg& c, avx::GemmArgs& g) {
    if (c.batch_invariant && g.variant == 0 && g.N % 256 == 0 && g.K >= 128 && g.M >= 2) g.variant = 5;
}

inline int self_attention(HandleBase* h, const CoreCfg& c, const Layer& ly, const CoreWs& w, const CoreIo& io, hipStream_t cs) {
    if (c.head_dim == 64)
        return avx::attention(w.qkv, io.Bc, io.Tt, c.H, io.bias_tab, ly.grep_w, ly.grep_b, ly.grep_a, io.pad, w.ah, h->dtype, cs, h->q_log2e ? 1 : 0);
    return avx::attention_hd(w.qkv, io.Bc, io.Tt, c.H, c.head_dim, io.pad, w.ah, h->dtype, cs, h->q_log2e ? 1 : 0);
}

// fc1 (+ activation) of a prepared GemmArgs `g` (A, W, bias, fold fields set; N = F, output w.hh): plain, or the gated linear unit
inline int ffn_hidden(HandleBase* h, const CoreCfg& c, const CoreWs& w, avx::GemmArgs& g, int M, Prof& prof, hipStream_t cs) {
    const double flops = 2.0 * (double)M * (c.glu ? 2 * c.F : c.F) * c.E;
    int rc;
    if (c.glu) { g.N = 2 * c.F; g.gelu = 0; g.out_half = w.hh2; g.ldh = 2 * c.F; }
    if (c.hidden_shift > 0) g.half_scale = ldexpf(1.0f, -c.hidden_shift);      // (refused with GLU when the handle is created)
    prof.begin("gemm.fc1", flops);
    pin_kernel(c, g);
    rc = avx::gemm(g, h->dtype, cs);
    prof.end();
    if (rc != AVEXHIP_OK || !c.glu) return rc;
    prof.begin("glu", 0.0);
    rc = avx::glu_swish(w.hh2, M, c.F, w.hh, h->d_ovf, h->dtype, cs);
    prof.end();
    return rc;
}

// Pre-LN blocks (backbone.py:328-348).  The un-normalised stream starts in w.pre (fp32 residual stream) / w.preh (fast) -- where the
// positional convolution left x + pos_conv(x) -- and is back there after every layer; each LayerNorm writes the GEMM operand to w.ah.
inline int run_layers_pre_ln(HandleBase* h, const CoreCfg& c, const std::vector<Layer>& layers, const CoreWs& w, CoreIo& io, Prof& prof, hipStream_t cs) {
    const int E = c.E, F = c.F, H = c.H, L = c.L, dt = h->dtype, Bc = io.Bc, Tt = io.Tt;
    const int M = Bc * Tt;
    const double Md = (double)M;
    const bool fast = c.fast;
    float* s0_32 = fast ? nullptr : w.pre;  void* s0_h = fast ? w.preh : nullptr;      // stream at layer boundaries
    float* s1_32 = fast ? nullptr : w.x;    void* s1_h = fast ? w.xh : nullptr;        // stream after the attention block
    int rc;
#define RC(x) do { rc = (x); if (rc != AVEXHIP_OK) return rc; } while (0)
    avx::GemmArgs g;
    io.final_f32 = nullptr;
    for (int i = 0; i < L; ++i) {
        const Layer& ly = layers[i];
        prof.begin("layernorm", 0.0);
        RC(avx::layernorm(s0_32, s0_h, E, ly.ln1_w, ly.ln1_b, c.eps, M, E, nullptr, E, w.ah, E, dt, cs));
        prof.end();
        memset(&g, 0, sizeof(g)); g.ovf = h->d_ovf;
        g.A = w.ah; g.lda = E; g.W = ly.w_qkv; g.ldw = E; g.M = M; g.N = 3 * E; g.K = E; g.bias = ly.b_qkv;
        g.out_half = w.qkv; g.ldh = 3 * E;
        prof.begin("gemm.qkv", 2.0 * Md * 3 * E * E);
        pin_kernel(c, g); RC(avx::gemm(g, dt, cs));
        prof.end();
        prof.begin("attention", 4.0 * Md * Tt * E + (ly.grep_w ? 2.0 * Md * 8 * (E / H) * H : 0.0));
        RC(self_attention(h, c, ly, w, io, cs));
        prof.end();
        memset(&g, 0, sizeof(g)); g.ovf = h->d_ovf;
        g.A = w.ah; g.lda = E; g.W = ly.w_o; g.ldw = E; g.M = M; g.N = E; g.K = E; g.bias = ly.b_o; g.alpha = c.alpha;
        if (fast) { g.resid_half = s0_h; g.ldrh = E; g.out_half = s1_h; g.ldh = E; }
        else { g.resid = s0_32; g.ldr = E; g.out_f32 = s1_32; g.ldo = E; }
        Tap tap_o;
        if (c.hook_site == 1) tap_o = tap_begin(c, w, io, i, g);
        prof.begin("gemm.out_proj", 2.0 * Md * E * E);
        pin_kernel(c, g); RC(avx::gemm(g, dt, cs));
        prof.end();
        RC(tap_finish(tap_o, c, w, io, cs));
        prof.begin("layernorm", 0.0);
        RC(avx::layernorm(s1_32, s1_h, E, ly.ln2_w, ly.ln2_b, c.eps, M, E, nullptr, E, w.ah, E, dt, cs));
        prof.end();
        memset(&g, 0, sizeof(g)); g.ovf = h->d_ovf;
        g.A = w.ah; g.lda = E; g.W = ly.w_fc1; g.ldw = E; g.M = M; g.N = F; g.K = E; g.bias = ly.b_fc1; g.gelu = c.act;
        g.out_half = w.hh; g.ldh = F;
        RC(ffn_hidden(h, c, w, g, M, prof, cs));
        memset(&g, 0, sizeof(g)); g.ovf = h->d_ovf;
        g.A = w.hh; g.lda = F; g.W = ly.w_fc2; g.ldw = F; g.M = M; g.N = E; g.K = F; g.bias = ly.b_fc2; g.alpha = c.alpha;
        if (M <= 8192 && !c.batch_invariant) { g.splitk_ws = w.splitk; g.splitk_bytes = w.splitk_bytes; }
        if (fast) { g.resid_half = s1_h; g.ldrh = E; g.out_half = s0_h; g.ldh = E; }
        else { g.resid = s1_32; g.ldr = E; g.out_f32 = s0_32; g.ldo = E; }
        Tap tap_f;
        if (c.hook_site == 0) tap_f = tap_begin(c, w, io, i, g);
        prof.begin("gemm.fc2", 2.0 * Md * E * F);
        pin_kernel(c, g); RC(avx::gemm(g, dt, cs));
        prof.end();
        RC(tap_finish(tap_f, c, w, io, cs));
    }
    if (L > 0 && (io.features_out || io.pooled_out)) {      // the encoder's LayerNorm after the stack (backbone.py:146-147)
        const bool fused_pool = io.pooled_out && !io.features_out && fast && E % 8 == 0 && E <= 768 && (Bc >= 32 || c.batch_invariant);
        if (fused_pool) {
            prof.begin("layernorm+mean_pool", 0.0);
            RC(avx::layernorm_pool(s0_h, E, c.final_ln_w, c.final_ln_b, c.eps, Bc, Tt, E, io.pooled_out + io.c0 * E, dt, cs));
            prof.end();
        } else {
            float* xo = io.features_out ? io.features_out + io.c0 * Tt * E : w.x;
            prof.begin("layernorm", 0.0);
            RC(avx::layernorm(s0_32, s0_h, E, c.final_ln_w, c.final_ln_b, c.eps, M, E, xo, E, nullptr, E, dt, cs));
            prof.end();
            io.final_f32 = xo;
            if (io.pooled_out) {
                prof.begin("mean_pool", 0.0);
                RC(avx::mean_pool(xo, Bc, Tt, E, nullptr, io.pooled_out + io.c0 * E, cs));
                prof.end();
            }
        }
    }
#undef RC
    return AVEXHIP_OK;
}

// x (the encoder input after its own LayerNorm) is in w.xh (fast) / w.x + w.xh (fp32 residual stream); runs the L layers and the outputs
inline int run_layers(HandleBase* h, const CoreCfg& c, const std::vector<Layer>& layers, const CoreWs& w, CoreIo& io, Prof& prof, hipStream_t cs) {
    const int E = c.E, F = c.F, H = c.H, L = c.L, dt = h->dtype, Bc = io.Bc, Tt = io.Tt;
    const int M = Bc * Tt;
    const double Md = (double)M;
    const bool fast = c.fast;
    if (c.pre_ln) return run_layers_pre_ln(h, c, layers, w, io, prof, cs);
    float* x32 = w.x;
    float* pre32 = fast ? nullptr : w.pre;
    void* preh = fast ? w.preh : nullptr;
    int rc;
#define RC(x) do { rc = (x); if (rc != AVEXHIP_OK) return rc; } while (0)
    // "fold": the two LayerNorms of a layer never run as kernels.  y1 = x*alpha + attn (preh) and y2 = x1*alpha + ffn (xh) stay raw
    // in the operand type with per-row partial statistics from the epilogue that wrote them; fc1 / the next QKV read them
    // through LayerNorm-folded weights, out_proj / fc2 apply LayerNorm to their residual on the fly (GemmArgs, gemm.hip).
    const bool fold = fast && c.fold && F > 0 && M >= c.fold_min_rows;      // default: any M, the same arithmetic whatever the chunking
    const int nseg = E / 64;
    avx::GemmArgs g;
    io.final_f32 = nullptr;
    for (int i = 0; i < L; ++i) {
        const Layer& ly = layers[i];
        const bool raw_in = fold && i > 0;      // xh holds y2 of layer i-1 (raw) instead of its LayerNorm
        memset(&g, 0, sizeof(g)); g.ovf = h->d_ovf;
        g.A = w.xh; g.lda = E; g.W = ly.w_qkv; g.ldw = E; g.M = M; g.N = 3 * E; g.K = E; g.bias = ly.b_qkv;
        g.out_half = w.qkv; g.ldh = 3 * E;
        if (raw_in) { g.W = ly.w_qkv_f; g.bias = ly.b_qkv_f; g.ln_rows = w.r2; g.ln_s = ly.s_qkv; }
        prof.begin("gemm.qkv", 2.0 * Md * 3 * E * E);
        pin_kernel(c, g); RC(avx::gemm(g, dt, cs));
        prof.end();
        prof.begin("attention", 4.0 * Md * Tt * E + (ly.grep_w ? 2.0 * Md * 8 * (E / H) * H : 0.0));
        RC(self_attention(h, c, ly, w, io, cs));
        prof.end();
        memset(&g, 0, sizeof(g)); g.ovf = h->d_ovf;
        g.A = w.ah; g.lda = E; g.W = ly.w_o; g.ldw = E; g.M = M; g.N = E; g.K = E; g.bias = ly.b_o; g.alpha = c.alpha;
        if (fast) { g.resid_half = w.xh; g.ldrh = E; g.out_half = preh; g.ldh = E; }
        else { g.resid = x32; g.ldr = E; g.out_f32 = pre32; g.ldo = E; }
        if (fold) {
            g.stats_out = w.st1;
            g.rows_out = w.r1; g.rows_eps = c.eps;      // the product finishes LN1's row statistics itself (gemm_row.hip at N = 768; otherwise avx::gemm runs ln_rowstats behind it)
            if (raw_in) {
                g.resid_half = nullptr; g.ldrh = 0;
                g.lnr_y = w.xh; g.ldy = E; g.lnr_rows = w.r2; g.lnr_gamma = ly.ga_o; g.lnr_beta = ly.bb_o; g.lnr_prefolded = 1;
            }
        }
        Tap tap_o;
        if (c.hook_site == 1) tap_o = tap_begin(c, w, io, i, g);
        // few rows, no fold: LayerNorm 1 rides in the split-K epilogue of the product (one kernel less per layer; x32 / xh are read as the
        // residual and rewritten row by row by the same wave)
        bool ln1_fused = false;
        if (!fold && F > 0 && M <= 8192 && !c.batch_invariant) {
            g.splitk_ws = w.splitk; g.splitk_bytes = w.splitk_bytes;
            if (avx::gemm_post_ln_ok(g)) {
                g.post_ln_w = ly.ln1_w; g.post_ln_b = ly.ln1_b; g.post_ln_eps = c.eps; g.post_ln_round = fast ? 1 : 0;
                g.post_ln_out_f32 = fast ? nullptr : x32; g.post_ln_ldo = E; g.post_ln_out_half = w.xh; g.post_ln_ldh = E;
                ln1_fused = true;
            } else { g.splitk_ws = nullptr; g.splitk_bytes = 0; }
        }
        prof.begin("gemm.out_proj", 2.0 * Md * E * E);
        pin_kernel(c, g); RC(avx::gemm(g, dt, cs));
        prof.end();
        RC(tap_finish(tap_o, c, w, io, cs));
        if (F == 0) {      // attention-only block: LN1 closes it
            const bool last_a = i == L - 1;
            float* xa = last_a ? (io.features_out ? io.features_out + io.c0 * Tt * E : x32) : (fast ? nullptr : x32);
            prof.begin("layernorm", 0.0);
            RC(avx::layernorm(pre32, preh, E, ly.ln1_w, ly.ln1_b, c.eps, M, E, xa, E, last_a ? nullptr : w.xh, E, dt, cs));
            prof.end();
            if (last_a) {
                io.final_f32 = xa;
                if (io.pooled_out) {
                    prof.begin("mean_pool", 0.0);
                    RC(avx::mean_pool(xa, Bc, Tt, E, nullptr, io.pooled_out + io.c0 * E, cs));
                    prof.end();
                }
            }
            continue;
        }
        if (fold) {
            // (w.r1 = LN1's (rstd, -mu rstd) came out of the product: GemmArgs::rows_out)
        } else if (!ln1_fused) {
            prof.begin("layernorm", 0.0);
            RC(avx::layernorm(pre32, preh, E, ly.ln1_w, ly.ln1_b, c.eps, M, E, fast ? nullptr : x32, E, w.xh, E, dt, cs));
            prof.end();
        }
        memset(&g, 0, sizeof(g)); g.ovf = h->d_ovf;
        g.A = w.xh; g.lda = E; g.W = ly.w_fc1; g.ldw = E; g.M = M; g.N = F; g.K = E; g.bias = ly.b_fc1; g.gelu = c.act;
        g.out_half = w.hh; g.ldh = F;
        if (fold) { g.A = preh; g.W = ly.w_fc1_f; g.bias = ly.b_fc1_f; g.ln_rows = w.r1; g.ln_s = ly.s_fc1; }
        RC(ffn_hidden(h, c, w, g, M, prof, cs));
        const bool last = i == L - 1;
        memset(&g, 0, sizeof(g)); g.ovf = h->d_ovf;
        g.A = w.hh; g.lda = F; g.W = ly.w_fc2; g.ldw = F; g.M = M; g.N = E; g.K = F; g.bias = ly.b_fc2; g.alpha = c.alpha;
        if (M <= 8192 && !c.batch_invariant) { g.splitk_ws = w.splitk; g.splitk_bytes = w.splitk_bytes; }      // (split-K adds in another order than one pass over K)
        if (fast) { g.resid_half = w.xh; g.ldrh = E; g.out_half = preh; g.ldh = E; }
        else { g.resid = x32; g.ldr = E; g.out_f32 = pre32; g.ldo = E; }
        if (fold) {   // residual = LN1(y1) on the fly; y2 (raw) goes to xh, which nothing reads any more in this layer
            g.resid_half = nullptr; g.ldrh = 0;
            g.lnr_y = preh; g.ldy = E; g.lnr_rows = w.r1; g.lnr_gamma = ly.ga_fc2; g.lnr_beta = ly.bb_fc2; g.lnr_prefolded = 1;
            g.out_half = w.xh; g.stats_out = !last ? w.st2 : nullptr;      // the last layer's y2 goes to a LayerNorm kernel that takes its own statistics
        }
        Tap tap_f;
        if (c.hook_site == 0) tap_f = tap_begin(c, w, io, i, g);
        // the last LayerNorm produces the fp32 features (caller's buffer, or scratch when only pooling)
        float* xo = nullptr;
        if (last) xo = io.features_out ? io.features_out + io.c0 * Tt * E : ((io.pooled_out || !fast) ? x32 : nullptr);
        else if (!fast) xo = x32;
        // pooled embedding only (the headline path): final LayerNorm and the mean over tokens in one pass, no fp32 feature tensor
        const bool fused_pool = last && io.pooled_out && !io.features_out && preh && !pre32 && E % 8 == 0 && E <= 768 && (Bc >= 32 || c.batch_invariant);
        // few rows, no fold: LayerNorm 2 in the split-K epilogue (as LayerNorm 1 above)
        bool ln2_fused = false;
        if (!fold && !fused_pool && (xo || !last) && avx::gemm_post_ln_ok(g)) {
            g.post_ln_w = ly.ln2_w; g.post_ln_b = ly.ln2_b; g.post_ln_eps = c.eps; g.post_ln_round = fast ? 1 : 0;
            g.post_ln_out_f32 = xo; g.post_ln_ldo = E; g.post_ln_out_half = last ? nullptr : w.xh; g.post_ln_ldh = E;
            ln2_fused = true;
        }
        prof.begin("gemm.fc2", 2.0 * Md * E * F);
        pin_kernel(c, g); RC(avx::gemm(g, dt, cs));
        prof.end();
        RC(tap_finish(tap_f, c, w, io, cs));
        if (fold && !last) {
            prof.begin("ln_rowstats", 0.0);
            RC(avx::ln_rowstats(w.st2, M, nseg, c.eps, w.r2, cs));
            prof.end();
        }
        if (fused_pool) {      // the pre-LayerNorm sums y2 sit in preh, with the fold in xh
            prof.begin("layernorm+mean_pool", 0.0);
            RC(avx::layernorm_pool(fold ? w.xh : preh, E, ly.ln2_w, ly.ln2_b, c.eps, Bc, Tt, E, io.pooled_out + io.c0 * E, dt, cs));
            prof.end();
        } else if (!fold) {
            prof.begin("layernorm", 0.0);
            if ((xo || !last) && !ln2_fused) RC(avx::layernorm(pre32, preh, E, ly.ln2_w, ly.ln2_b, c.eps, M, E, xo, E, last ? nullptr : w.xh, E, dt, cs));
            prof.end();
        } else if (last && xo) {   // the only LayerNorm of the layer stack that still runs: fp32 features from the raw y2
            prof.begin("layernorm", 0.0);
            RC(avx::layernorm(nullptr, w.xh, E, ly.ln2_w, ly.ln2_b, c.eps, M, E, xo, E, nullptr, E, dt, cs));
            prof.end();
        }
        if (last) io.final_f32 = xo;
        if (last && io.pooled_out && !fused_pool) {
            prof.begin("mean_pool", 0.0);
            RC(avx::mean_pool(xo, Bc, Tt, E, nullptr, io.pooled_out + io.c0 * E, cs));
            prof.end();
        }
    }
#undef RC
    return AVEXHIP_OK;
}

}  // namespace avxh
