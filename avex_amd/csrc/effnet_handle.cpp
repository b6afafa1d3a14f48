// EfficientNet-B0 / B1 encoder handle (BASELINE config C5's model): torchvision efficientnet_b0().features as the reference calls it
// (avex/models/efficientnet.py:55-66,116-137,208) on a mel image, with the BEATs handle's contract -- weights copied at creation from
// a name -> fp32 table with torchvision's keys, caller-owned workspace, chunking, hook taps, no hidden synchronisation.
//
// Layout and arithmetic are those of round 2's Python composition (avex_amd/effnet_encoder.py, now a thin wrapper over this file):
//  * activations NHWC in the operand type, channels padded to multiples of 128 (64 for the stem and the depthwise layer behind it);
//    padding channels are exactly zero in every layer (zero weights, zero bias, SiLU(0) = 0);
//  * every 1x1 convolution is avexhip_gemm over [B*H*W, Cp] rows with eval-mode BatchNorm folded into weight and bias, SiLU (gelu = 2)
//    or the residual add in the epilogue;
//  * the stem sums its weights over the three identical input channels (the reference repeats the mel image, efficientnet.py:138-140);
//  * depthwise k x k (+ folded BN + SiLU + the squeeze-excitation pool), the two SE fully connected layers and the channel rescale are
//    the bandwidth-bound kernels of effnet.hip;
//  * hook taps are the convolutions' outputs BEFORE their BatchNorm (efficientnet.py:82-114), NCHW fp32 like the reference's.
// PARITY UNPINNED: torchvision is absent from the reference tree and both machines; checker oracle/effnet_oracle.py.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <functional>
#include <vector>

#include "common.h"
#include "handle_core.h"

using avxh::align_up;
using avxh::Prof;
using avxh::Table;

namespace {

inline int pad128(int c) { return ((c + 127) / 128) * 128; }
inline bool skinny_enabled() { const char* e = getenv("AVEX_AMD_GEMM_SKINNY"); return !(e && atoi(e) == 0) && !getenv("AVEX_AMD_GEMM_VARIANT"); }
// channels of an activation tensor in memory: 64 for the narrow block outputs of the first stages (at the largest spatial sizes: padded
// to 128 they were 3 - 8x their size), else a multiple of 128; block outputs of <= 32 channels are 32 wide where every consumer takes
// K = 32 (effnet_build: narrow_ok).  GEMM outputs narrower than the 128-column tile
// are computed at pad128 and stored through GemmArgs::n_store.
inline int padc(int c) { return c <= 64 ? 64 : pad128(c); }
// ... and of an EXPANDED tensor: 96 and 144 (-> 160) channels stay that narrow when the skinny streaming kernel is there to take K, N = 96 / 160
// (padded to 128 / 256 the 144-channel tensors were 44 % padding)
inline int padx(int c) {
    const int p32 = ((c + 31) / 32) * 32;
    return (skinny_enabled() && (p32 == 96 || p32 == 160)) ? p32 : padc(c);
}
inline bool skinny_dim(int d) { return d == 32 || d == 64 || d == 96 || d == 128 || d == 160 || d == 256; }

struct Block {
    int k = 3, stride = 1, cin = 0, cexp = 0, cout = 0, cs = 0;     // cs: squeeze width
    int cp_in = 0, cp_exp = 0, cp_out = 0;                          // padded channel counts of the block's input / expanded / output activations
    bool has_expand = false, tap = false, residual = false;
    void* w_exp = nullptr; float* b_exp = nullptr;
    float* w_dw = nullptr; float* b_dw = nullptr;
    float* se_w1 = nullptr; float* se_b1 = nullptr; float* se_w2 = nullptr; float* se_b2 = nullptr;      // se_w2: [cs][cexp] (transposed)
    void* w_proj = nullptr; float* b_proj = nullptr;
    float* proj_scale = nullptr; float* proj_shift = nullptr;       // the folded BatchNorm of the projection, to undo it in a tap
};

// The front of a block (expansion + depthwise) in one kernel (avx::mbconv_front) where the block's input is narrow enough to expand on the
// fly: K = 32 or 64 input channels (EfficientNet-B0: the five blocks at 64 x 501 ... 16 x 126, whose expanded tensors are the largest of
// the network).  Returns -1 (unfused), 32 or 64.  AVEX_AMD_MBCONV=0: never.
inline int fused_kin(const Block& b) {
    const char* e = getenv("AVEX_AMD_MBCONV");      // read per call: tests flip it inside one process
    const bool off = e && atoi(e) == 0;
    if (off || b.cp_exp % 32 != 0) return -1;
    if (!b.has_expand) return -1;
    if (b.cin <= 32 && b.cp_in >= 32) return 32;
    if (b.cin <= 64 && b.cp_in >= 64) return 64;
    return -1;
}

// the other blocks' depthwise convolution: through LDS (avx::dwconv_lds_parts) or the register-tile kernel (AVEX_AMD_DW_LDS=0)
// Measured per 256 clips of EfficientNet-B0 (profiles/r04s_effnet_layers.txt): 32 channels at 64 x 501 414 -> 351 us, the 8 x 63 maps
// 97 -> 85 (3 x 3), 144 -> 120 and 205 -> 182 us (5 x 5); the 4 x 32 maps and the stride-2 step onto them lose (74 -> 120 us: half of an
// 8-row tile is outside the map and a chunk is too little work for its barrier), so those stay with the register-tile kernel.
inline bool dw_lds(const Block& b, int H) {
    const char* e = getenv("AVEX_AMD_DW_LDS");
    if (e && atoi(e) == 2) return b.cp_exp % 32 == 0;      // every block (tests)
    return !(e && atoi(e) == 0) && b.cp_exp % 32 == 0 && b.stride == 1 && H >= 8;
}
}  // namespace

struct avexhip_effnet : avxh::HandleBase {
    avexhip_effnet_config cfg;
    int chunk = 256;
    int c0 = 32, cp0 = 64, head = 1280, cp_last = 0;
    float* w_stem = nullptr; float* b_stem = nullptr; float* stem_scale = nullptr; float* stem_shift = nullptr;
    std::vector<Block> blocks;
    void* w_head = nullptr; float* b_head = nullptr; float* head_scale = nullptr; float* head_shift = nullptr;
    int n_taps = 0;
};

namespace {

int upload_f32(avexhip_effnet* h, const std::vector<float>& v, float** out) {
    float* d = nullptr;
    AVX_HIP_CHECK(hipMalloc((void**)&d, sizeof(float) * v.size()));
    h->allocs.push_back(d);
    AVX_HIP_CHECK(hipMemcpy(d, v.data(), sizeof(float) * v.size(), hipMemcpyHostToDevice));
    *out = d;
    return AVEXHIP_OK;
}

// eval-mode BatchNorm as an affine map: y = x * scale + shift
int bn_fold(avexhip_effnet* h, const Table& tb, const std::string& name, int C, std::vector<float>& scale, std::vector<float>& shift) {
    std::vector<float> w, b, m, v;
    int rc;
    if ((rc = avxh::host_f32(h, tb, name + ".weight", C, w)) != AVEXHIP_OK || (rc = avxh::host_f32(h, tb, name + ".bias", C, b)) != AVEXHIP_OK ||
        (rc = avxh::host_f32(h, tb, name + ".running_mean", C, m)) != AVEXHIP_OK || (rc = avxh::host_f32(h, tb, name + ".running_var", C, v)) != AVEXHIP_OK)
        return rc;
    scale.resize(C); shift.resize(C);
    for (int c = 0; c < C; ++c) {
        scale[c] = w[c] / sqrtf(v[c] + h->cfg.bn_eps);
        shift[c] = b[c] - m[c] * scale[c];
    }
    return AVEXHIP_OK;
}

// 1x1 convolution [N, K, 1, 1] + BatchNorm -> half [Np, Kp] weight with the BN scale folded in, fp32 [Np] bias = BN shift
int pointwise(avexhip_effnet* h, const Table& tb, const std::string& conv, const std::string& bn, int N, int K, int Kp, void** w_out, float** b_out,
              std::vector<float>* scale_out, std::vector<float>* shift_out) {
    std::vector<float> w, sc, sh;
    int rc;
    if ((rc = avxh::host_f32(h, tb, conv + ".weight", (int64_t)N * K, w)) != AVEXHIP_OK) return rc;
    if ((rc = bn_fold(h, tb, bn, N, sc, sh)) != AVEXHIP_OK) return rc;
    const int Np = pad128(N);
    std::vector<float> wp((size_t)Np * Kp, 0.f), bp(Np, 0.f);
    for (int n = 0; n < N; ++n) {
        for (int k = 0; k < K; ++k) wp[(size_t)n * Kp + k] = w[(size_t)n * K + k] * sc[n];
        bp[n] = sh[n];
    }
    AVX_HIP_CHECK(hipMalloc(w_out, 2 * wp.size()));
    h->allocs.push_back(*w_out);
    if ((rc = avxh::upload_half(h, wp.data(), (int64_t)wp.size(), *w_out, conv.c_str())) != AVEXHIP_OK) return rc;
    if ((rc = upload_f32(h, bp, b_out)) != AVEXHIP_OK) return rc;
    if (scale_out) { *scale_out = sc; *shift_out = sh; }
    return AVEXHIP_OK;
}

int effnet_build(avexhip_effnet* h, const avexhip_tensor* tensors, int n) {
    const avexhip_effnet_config& c = h->cfg;
    Table tb{tensors, n};
    tb.strip1 = "model."; tb.strip2 = nullptr;
    int rc;
#define RC(x) do { rc = (x); if (rc != AVEXHIP_OK) return rc; } while (0)
    // May the tensor that block `idx` reads (the stem's output for idx = 0, else block idx - 1's) be 32 channels wide in memory?  Every
    // consumer must take K = 32: the fused block front and the skinny GEMM do, the 128-tile kernels (K % 64) do not.  The head reads through
    // the 128-tile kernel; a block without an expansion feeds its 32 channels to its projection, which is skinny only up to 256 columns.
    // A block with a residual connection adds its INPUT rows to its output rows: the two tensors must have the same width, so its input may be
    // narrow only if its output may be (found by tests/tools/fuzz_effnet.py: a 32-wide input added into a 64-wide output read every row's
    // columns 32 .. 63 from the next row, and the last row's from stale memory).
    struct Bl { bool ex, res; int cexp, cout; };
    std::vector<Bl> plan;
    for (int si = 0; si < c.n_stages; ++si)
        for (int j = 0; j < c.stage[si][5]; ++j) {
            const int ci = j == 0 ? c.stage[si][3] : c.stage[si][4];
            const int st = j == 0 ? c.stage[si][2] : 1;
            plan.push_back({c.stage[si][0] != 1, st == 1 && ci == c.stage[si][4], ci * c.stage[si][0], c.stage[si][4]});
        }
    std::function<bool(size_t, int)> narrow_ok = [&](size_t idx, int ch) -> bool {
        if (ch > 32 || !skinny_enabled() || idx >= plan.size()) return false;
        const Bl& n = plan[idx];
        if (n.res && !narrow_ok(idx + 1, n.cout)) return false;
        if (n.ex) { const int ce = padx(n.cexp); return skinny_dim(ce) && ce * 32 <= 32768; }
        return pad128(n.cout) <= 256;
    };
    h->cp0 = narrow_ok(0, h->c0) ? 32 : ((h->c0 + 63) / 64) * 64;
    {   // stem: Conv2d(3, c0, 3, s2) on three copies of one image = a 1-channel 3x3 convolution with channel-summed weights; [9, cp0]
        std::vector<float> w, sc, sh;
        RC(avxh::host_f32(h, tb, "features.0.0.weight", (int64_t)h->c0 * 3 * 9, w));
        RC(bn_fold(h, tb, "features.0.1", h->c0, sc, sh));
        std::vector<float> ws((size_t)9 * h->cp0, 0.f), bs(h->cp0, 0.f);
        for (int o = 0; o < h->c0; ++o) {
            for (int t = 0; t < 9; ++t) {
                const float sum = w[((size_t)o * 3 + 0) * 9 + t] + w[((size_t)o * 3 + 1) * 9 + t] + w[((size_t)o * 3 + 2) * 9 + t];
                ws[(size_t)t * h->cp0 + o] = sum * sc[o];
            }
            bs[o] = sh[o];
        }
        RC(upload_f32(h, ws, &h->w_stem)); RC(upload_f32(h, bs, &h->b_stem));
        RC(upload_f32(h, sc, &h->stem_scale)); RC(upload_f32(h, sh, &h->stem_shift));
    }
    int cp = h->cp0;
    h->n_taps = 1;
    size_t bi = 0;                                         // index of the block being built in `plan`
    for (int si = 0; si < c.n_stages; ++si) {
        const int er = c.stage[si][0], k = c.stage[si][1], s = c.stage[si][2], cin = c.stage[si][3], cout = c.stage[si][4], reps = c.stage[si][5];
        for (int j = 0; j < reps; ++j) {
            Block b;
            b.k = k; b.stride = j == 0 ? s : 1; b.cin = j == 0 ? cin : cout; b.cexp = b.cin * er; b.cout = cout;
            b.has_expand = er != 1; b.tap = b.has_expand; b.residual = b.stride == 1 && b.cin == b.cout;
            b.cp_in = cp;
            const std::string p = "features." + std::to_string(si + 1) + "." + std::to_string(j) + ".block.";
            const int d = b.has_expand ? 1 : 0;
            if (b.has_expand) {
                RC(pointwise(h, tb, p + "0.0", p + "0.1", b.cexp, b.cin, cp, &b.w_exp, &b.b_exp, nullptr, nullptr));
                cp = padx(b.cexp);
            }
            b.cp_exp = cp;
            {   // depthwise k x k + BN: [k*k, cp]
                std::vector<float> w, sc, sh;
                RC(avxh::host_f32(h, tb, p + std::to_string(d) + ".0.weight", (int64_t)b.cexp * k * k, w));
                RC(bn_fold(h, tb, p + std::to_string(d) + ".1", b.cexp, sc, sh));
                std::vector<float> wd((size_t)k * k * cp, 0.f), bd(cp, 0.f);
                for (int ch = 0; ch < b.cexp; ++ch) {
                    for (int t = 0; t < k * k; ++t) wd[(size_t)t * cp + ch] = w[(size_t)ch * k * k + t] * sc[ch];
                    bd[ch] = sh[ch];
                }
                RC(upload_f32(h, wd, &b.w_dw)); RC(upload_f32(h, bd, &b.b_dw));
            }
            {   // squeeze-excitation: fc1 [cs, cexp, 1, 1], fc2 [cexp, cs, 1, 1]
                const std::string se = p + std::to_string(d + 1) + ".";
                const avexhip_tensor* t1 = tb.find(se + "fc1.bias");
                if (!t1 || t1->numel <= 0) { avexhip_set_error("effnet_create: tensor '%sfc1.bias' missing", se.c_str()); return AVEXHIP_ERR_MISSING; }
                b.cs = (int)t1->numel;
                RC(avxh::dev_f32(h, tb, se + "fc1.weight", (int64_t)b.cs * b.cexp, &b.se_w1)); RC(avxh::dev_f32(h, tb, se + "fc1.bias", b.cs, &b.se_b1));
                {   // second layer transposed [cs][cexp]: coalesced in avx::se_from_parts
                    std::vector<float> w2, w2t((size_t)b.cs * b.cexp);
                    RC(avxh::host_f32(h, tb, se + "fc2.weight", (int64_t)b.cexp * b.cs, w2));
                    for (int ch = 0; ch < b.cexp; ++ch)
                        for (int j = 0; j < b.cs; ++j) w2t[(size_t)j * b.cexp + ch] = w2[(size_t)ch * b.cs + j];
                    RC(upload_f32(h, w2t, &b.se_w2));
                }
                RC(avxh::dev_f32(h, tb, se + "fc2.bias", b.cexp, &b.se_b2));
            }
            std::vector<float> scp, shp;
            RC(pointwise(h, tb, p + std::to_string(d + 2) + ".0", p + std::to_string(d + 2) + ".1", b.cout, b.cexp, cp, &b.w_proj, &b.b_proj, &scp, &shp));
            RC(upload_f32(h, scp, &b.proj_scale)); RC(upload_f32(h, shp, &b.proj_shift));
            cp = narrow_ok(bi + 1, b.cout) ? 32 : padc(b.cout);
            b.cp_out = cp;
            ++bi;
            if (b.tap) ++h->n_taps;
            h->blocks.push_back(b);
        }
    }
    h->cp_last = cp;
    {
        const std::string hn = "features." + std::to_string(c.n_stages + 1);
        std::vector<float> sc, sh;
        RC(pointwise(h, tb, hn + ".0", hn + ".1", h->head, h->blocks.back().cout, cp, &h->w_head, &h->b_head, &sc, &sh));
        RC(upload_f32(h, sc, &h->head_scale)); RC(upload_f32(h, sh, &h->head_shift));
        ++h->n_taps;
    }
#undef RC
    AVX_REQUIRE(h->n_taps <= 32, "effnet_create: %d hookable layers (at most 32 fit the hook mask)", h->n_taps);
    AVX_HIP_CHECK(hipDeviceSynchronize());
    return AVEXHIP_OK;
}

inline int conv_out(int n, int k, int s) { const int pad = (k - 1) / 2; return (n + 2 * pad - k) / s + 1; }

struct EffWs {
    char* act[4]; float* raw; float* pool; float* scale; float* headf; float* part;
    size_t total, part_bytes;
};

// worst-case buffers for a chunk of Bc clips of H x W: four activation buffers (block input / expanded / depthwise output / block output
// rotate through them), one fp32 raw buffer for a tap, SE vectors, the head's fp32 output
EffWs eff_carve(const avexhip_effnet* h, char* base, int Bc, int H, int W) {
    size_t max_act = 0, max_raw = 0;
    int hh = (H - 1) / 2 + 1, ww = (W - 1) / 2 + 1;
    max_act = (size_t)hh * ww * h->cp0 * 2;
    max_raw = (size_t)hh * ww * h->cp0 * 4;
    int cpmax = h->cp0;
    size_t max_part = 0;                                   // the depthwise kernel's rows of partial squeeze sums
    for (const Block& b : h->blocks) {
        if ((size_t)hh * ww * b.cp_exp * 2 > max_act) max_act = (size_t)hh * ww * b.cp_exp * 2;
        const int kin = fused_kin(b);
        size_t part = kin >= 0 ? sizeof(float) * (size_t)Bc * (size_t)avx::mbconv_front_tiles(hh, ww, b.k, b.stride, kin) * b.cp_exp
                               : avexhip_effnet_dwconv_part_bytes(Bc, hh, ww, b.cp_exp, b.k, b.stride);
        if (kin < 0 && dw_lds(b, hh)) {
            const size_t pl = sizeof(float) * (size_t)Bc * (size_t)avx::dwconv_lds_tiles(hh, ww, b.k, b.stride) * b.cp_exp;
            if (pl > part) part = pl;
        }
        if (part > max_part) max_part = part;
        hh = conv_out(hh, b.k, b.stride); ww = conv_out(ww, b.k, b.stride);
        if ((size_t)hh * ww * b.cp_exp * 2 > max_act) max_act = (size_t)hh * ww * b.cp_exp * 2;
        if ((size_t)hh * ww * b.cp_out * 2 > max_act) max_act = (size_t)hh * ww * b.cp_out * 2;      // (the first block widens: 64 padded channels in, 128 out)
        if ((size_t)hh * ww * b.cp_out * 4 > max_raw) max_raw = (size_t)hh * ww * b.cp_out * 4;
        if (b.cp_exp > cpmax) cpmax = b.cp_exp;
    }
    const size_t headf = (size_t)hh * ww * pad128(h->head) * 4;
    if (headf > max_raw) max_raw = headf;
    EffWs w;
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += align_up(bytes); return p; };
    for (int i = 0; i < 4; ++i) w.act[i] = take((size_t)Bc * max_act);
    w.raw = (float*)take((size_t)Bc * max_raw);
    w.pool = (float*)take((size_t)Bc * cpmax * 4);
    w.scale = (float*)take((size_t)Bc * cpmax * 4);
    w.headf = (float*)take((size_t)Bc * headf);
    w.part = (float*)take(max_part);
    w.part_bytes = max_part;
    w.total = off;
    return w;
}

}  // namespace

extern "C" avexhip_effnet* avexhip_effnet_create(const avexhip_effnet_config* cfg, const avexhip_tensor* tensors, int n_tensors) {
    if (!cfg || !tensors || n_tensors <= 0) { avexhip_set_error("effnet_create: null config or empty weight table"); return nullptr; }
    if (avexhip_device_count() <= 0) { avexhip_set_error("effnet_create: no HIP device visible (this path has no CPU fallback)"); return nullptr; }
    const avexhip_effnet_config& c = *cfg;
    if (c.n_stages < 1 || c.n_stages > 8 || c.stem_channels <= 0 || c.stem_channels > 64 || c.head_channels <= 0) {
        avexhip_set_error("effnet_create: bad layout (stages %d, stem %d, head %d)", c.n_stages, c.stem_channels, c.head_channels);
        return nullptr;
    }
    for (int i = 0; i < c.n_stages; ++i) {
        const int k = c.stage[i][1], s = c.stage[i][2];
        if ((k != 3 && k != 5) || (s != 1 && s != 2) || c.stage[i][0] < 1 || c.stage[i][3] <= 0 || c.stage[i][4] <= 0 || c.stage[i][5] <= 0) {
            avexhip_set_error("effnet_create: stage %d (expand %d, k %d, s %d, %d -> %d, x%d) is outside what the depthwise kernel builds (k 3 | 5, s 1 | 2)", i,
                              c.stage[i][0], k, s, c.stage[i][3], c.stage[i][4], c.stage[i][5]);
            return nullptr;
        }
    }
    if (c.operand_dtype != AVEXHIP_F16 && c.operand_dtype != AVEXHIP_BF16) { avexhip_set_error("effnet_create: unknown operand dtype %d", c.operand_dtype); return nullptr; }
    avexhip_effnet* h = new avexhip_effnet();
    h->who = "effnet_create";
    h->cfg = c;
    if (!(h->cfg.bn_eps > 0.f)) h->cfg.bn_eps = 1e-5f;
    h->dtype = c.operand_dtype;
    h->c0 = c.stem_channels; h->cp0 = ((c.stem_channels + 63) / 64) * 64; h->head = c.head_channels;      // (effnet_build narrows cp0 to 32 for B0 / B1: the stem's output and the first depthwise run at the real width)
    h->chunk = c.max_chunk_clips > 0 ? c.max_chunk_clips : 256;
    if (h->init_alarm() != AVEXHIP_OK || effnet_build(h, tensors, n_tensors) != AVEXHIP_OK || h->weights_fit() != AVEXHIP_OK) {      // (BN-folded weights that leave the f16 range: refused here, like the other four create paths)
    delete h; return nullptr; }
    return h;
}

extern "C" void avexhip_effnet_destroy(avexhip_effnet* h) { delete h; }
extern "C" int avexhip_effnet_num_taps(const avexhip_effnet* h) { return h ? h->n_taps : 0; }

// channels and spatial size of tap `tap` (0 = stem conv, 1.. = the projection convs of the blocks with an expansion, last = head conv)
// for an H x W input image; tap == -1: the feature map (head channels)
extern "C" int avexhip_effnet_tap_shape(const avexhip_effnet* h, int tap, int H, int W, int* C, int* Ho, int* Wo) {
    AVX_REQUIRE(h && C && Ho && Wo && H > 0 && W > 0, "effnet_tap_shape: bad arguments");
    int hh = (H - 1) / 2 + 1, ww = (W - 1) / 2 + 1, t = 0;
    if (tap == 0) { *C = h->c0; *Ho = hh; *Wo = ww; return AVEXHIP_OK; }
    for (const Block& b : h->blocks) {
        hh = conv_out(hh, b.k, b.stride); ww = conv_out(ww, b.k, b.stride);
        if (b.tap && ++t == tap) { *C = b.cout; *Ho = hh; *Wo = ww; return AVEXHIP_OK; }
    }
    if (tap == -1 || tap == t + 1) { *C = h->head; *Ho = hh; *Wo = ww; return AVEXHIP_OK; }
    avexhip_set_error("effnet_tap_shape: tap %d out of range (0..%d)", tap, h->n_taps - 1);
    return AVEXHIP_ERR_INVALID;
}

extern "C" size_t avexhip_effnet_workspace_bytes(const avexhip_effnet* h, int B, int H, int W) {
    if (!h || B <= 0 || H <= 0 || W <= 0) return 0;
    return eff_carve(h, nullptr, B < h->chunk ? B : h->chunk, H, W).total;
}

extern "C" int avexhip_effnet_forward(avexhip_effnet* h, const float* mel, int B, int H, int W, uint32_t hook_mask, float* const* hook_out,
                                      float* features_out, float* pooled_out, void* workspace, size_t ws_bytes, void* stream) {
    AVX_REQUIRE(h && mel, "effnet_forward: null handle or input");
    AVX_REQUIRE(B > 0 && H >= 32 && W >= 32, "effnet_forward: image %d x %d x %d (at least 32 x 32: five stride-2 stages)", B, H, W);
    AVX_REQUIRE(hook_mask == 0 || hook_out, "effnet_forward: hook_mask set but hook_out is NULL");
    AVX_REQUIRE(h->n_taps >= 32 || (hook_mask >> h->n_taps) == 0, "effnet_forward: hook_mask has bits beyond tap %d", h->n_taps - 1);
    for (int i = 0; i < h->n_taps; ++i) AVX_REQUIRE(!((hook_mask >> i) & 1u) || hook_out[i], "effnet_forward: tap %d selected but hook_out[%d] is NULL", i, i);
    hipStream_t s = (hipStream_t)stream;
    const int dt = h->dtype;
    const int chunk = B < h->chunk ? B : h->chunk;
    const EffWs need = eff_carve(h, nullptr, chunk, H, W);
    if (!workspace || ws_bytes < need.total) {
        avexhip_set_error("effnet_forward: workspace too small (%zu bytes given, %zu needed)", ws_bytes, need.total);
        return AVEXHIP_ERR_WORKSPACE;
    }
    Prof prof{h, s};
    int rc;
#define RC(x) do { rc = (x); if (rc != AVEXHIP_OK) return rc; } while (0)
    for (int c0 = 0; c0 < B; c0 += chunk) {
        const int Bc = (B - c0) < chunk ? (B - c0) : chunk;
        const EffWs w = eff_carve(h, (char*)workspace, chunk, H, W);
        int hh = (H - 1) / 2 + 1, ww = (W - 1) / 2 + 1;
        int cur = 0;                                           // index of the activation buffer that holds the current tensor
        // stem (+ its tap: the kernel's raw output is the BatchNorm output before SiLU)
        const bool tap0 = hook_mask & 1u;
        prof.begin("stem", 2.0 * Bc * hh * ww * (double)h->c0 * 9);
        RC(avexhip_effnet_stem(mel + (size_t)c0 * H * W, Bc, H, W, h->w_stem, h->b_stem, h->cp0, w.act[cur], tap0 ? w.raw : nullptr, dt, s));
        prof.end();
        if (tap0) RC(avx::nhwc_to_nchw(w.raw, h->cp0, Bc, hh * ww, h->c0, h->stem_scale, h->stem_shift, hook_out[0] + (size_t)c0 * h->c0 * hh * ww, s));
        int tap = 0;
        for (const Block& b : h->blocks) {
            const int in_buf = cur;
            const int M_in = Bc * hh * ww;
            int x = in_buf;
            avx::GemmArgs g;
            const int kin = fused_kin(b);
            if (b.has_expand && kin < 0) {
                const int o = (in_buf + 1) & 3;
                memset(&g, 0, sizeof(g)); g.ovf = h->d_ovf;
                g.A = w.act[in_buf]; g.lda = b.cp_in; g.W = b.w_exp; g.ldw = b.cp_in; g.M = M_in; g.N = pad128(b.cexp); g.K = b.cp_in; g.bias = b.b_exp; g.gelu = 2;
                g.out_half = w.act[o]; g.ldh = b.cp_exp; g.n_store = b.cp_exp < g.N ? b.cp_exp : 0;
                if (skinny_enabled() && skinny_dim(b.cp_in) && skinny_dim(b.cp_exp) && b.cp_exp != 32 && b.cp_exp * b.cp_in <= 32768) {
                    g.N = b.cp_exp; g.n_store = 0; g.variant = 7;      // whatever the row count: the 128-tile kernels take neither K = 32 nor 96 / 160 columns
                }
                prof.begin("gemm.expand", 2.0 * M_in * (double)b.cexp * b.cin);
                RC(avx::gemm(g, dt, s));
                prof.end();
                x = o;
            }
            const int dw = (in_buf + 2) & 3;
            const int h2 = conv_out(hh, b.k, b.stride), w2 = conv_out(ww, b.k, b.stride);
            int64_t part_rows = 0;                       // rows of squeeze partials per clip the depthwise kernel left
            if (kin >= 0) {
                prof.begin("mbconv.front", 2.0 * Bc * h2 * w2 * (double)b.cexp * b.k * b.k + (b.has_expand ? 2.0 * M_in * (double)b.cexp * b.cin : 0.0));
                RC(avx::mbconv_front(w.act[in_buf], Bc, hh, ww, b.cp_in, kin, b.w_exp, b.cp_in, b.b_exp, b.k, b.stride, b.w_dw, b.b_dw, b.cp_exp, w.act[dw],
                                     nullptr, w.part, w.part_bytes, h->d_ovf, dt, s));
                part_rows = avx::mbconv_front_tiles(hh, ww, b.k, b.stride, kin);
                prof.end();
            } else {
                prof.begin("dwconv", 2.0 * Bc * h2 * w2 * (double)b.cexp * b.k * b.k);
                if (dw_lds(b, hh)) RC(avx::dwconv_lds_parts(w.act[x], Bc, hh, ww, b.cp_exp, b.k, b.stride, b.w_dw, b.b_dw, w.act[dw], w.part, w.part_bytes, &part_rows, dt, s));
                else RC(avx::dwconv_parts(w.act[x], Bc, hh, ww, b.cp_exp, b.k, b.stride, b.w_dw, b.b_dw, w.act[dw], w.part, w.part_bytes, &part_rows, dt, s));
                prof.end();
            }
            const int M2 = Bc * h2 * w2;
            const int out = (in_buf + 3) & 3;
            const bool hooked = b.tap && ((hook_mask >> (tap + 1)) & 1u);
            // long thin projections run in the skinny streaming kernel, which applies the squeeze-excitation scale to its A rows as it loads
            // them: the rescale pass over the expanded tensor (read + write) disappears
            static const bool no_se_fold = getenv("AVEX_AMD_SE_FOLD") && atoi(getenv("AVEX_AMD_SE_FOLD")) == 0;
            const bool skinny_proj = skinny_enabled() && skinny_dim(b.cp_exp) && pad128(b.cout) * b.cp_exp <= 32768 && (b.cp_out == 32 || b.cp_out == 64 || b.cp_out == 128 || b.cp_out == 256);
            bool se_fold = skinny_proj && (!no_se_fold || hooked);      // (the skinny kernel's raw tap comes with the scale)
            // the wider projections (K = 512 ... 1152): the register-staged form of the 128-tile kernel scales its A rows the same way
            static const bool se_fold_wide = !(getenv("AVEX_AMD_SE_FOLD_WIDE") && atoi(getenv("AVEX_AMD_SE_FOLD_WIDE")) == 0);
            const bool wide_fold = !skinny_proj && !no_se_fold && se_fold_wide && b.cp_exp % 64 == 0;
            se_fold = se_fold || wide_fold;
            prof.begin("se", 0.0);
            RC(avx::se_from_parts(w.part, part_rows, Bc, (int64_t)h2 * w2, b.cexp, b.cp_exp, b.cs, b.se_w1, b.se_b1, b.se_w2, b.se_b2, w.scale, se_fold ? nullptr : w.act[dw], dt, s));
            prof.end();
            memset(&g, 0, sizeof(g)); g.ovf = h->d_ovf;
            g.A = w.act[dw]; g.lda = b.cp_exp; g.W = b.w_proj; g.ldw = b.cp_exp; g.M = M2; g.N = pad128(b.cout); g.K = b.cp_exp; g.bias = b.b_proj; g.alpha = 1.0f;
            g.n_store = b.cp_out < g.N ? b.cp_out : 0;
            if (skinny_proj) {      // whatever the row count: the 128-tile kernels do not take K = 32 or 64 columns
                g.variant = 7;
                if (b.cp_out < 128) { g.N = b.cp_out; g.n_store = 0; }
                if (se_fold) { g.a_scale = w.scale; g.a_scale_rows = h2 * w2; g.a_scale_ld = b.cp_exp; }
            }
            if (wide_fold) { g.variant = 1; g.a_scale = w.scale; g.a_scale_rows = h2 * w2; g.a_scale_ld = b.cp_exp; }
            if (b.residual) { g.resid_half = w.act[in_buf]; g.ldrh = b.cp_in; }
            g.out_half = w.act[out]; g.ldh = b.cp_out;
            if (hooked) { g.out_raw = w.raw; g.ldraw = b.cp_out; }
            prof.begin("gemm.project", 2.0 * M2 * (double)b.cout * b.cexp);
            RC(avx::gemm(g, dt, s));
            prof.end();
            if (b.tap) {
                ++tap;
                // out_raw = conv * bn_scale + bn_shift (before the residual): the tap is the convolution before its BatchNorm
                if (hooked) RC(avx::nhwc_to_nchw(w.raw, b.cp_out, Bc, h2 * w2, b.cout, b.proj_scale, b.proj_shift, hook_out[tap] + (size_t)c0 * b.cout * h2 * w2, s));
            }
            cur = out; hh = h2; ww = w2;
        }
        // head 1x1 conv + BN + SiLU -> fp32 features
        const int M = Bc * hh * ww, Np = pad128(h->head);
        const bool tap_head = (hook_mask >> (h->n_taps - 1)) & 1u;
        avx::GemmArgs g;
        memset(&g, 0, sizeof(g)); g.ovf = h->d_ovf;
        g.A = w.act[cur]; g.lda = h->cp_last; g.W = h->w_head; g.ldw = h->cp_last; g.M = M; g.N = Np; g.K = h->cp_last; g.bias = h->b_head; g.gelu = 2;
        g.out_f32 = w.headf; g.ldo = Np;
        if (tap_head) { g.out_raw = w.raw; g.ldraw = Np; }
        prof.begin("gemm.head", 2.0 * M * (double)h->head * h->blocks.back().cout);
        RC(avx::gemm(g, dt, s));
        prof.end();
        if (tap_head) RC(avx::nhwc_to_nchw(w.raw, Np, Bc, hh * ww, h->head, h->head_scale, h->head_shift, hook_out[h->n_taps - 1] + (size_t)c0 * h->head * hh * ww, s));
        if (features_out) RC(avx::nhwc_to_nchw(w.headf, Np, Bc, hh * ww, h->head, nullptr, nullptr, features_out + (size_t)c0 * h->head * hh * ww, s));
        if (pooled_out) {
            AVX_REQUIRE(Np == h->head, "effnet_forward: pooled output needs a head width that is a multiple of 128 (%d)", h->head);
            RC(avx::mean_pool(w.headf, Bc, hh * ww, h->head, nullptr, pooled_out + (size_t)c0 * h->head, s));
        }
    }
#undef RC
    { const int rc2 = h->mirror_alarm(s); if (rc2 != AVEXHIP_OK) return rc2; }
    return prof.collect();
}

extern "C" int avexhip_effnet_overflow_count(avexhip_effnet* h, uint32_t* events, void* sync_stream, int synchronize) {
    AVX_REQUIRE(h && events, "effnet_overflow_count: null argument");
    return h->overflow_count(events, (hipStream_t)sync_stream, synchronize);
}
extern "C" int avexhip_effnet_set_profiling(avexhip_effnet* h, int enabled) {
    AVX_REQUIRE(h, "effnet_set_profiling: null handle");
    h->profiling = enabled != 0;
    return AVEXHIP_OK;
}
extern "C" int avexhip_effnet_last_profile(const avexhip_effnet* h, const char* const** names, const float** ms, const double** flops, int* count) {
    AVX_REQUIRE(h && names && ms && flops && count, "effnet_last_profile: null argument");
    *names = h->prof_name_ptrs.data(); *ms = h->prof_ms.data(); *flops = h->prof_flops.data(); *count = (int)h->prof_name_ptrs.size();
    return AVEXHIP_OK;
}
