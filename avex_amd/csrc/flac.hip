// FLAC decode for the audio ingest: what the reference gets from torchaudio.load / soundfile for its
// .flac samples (avex/data/augmentations.py:258-262; tests/samples/animalspeak2/16khz/*/*.flac), written from the format's published
// specification -- no codec library exists on either machine.
//
// FLAC is an integer, frame-parallel codec: every frame decodes on its own, and inside a frame every channel (subframe) is a linear
// predictor run over a residual.  The split follows that:
//   host    the bitstream: metadata blocks, frame headers (CRC-8), subframe headers, the Rice-coded residuals (bit-serial by nature),
//           frame CRC-16.  Every subframe is normalised to one form -- a constant, or an integer predictor of order 0..32
//           (coefficients, shift) over [warm-up samples | residuals]; VERBATIM is order 0, the FIXED predictors are their binomial
//           coefficient rows with shift 0.
//   device  flac_predict_kernel: one lane per subframe runs the recursion s[i] = r[i] + (sum_j c[j] s[i-1-j] >> shift) in 64-bit
//           arithmetic (sequential in time, parallel over frames x channels) and restores wasted bits;
//           flac_interleave_kernel: inter-channel decorrelation (left/side, right/side, mid/side) and the interleaved int32 output,
//           optionally left-justified so that avexhip_pcm_to_mono_f32(format 32) scales it like soundfile does.
// Bit-exactness is checked against the MD5 of the unencoded audio that every FLAC stream carries in STREAMINFO.
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "common.h"

namespace {

struct BitReader {
    const uint8_t* p;
    size_t n, pos = 0;          // pos in bits
    bool bad = false;
    uint32_t bits(int k) {      // k <= 32, MSB first
        uint64_t v = 0;
        for (int got = 0; got < k;) {
            const size_t byte = pos >> 3;
            if (byte >= n) { bad = true; return 0; }
            const int avail = 8 - (int)(pos & 7);
            const int take = (k - got) < avail ? (k - got) : avail;
            v = (v << take) | ((p[byte] >> (avail - take)) & ((1u << take) - 1u));
            pos += take; got += take;
        }
        return (uint32_t)v;
    }
    int32_t sbits(int k) {      // signed two's complement of k bits (k <= 32)
        if (k == 0) return 0;
        const uint32_t v = bits(k);
        const uint32_t sign = 1u << (k - 1);
        return (int32_t)((v ^ sign) - sign);
    }
    int unary() {               // number of 0 bits before the next 1 bit
        int q = 0;
        for (;;) {
            const size_t byte = pos >> 3;
            if (byte >= n) { bad = true; return 0; }
            const int off = (int)(pos & 7);
            const uint32_t rest = (uint32_t)(p[byte] << off) & 0xFFu;         // remaining bits of this byte, left-aligned in 8 bits
            if (rest == 0) { q += 8 - off; pos += 8 - off; continue; }
            const int lead = __builtin_clz(rest) - 24;
            q += lead; pos += lead + 1;
            return q;
        }
    }
    void align() { pos = (pos + 7) & ~(size_t)7; }
};

uint8_t crc8(const uint8_t* d, size_t n) {       // x^8 + x^2 + x + 1, init 0
    uint8_t c = 0;
    for (size_t i = 0; i < n; ++i) {
        c ^= d[i];
        for (int b = 0; b < 8; ++b) c = (uint8_t)((c & 0x80) ? ((c << 1) ^ 0x07) : (c << 1));
    }
    return c;
}
uint16_t crc16(const uint8_t* d, size_t n) {     // x^16 + x^15 + x^2 + 1, init 0
    uint16_t c = 0;
    for (size_t i = 0; i < n; ++i) {
        c ^= (uint16_t)(d[i] << 8);
        for (int b = 0; b < 8; ++b) c = (uint16_t)((c & 0x8000) ? ((c << 1) ^ 0x8005) : (c << 1));
    }
    return c;
}

// one channel of one frame, normalised (see the header)
struct Sub {
    int32_t order;          // -1: constant (value in coef[0]); 0..32: predictor order
    int32_t shift;          // right shift of the prediction (arithmetic)
    int32_t wasted;         // samples are shifted left by this at the end
    int32_t blocksize;
    int32_t coef[32];
    int64_t data_off;       // [blocksize] int32 in the residual buffer: `order` warm-up samples, then residuals
    int64_t out_off;        // first sample of the block within the channel's plane
    int32_t channel, assignment;    // assignment: 0 independent, 8 left/side, 9 right/side, 10 mid/side (the frame header's code)
};

__global__ __launch_bounds__(64) void flac_predict_kernel(const Sub* __restrict__ subs, int n_subs, const int32_t* __restrict__ data, int32_t* __restrict__ planes,
                                                          int64_t plane_stride) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n_subs) return;
    const Sub sb = subs[i];
    int32_t* out = planes + (int64_t)sb.channel * plane_stride + sb.out_off;
    const int32_t* r = data + sb.data_off;
    if (sb.order < 0) {
        const int32_t v = (int32_t)((uint32_t)sb.coef[0] << sb.wasted);
        for (int t = 0; t < sb.blocksize; ++t) out[t] = v;
        return;
    }
    // history of the last `order` UNSHIFTED samples in registers would need dynamic indexing; the plane itself is the history: samples are
    // written unshifted first and shifted in a second pass when the subframe has wasted bits
    for (int t = 0; t < sb.order && t < sb.blocksize; ++t) out[t] = r[t];
    for (int t = sb.order; t < sb.blocksize; ++t) {
        long long acc = 0;
        for (int j = 0; j < sb.order; ++j) acc += (long long)sb.coef[j] * (long long)out[t - 1 - j];
        out[t] = (int32_t)((long long)r[t] + (acc >> sb.shift));
    }
    if (sb.wasted > 0)
        for (int t = 0; t < sb.blocksize; ++t) out[t] = (int32_t)((uint32_t)out[t] << sb.wasted);
}

// planes [channels][total] (side / mid channels still coded) -> interleaved [total][channels]; block_assign[b] / block_start[b]: the
// channel assignment and first sample of frame b
__global__ __launch_bounds__(256) void flac_interleave_kernel(const int32_t* __restrict__ planes, int64_t plane_stride, int channels, int64_t total,
                                                              const int32_t* __restrict__ frame_of_block, int block_shift, const int32_t* __restrict__ frame_assign,
                                                              const int64_t* __restrict__ frame_start, int n_frames, int justify, int32_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    // the frame that holds sample i: a coarse table (one entry per 2^block_shift samples) then a short forward scan
    int f = frame_of_block[i >> block_shift];
    while (f + 1 < n_frames && frame_start[f + 1] <= i) ++f;
    const int assign = frame_assign[f];
    if (channels == 2 && assign >= 8) {
        const int32_t a = planes[i], b = planes[plane_stride + i];
        int32_t l, r;
        if (assign == 8) { l = a; r = a - b; }                       // left / side
        else if (assign == 9) { r = b; l = a + b; }                  // side / right
        else {                                                       // mid / side
            const int32_t mid = (int32_t)(((uint32_t)a << 1) | ((uint32_t)b & 1u));
            l = (mid + b) >> 1; r = (mid - b) >> 1;
        }
        out[2 * i] = (int32_t)((uint32_t)l << justify);
        out[2 * i + 1] = (int32_t)((uint32_t)r << justify);
    } else {
        for (int c = 0; c < channels; ++c) out[i * channels + c] = (int32_t)((uint32_t)planes[(int64_t)c * plane_stride + i] << justify);
    }
}

constexpr int FOB_SHIFT = 8;
const int32_t FIXED_COEF[5][4] = {{0, 0, 0, 0}, {1, 0, 0, 0}, {2, -1, 0, 0}, {3, -3, 1, 0}, {4, -6, 4, -1}};

}  // namespace

struct avexhip_flac {
    int sample_rate = 0, channels = 0, bps = 0;
    int64_t total = 0;
    uint8_t md5[16];
    std::vector<Sub> subs;
    std::vector<int32_t> data;            // per subframe: [blocksize] warm-up samples + residuals
    std::vector<int32_t> frame_assign;
    std::vector<int64_t> frame_start;
    std::vector<int32_t> frame_of_block;  // coarse sample -> frame table: one entry per 2^FOB_SHIFT samples
    std::string error;
};

namespace {

bool parse(avexhip_flac* h, const uint8_t* d, size_t n) {
    auto fail = [&](const std::string& m) { h->error = m; return false; };
    if (n < 42 || memcmp(d, "fLaC", 4) != 0) return fail("not a FLAC stream (no fLaC marker)");
    size_t pos = 4;
    bool have_info = false;
    int min_bs = 0, max_bs = 0;
    for (;;) {
        if (pos + 4 > n) return fail("truncated metadata");
        const int last = d[pos] >> 7, type = d[pos] & 0x7F;
        const size_t len = ((size_t)d[pos + 1] << 16) | ((size_t)d[pos + 2] << 8) | d[pos + 3];
        if (pos + 4 + len > n) return fail("truncated metadata block");
        const uint8_t* b = d + pos + 4;
        if (type == 0) {
            if (len < 34) return fail("short STREAMINFO");
            min_bs = (b[0] << 8) | b[1]; max_bs = (b[2] << 8) | b[3];
            const uint64_t x = ((uint64_t)b[10] << 56) | ((uint64_t)b[11] << 48) | ((uint64_t)b[12] << 40) | ((uint64_t)b[13] << 32) | ((uint64_t)b[14] << 24) |
                               ((uint64_t)b[15] << 16) | ((uint64_t)b[16] << 8) | b[17];
            h->sample_rate = (int)(x >> 44); h->channels = (int)((x >> 41) & 7) + 1; h->bps = (int)((x >> 36) & 31) + 1;
            h->total = (int64_t)(x & ((1ull << 36) - 1));
            memcpy(h->md5, b + 18, 16);
            have_info = true;
        }
        pos += 4 + len;
        if (last) break;
    }
    if (!have_info) return fail("no STREAMINFO block");
    if (h->channels < 1 || h->channels > 8 || h->bps < 4 || h->bps > 32 || h->sample_rate <= 0) return fail("STREAMINFO out of range");
    (void)min_bs; (void)max_bs;
    int64_t done = 0;
    while (pos + 2 <= n) {
        if (!(d[pos] == 0xFF && (d[pos + 1] & 0xFE) == 0xF8)) {
            if (h->total > 0 && done >= h->total) break;     // trailing bytes after the last frame (ID3v1 tags and the like)
            return fail("lost frame synchronisation at byte " + std::to_string(pos));
        }
        BitReader br{d + pos, n - pos};
        br.bits(14); br.bits(1);
        const int variable = (int)br.bits(1);
        const int bs_code = (int)br.bits(4), sr_code = (int)br.bits(4), assign = (int)br.bits(4), ss_code = (int)br.bits(3);
        br.bits(1);
        // UTF-8-style coded frame / sample number
        {
            const uint32_t first = br.bits(8);
            int extra = 0;
            if (first & 0x80) { uint32_t m = 0x40; while (first & m) { ++extra; m >>= 1; } if (extra == 0 || extra > 6) return fail("bad coded number in a frame header"); }
            for (int i = 0; i < extra; ++i) br.bits(8);
        }
        (void)variable;
        int bs;
        if (bs_code == 0) return fail("reserved block size code");
        else if (bs_code == 1) bs = 192;
        else if (bs_code <= 5) bs = 576 << (bs_code - 2);
        else if (bs_code == 6) bs = (int)br.bits(8) + 1;
        else if (bs_code == 7) bs = (int)br.bits(16) + 1;
        else bs = 256 << (bs_code - 8);
        if (sr_code == 12) br.bits(8); else if (sr_code == 13 || sr_code == 14) br.bits(16); else if (sr_code == 15) return fail("invalid sample rate code");
        if (br.bad || (br.pos & 7)) return fail("truncated frame header");
        if (bs < 1) return fail("empty block");
        const size_t hdr_bytes = br.pos >> 3;
        const uint8_t want8 = (uint8_t)br.bits(8);
        if (crc8(d + pos, hdr_bytes) != want8) return fail("frame header CRC-8 mismatch at byte " + std::to_string(pos));
        int bps = h->bps;
        const int ss_table[8] = {0, 8, 12, -1, 16, 20, 24, 32};
        if (ss_code != 0) { if (ss_table[ss_code] < 0) return fail("reserved sample size code"); bps = ss_table[ss_code]; }
        int nch;
        if (assign < 8) nch = assign + 1; else if (assign <= 10) nch = 2; else return fail("reserved channel assignment");
        if (nch != h->channels) return fail("a frame's channel count differs from STREAMINFO");
        h->frame_assign.push_back(assign);
        h->frame_start.push_back(done);
        for (int ch = 0; ch < nch; ++ch) {
            if (br.bits(1) != 0) return fail("subframe padding bit set");
            const int type = (int)br.bits(6);
            int wasted = 0;
            if (br.bits(1)) wasted = br.unary() + 1;
            int sbps = bps - wasted;
            if ((assign == 8 && ch == 1) || (assign == 9 && ch == 0) || (assign == 10 && ch == 1)) sbps += 1;      // the side channel has one more bit
            if (sbps < 1 || sbps > 33) return fail("subframe sample width out of range");
            Sub sb;
            memset(&sb, 0, sizeof(sb));
            sb.blocksize = bs; sb.wasted = wasted; sb.channel = ch; sb.assignment = assign; sb.out_off = done;
            sb.data_off = (int64_t)h->data.size();
            h->data.resize(h->data.size() + (size_t)bs, 0);
            int32_t* dst = h->data.data() + sb.data_off;
            auto sample = [&](int width) -> int32_t {        // a raw sample of `width` bits (33 only for a side channel of a 32-bit stream: not built)
                return br.sbits(width > 32 ? 32 : width);
            };
            if (sbps > 32) return fail("33-bit side channels (32-bit stereo streams) are not built");
            int order = 0;
            if (type == 0) {                                  // CONSTANT
                sb.order = -1; sb.coef[0] = sample(sbps);
            } else if (type == 1) {                           // VERBATIM
                sb.order = 0;
                for (int t = 0; t < bs; ++t) dst[t] = sample(sbps);
            } else if (type >= 8 && type <= 12) {             // FIXED
                order = type - 8;
                if (order > bs) return fail("predictor order exceeds the block size");      // before the warm-up samples are written: dst holds bs values
                sb.order = order; sb.shift = 0;
                for (int j = 0; j < order; ++j) sb.coef[j] = FIXED_COEF[order][j];
                for (int t = 0; t < order; ++t) dst[t] = sample(sbps);
            } else if (type >= 32) {                          // LPC
                order = type - 31;
                if (order > bs) return fail("predictor order exceeds the block size");      // (a one-sample block with an order-32 predictor would write 31 values past dst)
                sb.order = order;
                for (int t = 0; t < order; ++t) dst[t] = sample(sbps);
                const int prec = (int)br.bits(4) + 1;
                if (prec == 16) return fail("invalid LPC precision");
                sb.shift = br.sbits(5);
                if (sb.shift < 0) return fail("negative LPC shift");
                for (int j = 0; j < order; ++j) sb.coef[j] = br.sbits(prec);
            } else {
                return fail("reserved subframe type " + std::to_string(type));
            }
            if (type >= 8) {                                  // residual (FIXED and LPC)
                const int method = (int)br.bits(2);
                if (method > 1) return fail("reserved residual coding method");
                const int pbits = method == 0 ? 4 : 5, esc = method == 0 ? 15 : 31;
                const int porder = (int)br.bits(4);
                const int nparts = 1 << porder;
                if ((bs >> porder) << porder != bs && porder > 0) return fail("block size not divisible by the partition count");
                int t = order;
                for (int part = 0; part < nparts; ++part) {
                    int cnt = porder == 0 ? bs - order : (part == 0 ? (bs >> porder) - order : (bs >> porder));
                    if (cnt < 0) return fail("partition shorter than the predictor order");
                    const int k = (int)br.bits(pbits);
                    if (k == esc) {
                        const int raw = (int)br.bits(5);
                        for (int i = 0; i < cnt; ++i) dst[t++] = br.sbits(raw);
                    } else {
                        for (int i = 0; i < cnt; ++i) {
                            const uint32_t q = (uint32_t)br.unary();
                            const uint32_t u = (q << k) | (k ? br.bits(k) : 0u);
                            dst[t++] = (int32_t)(u >> 1) ^ -(int32_t)(u & 1u);
                        }
                    }
                    if (br.bad) return fail("truncated residual");
                }
            }
            if (br.bad) return fail("truncated subframe");
            h->subs.push_back(sb);
        }
        br.align();
        const size_t body = br.pos >> 3;
        const uint16_t want16 = (uint16_t)br.bits(16);
        if (br.bad) return fail("truncated frame");
        if (crc16(d + pos, body) != want16) return fail("frame CRC-16 mismatch at byte " + std::to_string(pos));
        pos += body + 2;
        done += bs;
    }
    if (h->total == 0) h->total = done;          // (a streamed file may leave the count at zero)
    if (done != h->total) return fail("decoded " + std::to_string(done) + " samples, STREAMINFO says " + std::to_string(h->total));
    const int nfr = (int)h->frame_start.size();
    h->frame_of_block.resize((size_t)((h->total >> FOB_SHIFT) + 1));
    int f = 0;
    for (size_t b = 0; b < h->frame_of_block.size(); ++b) {
        const int64_t i = (int64_t)b << FOB_SHIFT;
        while (f + 1 < nfr && h->frame_start[f + 1] <= i) ++f;
        h->frame_of_block[b] = f;
    }
    return true;
}

}  // namespace

extern "C" avexhip_flac* avexhip_flac_open(const uint8_t* data, size_t n_bytes) {
    if (!data || n_bytes == 0) { avexhip_set_error("flac_open: empty input"); return nullptr; }
    avexhip_flac* h = new avexhip_flac();
    if (!parse(h, data, n_bytes)) {
        avexhip_set_error("flac_open: %s", h->error.c_str());
        delete h;
        return nullptr;
    }
    return h;
}

extern "C" void avexhip_flac_close(avexhip_flac* h) { delete h; }

extern "C" int avexhip_flac_info(const avexhip_flac* h, int* sample_rate, int* channels, int* bits_per_sample, int64_t* total_samples, uint8_t* md5_16) {
    AVX_REQUIRE(h, "flac_info: null handle");
    if (sample_rate) *sample_rate = h->sample_rate;
    if (channels) *channels = h->channels;
    if (bits_per_sample) *bits_per_sample = h->bps;
    if (total_samples) *total_samples = h->total;
    if (md5_16) memcpy(md5_16, h->md5, 16);
    return AVEXHIP_OK;
}

extern "C" int avexhip_flac_decode_i32(const avexhip_flac* h, int32_t* out_dev, int left_justify, void* stream) {
    AVX_REQUIRE(h && out_dev, "flac_decode_i32: null argument");
    AVX_REQUIRE(h->total > 0 && !h->subs.empty(), "flac_decode_i32: empty stream");
    hipStream_t s = (hipStream_t)stream;
    const int64_t total = h->total;
    const int nsub = (int)h->subs.size(), nfr = (int)h->frame_start.size();
    const int shift = FOB_SHIFT;
    const std::vector<int32_t>& fob = h->frame_of_block;
    Sub* d_subs = nullptr; int32_t* d_data = nullptr; int32_t* d_planes = nullptr; int32_t* d_fob = nullptr; int32_t* d_assign = nullptr; int64_t* d_start = nullptr;
    // stream-ordered scratch: everything is released behind the kernels that use it
    AVX_HIP_CHECK(hipMallocAsync((void**)&d_subs, sizeof(Sub) * (size_t)nsub, s));
    AVX_HIP_CHECK(hipMallocAsync((void**)&d_data, sizeof(int32_t) * h->data.size(), s));
    AVX_HIP_CHECK(hipMallocAsync((void**)&d_planes, sizeof(int32_t) * (size_t)total * h->channels, s));
    AVX_HIP_CHECK(hipMallocAsync((void**)&d_fob, sizeof(int32_t) * fob.size(), s));
    AVX_HIP_CHECK(hipMallocAsync((void**)&d_assign, sizeof(int32_t) * (size_t)nfr, s));
    AVX_HIP_CHECK(hipMallocAsync((void**)&d_start, sizeof(int64_t) * (size_t)nfr, s));
    AVX_HIP_CHECK(hipMemcpyAsync(d_subs, h->subs.data(), sizeof(Sub) * (size_t)nsub, hipMemcpyHostToDevice, s));
    AVX_HIP_CHECK(hipMemcpyAsync(d_data, h->data.data(), sizeof(int32_t) * h->data.size(), hipMemcpyHostToDevice, s));
    AVX_HIP_CHECK(hipMemcpyAsync(d_fob, fob.data(), sizeof(int32_t) * fob.size(), hipMemcpyHostToDevice, s));
    AVX_HIP_CHECK(hipMemcpyAsync(d_assign, h->frame_assign.data(), sizeof(int32_t) * (size_t)nfr, hipMemcpyHostToDevice, s));
    AVX_HIP_CHECK(hipMemcpyAsync(d_start, h->frame_start.data(), sizeof(int64_t) * (size_t)nfr, hipMemcpyHostToDevice, s));
    // (the host vectors are pageable: the copies above have completed against them when hipMemcpyAsync returns)
    hipLaunchKernelGGL(flac_predict_kernel, dim3((nsub + 63) / 64), dim3(64), 0, s, d_subs, nsub, d_data, d_planes, total);
    const int justify = left_justify ? 32 - h->bps : 0;
    hipLaunchKernelGGL(flac_interleave_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, d_planes, total, h->channels, total, d_fob, shift, d_assign, d_start,
                       nfr, justify, out_dev);
    AVX_LAUNCH_CHECK();
    (void)hipFreeAsync(d_subs, s); (void)hipFreeAsync(d_data, s); (void)hipFreeAsync(d_planes, s);
    (void)hipFreeAsync(d_fob, s); (void)hipFreeAsync(d_assign, s); (void)hipFreeAsync(d_start, s);
    return AVEXHIP_OK;
}
