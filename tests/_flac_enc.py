"""A small FLAC ENCODER for tests (test infrastructure): known PCM in, a standards-conforming stream out, so that the decoder can be
checked bit-exactly on every syntax element the two real fixture files (mono, LPC only) do not contain: CONSTANT / VERBATIM / FIXED
subframes, left-side / side-right / mid-side stereo, 4- and 5-bit Rice parameters, escaped (raw) partitions, wasted bits, explicit
8- and 16-bit block sizes, 8 / 12 / 20 / 24-bit samples, multi-channel.  Written from the format specification (RFC 9639), like the decoder;
the CRC-8 / CRC-16 and the STREAMINFO MD5 it writes are checked by the decoder under test."""
import hashlib

import numpy as np

from oracle.flac_oracle import crc8, crc16

FIXED = {0: (), 1: (1,), 2: (2, -1), 3: (3, -3, 1), 4: (4, -6, 4, -1)}


class Bits:
    def __init__(self):
        self.v, self.n = 0, 0

    def put(self, val, k):
        if k:
            self.v = (self.v << k) | (int(val) & ((1 << k) - 1))
            self.n += k

    def unary(self, q):
        self.put(1, q + 1)

    def align(self):
        self.put(0, (-self.n) % 8)

    def bytes(self):
        assert self.n % 8 == 0
        return self.v.to_bytes(self.n // 8, "big")


def _utf8(n):
    if n < 0x80:
        return bytes([n])
    out, lead = [], 0
    nb = 2
    while n >= (1 << (5 * nb + 1)):
        nb += 1
    for i in range(nb - 1):
        out.append(0x80 | ((n >> (6 * i)) & 0x3F))
    lead = ((0xFF << (8 - nb)) & 0xFF) | (n >> (6 * (nb - 1)))
    return bytes([lead] + out[::-1])


def _residual(bw, res, order, bs, porder, method, escape_parts=()):
    pbits, esc = (4, 15) if method == 0 else (5, 31)
    bw.put(method, 2)
    bw.put(porder, 4)
    t = 0
    for part in range(1 << porder):
        cnt = bs - order if porder == 0 else ((bs >> porder) - order if part == 0 else bs >> porder)
        seg = res[t:t + cnt]
        t += cnt
        if part in escape_parts:
            raw = max([1] + [int(abs(int(v))).bit_length() + 1 for v in seg])
            bw.put(esc, pbits)
            bw.put(raw, 5)
            for v in seg:
                bw.put(v, raw)
            continue
        mean = float(np.mean(np.abs(seg))) if len(seg) else 0.0
        k = min(max(int(np.log2(mean + 1.0)), 0), esc - 1)
        bw.put(k, pbits)
        for v in seg:
            v = int(v)
            u = (v << 1) if v >= 0 else ((-v << 1) - 1)
            bw.unary(u >> k)
            bw.put(u & ((1 << k) - 1), k)


def _subframe(bw, x, bps, kind, order=0, coef=None, shift=0, prec=12, porder=0, method=0, escape_parts=(), wasted=0):
    """x: the channel's samples as Python ints (already the side / mid signal where applicable)."""
    bs = len(x)
    if wasted:
        assert all(v % (1 << wasted) == 0 for v in x)
        x = [v >> wasted for v in x]
        bps -= wasted
    code = {"constant": 0, "verbatim": 1}.get(kind)
    if kind == "fixed":
        code = 8 + order
    elif kind == "lpc":
        code = 31 + order
    bw.put(0, 1)
    bw.put(code, 6)
    if wasted:
        bw.put(1, 1)
        bw.unary(wasted - 1)
    else:
        bw.put(0, 1)
    if kind == "constant":
        assert all(v == x[0] for v in x)
        bw.put(x[0], bps)
        return
    if kind == "verbatim":
        for v in x:
            bw.put(v, bps)
        return
    if kind == "fixed":
        coef, shift = FIXED[order], 0
    for v in x[:order]:
        bw.put(v, bps)
    if kind == "lpc":
        bw.put(prec - 1, 4)
        bw.put(shift, 5)
        for c in coef:
            bw.put(c, prec)
    res = []
    for t in range(order, bs):
        acc = sum(c * x[t - 1 - j] for j, c in enumerate(coef))
        res.append(x[t] - (acc >> shift))
    _residual(bw, res, order, bs, porder, method, escape_parts)


def encode(pcm: np.ndarray, sample_rate: int, bps: int, blocksize: int, plan):
    """``pcm`` int [n, channels]; ``plan(frame_index, n_in_block)`` -> dict(assign=0 | 8 | 9 | 10 (stereo modes), subs=[kwargs per channel for _subframe]).
    Returns the stream bytes (fixed-blocksize stream; the last block is shorter when n is not a multiple of blocksize)."""
    n, nch = pcm.shape
    bs_codes = {192: 1, 576: 2, 1152: 3, 2304: 4, 4608: 5, 256: 8, 512: 9, 1024: 10, 2048: 11, 4096: 12, 8192: 13, 16384: 14, 32768: 15}
    ss_codes = {8: 1, 12: 2, 16: 4, 20: 5, 24: 6, 32: 7}
    sr_codes = {88200: 1, 176400: 2, 192000: 3, 8000: 4, 16000: 5, 22050: 6, 24000: 7, 32000: 8, 44100: 9, 48000: 10, 96000: 11}
    frames = []
    fi = 0
    for start in range(0, n, blocksize):
        blk = pcm[start:start + blocksize]
        bs = len(blk)
        p = plan(fi, bs)
        assign = p.get("assign", nch - 1)
        bw = Bits()
        bw.put(0x3FFE, 14); bw.put(0, 1); bw.put(0, 1)
        if bs in bs_codes and bs == blocksize:
            bcode, btail = bs_codes[bs], None
        elif bs <= 256:
            bcode, btail = 6, (bs - 1, 8)
        else:
            bcode, btail = 7, (bs - 1, 16)
        bw.put(bcode, 4)
        bw.put(sr_codes.get(sample_rate, 0), 4)
        bw.put(assign, 4)
        bw.put(ss_codes.get(bps, 0), 3)
        bw.put(0, 1)
        for b in _utf8(fi):
            bw.put(b, 8)
        if btail:
            bw.put(*btail)
        hdr = bw.bytes()
        bw.put(crc8(hdr), 8)
        cols = [[int(v) for v in blk[:, c]] for c in range(nch)]
        widths = [bps] * nch
        if assign == 8:
            cols = [cols[0], [a - b for a, b in zip(cols[0], cols[1])]]; widths = [bps, bps + 1]
        elif assign == 9:
            cols = [[a - b for a, b in zip(cols[0], cols[1])], cols[1]]; widths = [bps + 1, bps]
        elif assign == 10:
            cols = [[(a + b) >> 1 for a, b in zip(cols[0], cols[1])], [a - b for a, b in zip(cols[0], cols[1])]]; widths = [bps, bps + 1]
        for c in range(nch):
            _subframe(bw, cols[c], widths[c], **p["subs"][c])
        bw.align()
        body = bw.bytes()
        frames.append(body + crc16(body).to_bytes(2, "big"))
        fi += 1
    nbytes = (bps + 7) // 8
    md5 = hashlib.md5(np.ascontiguousarray(np.ascontiguousarray(pcm.astype("<i4")).view(np.uint8).reshape(-1, 4)[:, :nbytes]).tobytes()).digest()
    x = (sample_rate << 44) | ((nch - 1) << 41) | ((bps - 1) << 36) | n
    fmin, fmax = min(len(f) for f in frames), max(len(f) for f in frames)
    info = blocksize.to_bytes(2, "big") * 2 + fmin.to_bytes(3, "big") + fmax.to_bytes(3, "big") + x.to_bytes(8, "big") + md5
    pad = bytes([0x81, 0, 0, 4]) + b"\0" * 4                     # a PADDING block, flagged last
    return b"fLaC" + bytes([0x00, 0, 0, 34]) + info + pad + b"".join(frames)


def cases():
    """name -> (pcm, sample_rate, bps, blocksize, plan): every syntax element the decoder knows."""
    rng = np.random.RandomState(11)
    out = {}
    n = 5 * 1024 + 300
    t = np.arange(n)
    left = (9000 * np.sin(2 * np.pi * 440 * t / 16000) + 300 * rng.randn(n)).astype(np.int64)
    right = (8000 * np.sin(2 * np.pi * 440 * t / 16000 + 0.3) + 300 * rng.randn(n)).astype(np.int64)
    st = np.stack([left, right], 1)
    lpc = dict(kind="lpc", order=3, coef=(1900, -1100, 200), shift=10, prec=12, porder=2, method=0)

    def stereo_plan(fi, bs):
        assign = (1, 8, 9, 10, 10, 8)[fi % 6]
        a = dict(kind="fixed", order=(0, 1, 2, 3, 4)[fi % 5], porder=(0, 1, 3)[fi % 3] if bs % 8 == 0 else 0, method=fi % 2)
        b = dict(lpc, porder=lpc["porder"] if bs % 4 == 0 else 0, method=1 - fi % 2, escape_parts=(1,) if fi % 4 == 1 and bs % 4 == 0 else ())
        return dict(assign=assign, subs=[a, b])
    out["stereo16_all_modes"] = (st, 16000, 16, 1024, stereo_plan)
    mono = (left // 4 * 4).reshape(-1, 1)                       # two wasted bits
    out["mono16_wasted_bits_and_verbatim"] = (mono, 44100, 16, 4096, lambda fi, bs: dict(subs=[dict(kind="verbatim", wasted=2) if fi % 2 else dict(kind="fixed", order=2, wasted=2, porder=0)]))
    const = np.full((700, 1), -321, np.int64)
    out["mono8_constant_short_block"] = (const // 4, 8000, 8, 192, lambda fi, bs: dict(subs=[dict(kind="constant")]))
    big = (rng.randn(3000, 3) * 200000).astype(np.int64)
    out["three_channels_24bit"] = (big, 48000, 24, 576, lambda fi, bs: dict(subs=[dict(kind="fixed", order=1, porder=0, method=1), dict(kind="verbatim"),
                                                                                 dict(kind="lpc", order=2, coef=(3, -1), shift=1, prec=5, porder=0, method=1, escape_parts=(0,))]))
    odd = (rng.randn(1000, 2) * 1500).astype(np.int64)
    out["stereo12_block_of_250"] = (odd // 16 * 1, 22050, 12, 250, lambda fi, bs: dict(assign=10, subs=[dict(kind="fixed", order=2, porder=0), dict(kind="fixed", order=1, porder=0)]))
    fixed = {}
    for name, (pcm, sr, bps, bs, plan) in out.items():
        lim = 1 << (bps - 1)
        fixed[name] = (np.clip(pcm, -lim, lim - 1), sr, bps, bs, plan)
    return fixed
