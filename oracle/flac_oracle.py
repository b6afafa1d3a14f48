"""CPU oracle of the FLAC decoder: the format's published specification restated in plain Python.

TEST INFRASTRUCTURE ONLY (tests/ may import it; the product path is avex_amd/csrc/flac.hip).  The reference decodes its .flac samples
with third-party libraries (torchaudio.load / soundfile -> libFLAC; avex/data/augmentations.py:258-262) that are absent from the
reference tree and from both machines, so this follows the format specification (xiph.org "FLAC format", RFC 9639) instead.

PINNED all the same: every FLAC stream carries, in STREAMINFO, the MD5 of its unencoded audio.  ``flac_decode`` verifies it, so the two
real libFLAC-encoded files the reference's own tests hold (tests/samples/animalspeak2/16khz/...: copied as data to tests/golden/flac/) are
known-answer tests for this file and, through it, for the device decoder -- bit for bit.
"""
from __future__ import annotations

import hashlib
from typing import Dict, Tuple

import numpy as np

FIXED = {0: (), 1: (1,), 2: (2, -1), 3: (3, -3, 1), 4: (4, -6, 4, -1)}


class _Bits:
    def __init__(self, data: bytes, pos: int) -> None:
        self.d, self.pos = data, pos * 8          # bit position

    def bits(self, k: int) -> int:
        if k == 0:
            return 0
        p, e = self.pos, self.pos + k
        b0, b1 = p >> 3, (e + 7) >> 3
        if b1 > len(self.d):
            raise ValueError("truncated FLAC stream")
        v = int.from_bytes(self.d[b0:b1], "big")
        self.pos = e
        return (v >> (b1 * 8 - e)) & ((1 << k) - 1)

    def sbits(self, k: int) -> int:
        v = self.bits(k)
        return v - (1 << k) if k and v >> (k - 1) else v

    def unary(self) -> int:
        q = 0
        while True:
            byte = self.pos >> 3
            if byte >= len(self.d):
                raise ValueError("truncated FLAC stream")
            off = self.pos & 7
            rest = (self.d[byte] << off) & 0xFF
            if rest == 0:
                q += 8 - off
                self.pos += 8 - off
                continue
            lead = 8 - rest.bit_length()
            self.pos += lead + 1
            return q + lead

    def align(self) -> None:
        self.pos = (self.pos + 7) & ~7


def crc8(b: bytes) -> int:
    c = 0
    for x in b:
        c ^= x
        for _ in range(8):
            c = ((c << 1) ^ 0x07) & 0xFF if c & 0x80 else (c << 1) & 0xFF
    return c


_CRC16 = []
for _i in range(256):
    _c = _i << 8
    for _ in range(8):
        _c = ((_c << 1) ^ 0x8005) & 0xFFFF if _c & 0x8000 else (_c << 1) & 0xFFFF
    _CRC16.append(_c)


def crc16(b: bytes) -> int:
    c = 0
    for x in b:
        c = ((c << 8) & 0xFFFF) ^ _CRC16[(c >> 8) ^ x]
    return c


def streaminfo(data: bytes) -> Tuple[Dict[str, object], int]:
    """STREAMINFO fields and the byte offset of the first frame."""
    if data[:4] != b"fLaC":
        raise ValueError("not a FLAC stream")
    pos, info = 4, None
    while True:
        last, typ = data[pos] >> 7, data[pos] & 0x7F
        ln = int.from_bytes(data[pos + 1:pos + 4], "big")
        body = data[pos + 4:pos + 4 + ln]
        if typ == 0:
            x = int.from_bytes(body[10:18], "big")
            info = dict(min_block=int.from_bytes(body[0:2], "big"), max_block=int.from_bytes(body[2:4], "big"), sample_rate=x >> 44,
                        channels=((x >> 41) & 7) + 1, bps=((x >> 36) & 31) + 1, total=x & ((1 << 36) - 1), md5=bytes(body[18:34]))
        pos += 4 + ln
        if last:
            break
    if info is None:
        raise ValueError("no STREAMINFO")
    return info, pos


def _subframe(br: _Bits, bs: int, bps: int) -> list:
    if br.bits(1):
        raise ValueError("subframe padding bit set")
    typ = br.bits(6)
    wasted = br.unary() + 1 if br.bits(1) else 0
    bps -= wasted
    if typ == 0:
        out = [br.sbits(bps)] * bs
    elif typ == 1:
        out = [br.sbits(bps) for _ in range(bs)]
    else:
        if 8 <= typ <= 12:
            order = typ - 8
            out = [br.sbits(bps) for _ in range(order)]
            coef, shift = FIXED[order], 0
        elif typ >= 32:
            order = typ - 31
            out = [br.sbits(bps) for _ in range(order)]
            prec = br.bits(4) + 1
            if prec == 16:
                raise ValueError("invalid LPC precision")
            shift = br.sbits(5)
            if shift < 0:
                raise ValueError("negative LPC shift")
            coef = tuple(br.sbits(prec) for _ in range(order))
        else:
            raise ValueError(f"reserved subframe type {typ}")
        method = br.bits(2)
        if method > 1:
            raise ValueError("reserved residual coding method")
        pbits, esc = (4, 15) if method == 0 else (5, 31)
        porder = br.bits(4)
        res = []
        for part in range(1 << porder):
            cnt = bs - order if porder == 0 else ((bs >> porder) - order if part == 0 else bs >> porder)
            k = br.bits(pbits)
            if k == esc:
                raw = br.bits(5)
                res.extend(br.sbits(raw) for _ in range(cnt))
            else:
                unary, bits = br.unary, br.bits
                for _ in range(cnt):
                    u = (unary() << k) | bits(k)
                    res.append((u >> 1) ^ -(u & 1))
        rc = coef[::-1]
        for r in res:                                   # s[t] = r[t] + (sum_j coef[j] s[t-1-j] >> shift), exact integers
            acc = 0
            if order:
                for c, s in zip(rc, out[-order:]):
                    acc += c * s
            out.append(r + (acc >> shift))
    if wasted:
        out = [v << wasted for v in out]
    return out


def flac_decode(data: bytes, verify_md5: bool = True) -> Tuple[np.ndarray, Dict[str, object]]:
    """``(samples int32 [total, channels], streaminfo)``; raises ``ValueError`` on any CRC / MD5 / syntax violation."""
    info, pos = streaminfo(data)
    nch, total = info["channels"], info["total"]
    planes = [[] for _ in range(nch)]
    done = 0
    ss_table = {1: 8, 2: 12, 4: 16, 5: 20, 6: 24, 7: 32}
    while pos + 2 <= len(data) and (total == 0 or done < total):
        if not (data[pos] == 0xFF and (data[pos + 1] & 0xFE) == 0xF8):
            raise ValueError(f"lost frame synchronisation at byte {pos}")
        br = _Bits(data, pos)
        br.bits(16)
        bs_code, sr_code, assign, ss_code = br.bits(4), br.bits(4), br.bits(4), br.bits(3)
        br.bits(1)
        first = br.bits(8)
        extra = 0
        if first & 0x80:
            m = 0x40
            while first & m:
                extra += 1
                m >>= 1
        for _ in range(extra):
            br.bits(8)
        if bs_code == 1:
            bs = 192
        elif 2 <= bs_code <= 5:
            bs = 576 << (bs_code - 2)
        elif bs_code == 6:
            bs = br.bits(8) + 1
        elif bs_code == 7:
            bs = br.bits(16) + 1
        elif bs_code >= 8:
            bs = 256 << (bs_code - 8)
        else:
            raise ValueError("reserved block size code")
        if sr_code == 12:
            br.bits(8)
        elif sr_code in (13, 14):
            br.bits(16)
        hdr = br.pos >> 3
        if crc8(data[pos:hdr]) != br.bits(8):
            raise ValueError("frame header CRC-8 mismatch")
        bps = info["bps"] if ss_code == 0 else ss_table[ss_code]
        if (assign + 1 if assign < 8 else 2) != nch:
            raise ValueError("channel count differs from STREAMINFO")
        subs = []
        for ch in range(nch):
            side = (assign == 8 and ch == 1) or (assign == 9 and ch == 0) or (assign == 10 and ch == 1)
            subs.append(_subframe(br, bs, bps + (1 if side else 0)))
        br.align()
        end = br.pos >> 3
        if crc16(data[pos:end]) != br.bits(16):
            raise ValueError("frame CRC-16 mismatch")
        if assign == 8:
            subs[1] = [a - b for a, b in zip(subs[0], subs[1])]
        elif assign == 9:
            subs[0] = [a + b for a, b in zip(subs[0], subs[1])]
        elif assign == 10:
            mids = [(a << 1) | (b & 1) for a, b in zip(subs[0], subs[1])]
            subs = [[(m + s) >> 1 for m, s in zip(mids, subs[1])], [(m - s) >> 1 for m, s in zip(mids, subs[1])]]
        for ch in range(nch):
            planes[ch].extend(subs[ch])
        pos = end + 2
        done += bs
    out = np.array(planes, dtype=np.int64).T.astype(np.int32).reshape(done, nch)
    if total and done != total:
        raise ValueError(f"decoded {done} samples, STREAMINFO says {total}")
    if verify_md5 and any(info["md5"]):
        if pcm_md5(out, info["bps"]) != info["md5"]:
            raise ValueError("MD5 of the decoded audio differs from STREAMINFO's")
    return out, info


def pcm_md5(samples: np.ndarray, bps: int) -> bytes:
    """MD5 as FLAC defines it: interleaved samples, little-endian, ceil(bps / 8) bytes each, sign-extended."""
    nbytes = (bps + 7) // 8
    a = np.ascontiguousarray(samples.astype("<i4")).view(np.uint8).reshape(-1, 4)[:, :nbytes]
    return hashlib.md5(np.ascontiguousarray(a).tobytes()).digest()
