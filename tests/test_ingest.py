"""Ingest: WAV container parsing on the host (no GPU), PCM -> mono and the sinc resampler on the device vs the oracle."""
import io
import os
import struct
import wave

import numpy as np
import pytest
import torch

from avex_amd import ingest, synth
from oracle import ingest_oracle as IO


def _wav_bytes(x: np.ndarray, sr: int, width: int) -> bytes:
    """int PCM WAV via the standard library (width in bytes); x is [frames, channels] float in [-1, 1)."""
    buf = io.BytesIO()
    with wave.open(buf, "wb") as w:
        w.setnchannels(x.shape[1]); w.setsampwidth(width); w.setframerate(sr)
        if width == 1:
            w.writeframes((np.clip(x * 128 + 128, 0, 255)).astype(np.uint8).tobytes())
        elif width == 2:
            w.writeframes((x * 32767).astype("<i2").tobytes())
        elif width == 3:
            v = (x * 8388607).astype(np.int32)
            w.writeframes(np.stack([v & 255, (v >> 8) & 255, (v >> 16) & 255], -1).astype(np.uint8).tobytes())
        else:
            w.writeframes((x * 2147483647).astype("<i4").tobytes())
    return buf.getvalue()


def _float_wav_bytes(x: np.ndarray, sr: int, bits: int) -> bytes:
    data = x.astype("<f4" if bits == 32 else "<f8").tobytes()
    ch = x.shape[1]
    fmt = struct.pack("<HHIIHH", 3, ch, sr, sr * ch * bits // 8, ch * bits // 8, bits)
    return b"RIFF" + struct.pack("<I", 4 + 8 + len(fmt) + 8 + len(data)) + b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"data" + struct.pack("<I", len(data)) + data


def test_parse_wav_and_oracle_decoding():
    x = (synth.normal("wavx", (1000, 2), 0.2)).astype(np.float32).clip(-0.99, 0.99)
    for width, fmt, tol in ((1, 8, 1e-2), (2, 16, 8e-5), (3, 24, 4e-7), (4, 32, 1e-7)):
        raw, sr, ch, code = ingest.parse_wav(_wav_bytes(x, 44100, width))
        assert (sr, ch, code) == (44100, 2, fmt) and raw.dtype == np.uint8 and raw.size == 1000 * 2 * width
        assert np.abs(IO.pcm_to_mono(raw, ch, code) - x.mean(1)).max() < tol
    raw, sr, ch, code = ingest.parse_wav(_float_wav_bytes(x, 22050, 32))
    assert (sr, ch, code) == (22050, 2, 0) and np.allclose(IO.pcm_to_mono(raw, ch, code), x.mean(1), atol=1e-7)
    raw, sr, ch, code = ingest.parse_wav(_float_wav_bytes(x[:, :1], 8000, 64))
    assert (sr, ch, code) == (8000, 1, 64) and np.allclose(IO.pcm_to_mono(raw, ch, code), x[:, 0], atol=1e-7)
    with pytest.raises(ValueError):
        ingest.parse_wav(b"fLaC" + b"\\0" * 64)
    # malformed headers are ValueError, never struct.error / ZeroDivisionError
    good = _wav_bytes(x, 44100, 2)
    fmt_at = good.index(b"fmt ") + 8

    def patched(off, fmtc, *vals):
        b = bytearray(good)
        struct.pack_into(fmtc, b, fmt_at + off, *vals)
        return bytes(b)
    for bad in (patched(2, "<H", 0),                      # zero channels
                patched(4, "<I", 0),                      # zero sample rate
                patched(12, "<H", 6),                     # block_align that is not channels x bytes per sample
                patched(14, "<H", 12),                    # a width that is not whole bytes
                good[:fmt_at - 4] + struct.pack("<I", 8) + good[fmt_at:fmt_at + 8] + good[fmt_at + 16:],      # fmt chunk of 8 bytes
                patched(0, "<H", 0xFFFE)):                # WAVE_FORMAT_EXTENSIBLE tag on a 16-byte fmt chunk
        with pytest.raises(ValueError):
            ingest.parse_wav(bad)


def test_oracle_resampler_properties():
    """The restated sinc resampler: output length ceil(new T / orig), a tone below both Nyquists keeps frequency and amplitude, DC stays DC
    (away from the edges), identity for equal rates."""
    sr, new = 44100, 16000
    t = np.arange(44100) / sr
    y = IO.resample(np.sin(2 * np.pi * 1000 * t).astype(np.float32), sr, new)
    assert y.shape == (16000,)
    tn = np.arange(16000) / new
    assert np.abs(y[200:-200] - np.sin(2 * np.pi * 1000 * tn)[200:-200]).max() < 2e-3
    dc = IO.resample(np.ones(8000, np.float32), 48000, 16000)
    assert dc.shape == (2667,) and np.abs(dc[50:-50] - 1.0).max() < 2e-3
    x = synth.normal("rsx", (3, 500), 1.0)
    assert np.array_equal(IO.resample(x, 16000, 16000), x)
    assert IO.resample(x, 8000, 16000).shape == (3, 1000)


@pytest.mark.gpu
@pytest.mark.parametrize("orig,new", [(44100, 16000), (48000, 16000), (22050, 16000), (8000, 16000), (32000, 16000), (16000, 22050)])
def test_device_resampler_matches_oracle(built_lib, orig, new):
    x = synth.normal(f"rs{orig}", (3, 30011), 0.3)
    ref = IO.resample(x, orig, new)
    rs = ingest.Resampler(orig, new)
    y = rs(torch.from_numpy(x).cuda())
    assert y.shape == ref.shape == (3, rs.out_length(30011))
    assert np.abs(y.cpu().numpy() - ref).max() < 2e-6
    y1 = rs(torch.from_numpy(x[1]).cuda())
    assert torch.equal(y1, y[1])
    yk = ingest.Resampler(orig, new, lowpass_filter_width=16, rolloff=0.945, beta=14.77)(torch.from_numpy(x).cuda())      # torchaudio's "kaiser_best"-like settings
    assert np.abs(yk.cpu().numpy() - IO.resample(x, orig, new, 16, 0.945, 14.77)).max() < 2e-6


def test_oracle_librosa_resampler_properties():
    """The restated resampy / librosa `kaiser_best` path (birdset_train_splits.py:190-196): ceil(T ratio) samples, the last one(s) beyond
    int(T ratio) zero-filled, integer table steps (2:1, 1:2) reproduce DC exactly, the others carry resampy's index_step truncation
    (int(scale * 512): a few 1e-3 of gain), a tone keeps its frequency, `scale` divides by sqrt(ratio)."""
    y = IO.resample_librosa(np.ones(32000, np.float32), 32000, 16000, scale=False)
    assert y.shape == (16000,) and np.abs(y[200:-200] - 1.0).max() < 1e-6
    y = IO.resample_librosa(np.ones(22051, np.float32), 22050, 16000, scale=False)
    assert y.shape == (16001,) and y[-1] == 0.0 and np.abs(y[200:-200] - 1.0).max() < 1e-3
    t = np.arange(48000) / 48000.0
    y = IO.resample_librosa(np.sin(2 * np.pi * 1000 * t).astype(np.float32), 48000, 16000, scale=False)
    ref = np.sin(2 * np.pi * 1000 * np.arange(16000) / 16000.0)
    assert np.abs(y[300:-300] - 1.0027175 * ref[300:-300]).max() < 5e-4
    ys = IO.resample_librosa(np.sin(2 * np.pi * 1000 * t).astype(np.float32), 48000, 16000, scale=True)
    assert np.allclose(ys, y / np.sqrt(1 / 3), atol=1e-6)
    x = synth.normal("rsl", (500,), 1.0)
    assert np.array_equal(IO.resample_librosa(x, 16000, 16000), x)
    assert IO.resample_librosa(x, 8000, 16000).shape == (1000,)


@pytest.mark.gpu
@pytest.mark.parametrize("orig,new", [(44100, 16000), (48000, 16000), (22050, 16000), (8000, 16000), (32000, 16000), (16000, 22050)])
def test_device_librosa_resampler_matches_oracle(built_lib, orig, new):
    x = synth.normal(f"rsl{orig}", (3, 30011), 0.3)
    rs = ingest.Resampler(orig, new, res_type="kaiser_best")
    y = rs(torch.from_numpy(x).cuda()).cpu().numpy()
    assert y.shape == (3, rs.out_length(30011)) and rs.out_length(30011) == int(np.ceil(30011 * new / orig))
    for b in range(3):
        ref = IO.resample_librosa(x[b], orig, new)
        assert ref.shape == y[b].shape and np.abs(y[b] - ref).max() < 5e-6
    yf = ingest.Resampler(orig, new, res_type="kaiser_fast", scale=False)(torch.from_numpy(x[0]).cuda()).cpu().numpy()
    assert np.abs(yf - IO.resample_librosa(x[0], orig, new, res_type="kaiser_fast", scale=False)).max() < 5e-6
    with pytest.raises(ValueError):
        ingest.Resampler(orig, new, res_type="soxr_hq")


@pytest.mark.gpu
def test_load_audio_end_to_end(built_lib, tmp_path):
    """A 44.1 kHz stereo 24-bit file -> mono 16 kHz on the device, then through the BEATs frontend like any clip."""
    x = (synth.normal("wavfile", (44100, 2), 0.2)).astype(np.float32).clip(-0.99, 0.99)
    path = tmp_path / "clip.wav"
    path.write_bytes(_wav_bytes(x, 44100, 3))
    wav, sr = ingest.load_audio(str(path), 16000)
    raw, _, ch, code = ingest.parse_wav(str(path))
    ref = IO.resample(IO.pcm_to_mono(raw, ch, code), 44100, 16000)
    assert sr == 16000 and wav.shape == (16000,) and np.abs(wav.cpu().numpy() - ref).max() < 2e-6
    same, sr2 = ingest.load_audio(_wav_bytes(x[:, :1], 16000, 2), 16000)
    assert sr2 == 16000 and np.abs(same.cpu().numpy() - (x[:, 0] * 32767).astype(np.int16) / 32768.0).max() < 1e-7
    from avex_amd import kernels as K
    fb = K.FbankPlan()(wav.unsqueeze(0))
    assert fb.shape == (1, 98, 128) and torch.isfinite(fb).all()


# ------------------------------------------------------------------------------------------------------------------------
# FLAC: host bitstream parser + device predictors, bit-exact
# ------------------------------------------------------------------------------------------------------------------------
FLAC_FIXTURES = {"inaturalist_246886.flac": (505313, "843507851a89aa36d4ca0dd93b9720d1"), "xenocanto_XC564654.flac": (386361, "f795f9fe47921f35a64288483dfca4a1")}


def _fixture(name):
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "flac", name)


def test_flac_oracle_reproduces_the_md5_of_the_reference_samples():
    """The two libFLAC-encoded files of the reference's own tests: the oracle's output hashes to the MD5 of the unencoded audio that
    the encoder stored in STREAMINFO -- the restatement is pinned bit for bit."""
    from oracle import flac_oracle as F
    for name, (total, md5) in FLAC_FIXTURES.items():
        data = open(_fixture(name), "rb").read()
        x, info = F.flac_decode(data)                     # raises on any CRC-8 / CRC-16 / MD5 mismatch
        assert x.shape == (total, 1) and x.dtype == np.int32 and info["sample_rate"] == 16000 and info["bps"] == 16
        assert F.pcm_md5(x, 16).hex() == md5 == info["md5"].hex()


def test_flac_encoder_cases_round_trip_through_the_oracle():
    """Every syntax element the fixtures lack (CONSTANT / VERBATIM / FIXED subframes, the three stereo decorrelations, 4- and 5-bit Rice
    parameters, escaped partitions, wasted bits, explicit block sizes, 8 / 12 / 24-bit samples, three channels), encoded by tests/_flac_enc.py."""
    import _flac_enc as E
    from oracle import flac_oracle as F
    for name, (pcm, sr, bps, bs, plan) in E.cases().items():
        data = E.encode(pcm, sr, bps, bs, plan)
        x, info = F.flac_decode(data)
        assert np.array_equal(x, pcm.astype(np.int32)), name
        assert (info["sample_rate"], info["bps"], info["channels"], info["total"]) == (sr, bps, pcm.shape[1], pcm.shape[0])


def test_flac_host_parser_accepts_and_refuses(built_lib):
    """avexhip_flac_open runs on the host: stream info of the fixtures, every CRC checked, violations refused with ValueError."""
    import _flac_enc as E
    for name, (total, md5) in FLAC_FIXTURES.items():
        f = ingest.FlacStream(_fixture(name))
        assert (f.sample_rate, f.channels, f.bits_per_sample, f.total_samples, f.md5.hex()) == (16000, 1, 16, total, md5)
        f.close()
    for name, (pcm, sr, bps, bs, plan) in E.cases().items():
        f = ingest.FlacStream(E.encode(pcm, sr, bps, bs, plan))
        assert (f.sample_rate, f.channels, f.bits_per_sample, f.total_samples) == (sr, pcm.shape[1], bps, pcm.shape[0]), name
    data = bytearray(open(_fixture("xenocanto_XC564654.flac"), "rb").read())
    bad = bytearray(data); bad[3000] ^= 0x04                                   # a flipped bit inside a frame: CRC-16
    with pytest.raises(ValueError, match="CRC"):
        ingest.FlacStream(bytes(bad))
    bad = bytearray(data); bad[88] ^= 0x01                                     # inside the first frame header: CRC-8 (or a reserved code)
    with pytest.raises(ValueError):
        ingest.FlacStream(bytes(bad))
    with pytest.raises(ValueError):
        ingest.FlacStream(bytes(data[:20000]))                                  # truncated
    with pytest.raises(ValueError, match="fLaC"):
        ingest.FlacStream(b"RIFF" + bytes(64))
    with pytest.raises(ValueError):
        ingest.parse_wav(bytes(data))                                           # and the WAV parser does not take it for a WAV


def test_flac_host_parser_refuses_order_beyond_block(built_lib):
    """A crafted frame: block-size code 6 with a size byte of 0 (a ONE-sample block) and an LPC subframe of order 32.  The warm-up samples
    of such a subframe would be written past the block's buffer; the parser must refuse it before writing anything (valid CRCs, so only
    the order check can stop it).  Same for a FIXED subframe of order 4 in a two-sample block."""
    import _flac_enc as E
    from oracle.flac_oracle import crc8, crc16

    def stream(size_byte, sub_type, warm):
        bw = E.Bits()
        bw.put(0x3FFE, 14); bw.put(0, 1); bw.put(0, 1)
        bw.put(6, 4); bw.put(5, 4); bw.put(0, 4); bw.put(4, 3); bw.put(0, 1)      # bs code 6, 16 kHz, mono, 16 bit
        bw.put(0, 8)                                                                # frame number 0
        bw.put(size_byte, 8)                                                        # block size - 1
        hdr = bw.bytes()
        bw.put(crc8(hdr), 8)
        bw.put(0, 1); bw.put(sub_type, 6); bw.put(0, 1)                             # subframe header
        for _ in range(warm):
            bw.put(0x1234, 16)
        bw.put(11, 4); bw.put(3, 5)                                                 # precision 12, shift 3 (LPC only reads them; harmless for FIXED)
        for _ in range(32):
            bw.put(1, 12)
        bw.put(0, 2); bw.put(0, 4); bw.put(0, 4)
        bw.align()
        body = bw.bytes()
        frame = body + crc16(body).to_bytes(2, "big")
        n = size_byte + 1
        x = (16000 << 44) | (0 << 41) | (15 << 36) | n
        info = (16).to_bytes(2, "big") * 2 + len(frame).to_bytes(3, "big") * 2 + x.to_bytes(8, "big") + bytes(16)
        return b"fLaC" + bytes([0x80, 0, 0, 34]) + info + frame

    with pytest.raises(ValueError, match="predictor order"):
        ingest.FlacStream(stream(0, 63, 32))          # LPC order 32 in a 1-sample block
    with pytest.raises(ValueError, match="predictor order"):
        ingest.FlacStream(stream(1, 12, 4))           # FIXED order 4 in a 2-sample block


@pytest.mark.gpu
def test_flac_device_decoder_is_bit_exact(built_lib):
    import hashlib
    import _flac_enc as E
    from oracle import flac_oracle as F
    for name, (total, md5) in FLAC_FIXTURES.items():
        f = ingest.FlacStream(_fixture(name))
        x = f.decode().cpu().numpy()
        assert x.shape == (total, 1)
        assert F.pcm_md5(x, 16).hex() == md5                                   # the encoder's input, bit for bit
        xj = f.decode(left_justify=True).cpu().numpy()
        assert np.array_equal(xj, x << 16)
    for name, (pcm, sr, bps, bs, plan) in E.cases().items():
        f = ingest.FlacStream(E.encode(pcm, sr, bps, bs, plan))
        assert np.array_equal(f.decode().cpu().numpy(), pcm.astype(np.int32)), name


@pytest.mark.gpu
def test_load_audio_reads_flac(built_lib):
    """load_audio on a FLAC file = the decoded integers / 2^(bits - 1), channels averaged, resampled like a WAV of the same samples."""
    import _flac_enc as E
    from oracle import flac_oracle as F
    x, sr = ingest.load_audio(_fixture("xenocanto_XC564654.flac"), target_sr=None)
    ref, _ = F.flac_decode(open(_fixture("xenocanto_XC564654.flac"), "rb").read())
    assert sr == 16000 and np.array_equal(x.cpu().numpy(), ref[:, 0].astype(np.float32) / np.float32(32768.0))
    pcm, fsr, bps, bs, plan = E.cases()["stereo16_all_modes"]
    y, sr2 = ingest.load_audio(E.encode(pcm, fsr, bps, bs, plan), target_sr=None)
    assert np.allclose(y.cpu().numpy(), pcm.astype(np.float32).mean(1) / 32768.0, atol=1e-7)
    z, sr3 = ingest.load_audio(_fixture("xenocanto_XC564654.flac"), target_sr=8000)
    want = IO.resample(ref[:, 0].astype(np.float32) / np.float32(32768.0), 16000, 8000)
    assert sr3 == 8000 and np.abs(z.cpu().numpy() - want).max() < 2e-6
