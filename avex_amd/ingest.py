"""Audio ingest for the embedding path: file -> mono float32 at the model's sample rate, decoded and resampled on the GPU
(SURVEY.md section 8, row f4).

What the reference does on the host before a clip reaches a model: read the file (``soundfile`` / ``torchaudio.load``), average the
channels (``noise_wav.mean(dim=0)``, avex/data/augmentations.py:269-271; ``audio_stereo_to_mono(..., "average")``,
avex/data/birdset_train_splits.py:184-186) and resample when the rate differs (``torchaudio.transforms.Resample(sr, self.sr)``,
augmentations.py:274-276; ``librosa.resample(..., res_type="kaiser_best")``, birdset_train_splits.py:190-196).  Here the host only
parses the container: the raw PCM bytes go to the device as they are and ``avexhip_pcm_to_mono_f32`` / ``avexhip_resample_forward``
do the rest (a 44.1 kHz stereo minute is 10 MB over PCIe instead of 3.8 MB of finished floats, but no host core touches a sample).

Containers: RIFF/WAVE with integer PCM (8 / 16 / 24 / 32 bit) or IEEE float (32 / 64 bit), incl. WAVE_FORMAT_EXTENSIBLE, and -- round 3 --
FLAC (:class:`FlacStream`: the bitstream is parsed on the host, the predictors run on the device, bit-exact against the MD5 every
stream carries); lossy codecs (MP3, OGG Vorbis) need a codec library that neither machine has and raise ``ValueError`` -- decode
those with the reference's reader and hand the array to :func:`to_device_mono`.  The resampler is torchaudio's algorithm (Hann-windowed sinc); librosa's ``kaiser_best``
filter is a different low-pass design and is NOT reproduced sample for sample (PARITY UNPINNED for both: neither library is
installed; checker = oracle/ingest_oracle.py).
"""
from __future__ import annotations

import struct
from typing import Dict, Optional, Tuple, Union

import numpy as np
import torch

from . import _capi
from ._capi import AvexHipError, check, lib

__all__ = ["parse_wav", "FlacStream", "Resampler", "to_device_mono", "load_audio"]


def parse_wav(path_or_bytes: Union[str, bytes]) -> Tuple[np.ndarray, int, int, int]:
    """``(raw uint8 samples, sample_rate, channels, sample_format)`` of a RIFF/WAVE file; ``sample_format`` as avexhip_pcm_to_mono_f32
    takes it (8 / 16 / 24 / 32 integer PCM, 0 float32, 64 float64).  No sample is converted on the host."""
    data = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray)) else open(path_or_bytes, "rb").read()
    if len(data) < 12 or data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError("not a RIFF/WAVE file (only PCM / float WAV is decoded here)")
    pos, fmt, payload = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack_from("<I", data, pos + 4)[0]
        body = data[pos + 8: pos + 8 + size]
        if cid == b"fmt ":
            if len(body) < 16:
                raise ValueError(f"WAVE fmt chunk of {len(body)} bytes (at least 16 expected)")
            tag, ch, sr, _br, block_align, bits = struct.unpack_from("<HHIIHH", body, 0)
            if tag == 0xFFFE:                                           # WAVE_FORMAT_EXTENSIBLE: the real tag is in the sub-format GUID
                if len(body) < 26:
                    raise ValueError(f"WAVE_FORMAT_EXTENSIBLE fmt chunk of {len(body)} bytes (at least 26 expected)")
                tag = struct.unpack_from("<H", body, 24)[0]
            if ch <= 0 or sr <= 0:
                raise ValueError(f"WAVE fmt chunk with {ch} channels at {sr} Hz")
            if bits % 8 or block_align != ch * bits // 8:
                raise ValueError(f"WAVE fmt chunk: block_align {block_align} is not channels x bytes per sample ({ch} x {bits} bits); "
                                 "padded containers are not decoded here")
            fmt = (tag, ch, sr, bits)
        elif cid == b"data":
            payload = body
        pos += 8 + size + (size & 1)
    if fmt is None or payload is None:
        raise ValueError("WAVE file without a fmt or data chunk")
    tag, ch, sr, bits = fmt
    if tag == 1 and bits in (8, 16, 24, 32):
        code = bits
    elif tag == 3 and bits in (32, 64):
        code = 0 if bits == 32 else 64
    else:
        raise ValueError(f"unsupported WAVE encoding (format tag {tag}, {bits} bits)")
    frame_bytes = ch * bits // 8
    n = len(payload) // frame_bytes                                     # a truncated last frame is dropped
    return np.frombuffer(payload, dtype=np.uint8, count=n * frame_bytes), sr, ch, code


class FlacStream:
    """A FLAC stream parsed on the host (``avexhip_flac_open``: metadata, frame / subframe headers, Rice-coded residuals, every CRC) and
    decoded on the device (``avexhip_flac_decode_i32``: predictors + inter-channel decorrelation), bit-exact -- the stream's STREAMINFO
    carries the MD5 of its unencoded audio (``md5``).  Raises ``ValueError`` for anything that is not a well-formed FLAC stream."""

    def __init__(self, data: Union[str, bytes]) -> None:
        raw = data if isinstance(data, (bytes, bytearray)) else open(data, "rb").read()
        self._buf = bytes(raw)
        self._h = lib().avexhip_flac_open(self._buf, len(self._buf))
        if not self._h:
            raise ValueError(f"FLAC: {_capi.last_error()}")
        import ctypes as C
        sr, ch, bps, tot, md5 = C.c_int(), C.c_int(), C.c_int(), C.c_int64(), (C.c_uint8 * 16)()
        check(lib().avexhip_flac_info(self._h, C.byref(sr), C.byref(ch), C.byref(bps), C.byref(tot), md5), "flac_info")
        self.sample_rate, self.channels, self.bits_per_sample, self.total_samples, self.md5 = sr.value, ch.value, bps.value, tot.value, bytes(md5)

    def decode(self, device: Optional[torch.device] = None, left_justify: bool = False) -> torch.Tensor:
        """Interleaved ``[total_samples, channels]`` int32 samples on the device (``left_justify``: shifted to 32-bit full scale)."""
        _capi.require_gpu()
        dev = device or torch.device("cuda", torch.cuda.current_device())
        with torch.cuda.device(dev):
            out = torch.empty((self.total_samples, self.channels), dtype=torch.int32, device=dev)
            check(lib().avexhip_flac_decode_i32(self._h, out.data_ptr(), int(left_justify), torch.cuda.current_stream().cuda_stream), "flac_decode_i32")
        return out

    def close(self) -> None:
        if getattr(self, "_h", None):
            lib().avexhip_flac_close(self._h)
            self._h = None

    def __del__(self) -> None:
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


# resampy's published filters: (num_zeros, precision, rolloff, Kaiser beta)
RESAMPY_FILTERS = {"kaiser_best": (64, 9, 0.9475937167399596, 14.769656459379492),
                   "kaiser_fast": (16, 9, 0.85, 8.555504641634386)}


class Resampler:
    """``torchaudio.transforms.Resample(orig_freq, new_freq)`` on the device (defaults as torchaudio's: ``sinc_interp_hann``,
    ``lowpass_filter_width=6``, ``rolloff=0.99``; ``beta`` > 0 selects the Kaiser window, ``sinc_interp_kaiser``)."""

    def __init__(self, orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99, beta: float = 0.0,
                 res_type: Optional[str] = None, scale: bool = True) -> None:
        """``res_type="kaiser_best"`` selects librosa's resampler instead (``librosa.resample(y, orig_sr=, target_sr=, scale=True,
        res_type="kaiser_best")``, birdset_train_splits.py:190-196 = resampy's interpolating kernel with its published kaiser_best
        filter); ``scale`` is librosa's energy-preserving ``scale`` flag."""
        _capi.require_gpu()
        self.orig_freq, self.new_freq = int(orig_freq), int(new_freq)
        self._h = None
        if self.orig_freq != self.new_freq:
            if res_type is None:
                self._h = lib().avexhip_resample_plan_create(self.orig_freq, self.new_freq, int(lowpass_filter_width), float(rolloff), float(beta))
            elif res_type in RESAMPY_FILTERS:
                nz, prec, ro, kb = RESAMPY_FILTERS[res_type]
                self._h = lib().avexhip_resample_interp_plan_create(self.orig_freq, self.new_freq, nz, prec, ro, kb, int(bool(scale)))
            else:
                raise ValueError(f"res_type {res_type!r}: only {sorted(RESAMPY_FILTERS)} (librosa / resampy) or None (torchaudio sinc) are built")
            if not self._h:
                raise AvexHipError(f"resample_plan_create failed: {_capi.last_error()}")

    def out_length(self, T: int) -> int:
        return T if self._h is None else int(lib().avexhip_resample_out_length(self._h, T))

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        if not x.is_cuda or x.dtype != torch.float32:
            raise ValueError("Resampler takes float32 CUDA tensors ([T] or [B, T])")
        if self._h is None:
            return x
        squeeze = x.dim() == 1
        x2 = (x.unsqueeze(0) if squeeze else x).contiguous()
        B, T = x2.shape
        out = torch.empty((B, self.out_length(T)), dtype=torch.float32, device=x.device)
        check(lib().avexhip_resample_forward(self._h, x2.data_ptr(), B, T, T, out.data_ptr(), out.shape[1], torch.cuda.current_stream().cuda_stream),
              "resample_forward")
        return out[0] if squeeze else out

    def __del__(self) -> None:
        try:
            if getattr(self, "_h", None):
                lib().avexhip_resample_plan_destroy(self._h)
                self._h = None
        except Exception:  # noqa: BLE001
            pass


def to_device_mono(raw: Union[np.ndarray, torch.Tensor], channels: int, sample_format: int, device: Optional[torch.device] = None) -> torch.Tensor:
    """Interleaved samples as the file holds them (uint8 view, or a float32 / int16 / ... array of shape ``[frames, channels]``) -> mono
    float32 ``[frames]`` on the device, channels averaged."""
    _capi.require_gpu()
    dev = device or torch.device("cuda", torch.cuda.current_device())
    if isinstance(raw, np.ndarray):
        a = np.ascontiguousarray(raw)
        t = torch.from_numpy(a if a.flags.writeable else a.copy())      # (views of a bytes object are read-only; torch wants to own writable memory)
    else:
        t = raw
    buf = t.contiguous().view(torch.uint8).reshape(-1)
    width = {8: 1, 16: 2, 24: 3, 32: 4, 0: 4, 64: 8}[sample_format]
    frames = buf.numel() // (width * channels)
    if frames <= 0:
        raise ValueError("no audio frames")
    d = buf.to(dev, non_blocking=True)
    out = torch.empty((frames,), dtype=torch.float32, device=dev)
    check(lib().avexhip_pcm_to_mono_f32(d.data_ptr(), sample_format, channels, frames, out.data_ptr(), torch.cuda.current_stream().cuda_stream), "pcm_to_mono_f32")
    return out


_RESAMPLERS: Dict[Tuple[int, int], Resampler] = {}


def load_audio(path_or_bytes: Union[str, bytes], target_sr: Optional[int] = 16000, device: Optional[torch.device] = None) -> Tuple[torch.Tensor, int]:
    """WAV or FLAC file -> ``(mono float32 [T] on the device, sample_rate)``; resampled to ``target_sr`` when it differs (``None``: keep)."""
    data = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray)) else open(path_or_bytes, "rb").read()
    if data[:4] == b"fLaC":
        fl = FlacStream(data)
        pcm = fl.decode(device, left_justify=True)          # int32 at 32-bit full scale: normalised like 32-bit PCM (soundfile's float32 reading)
        x, sr = to_device_mono(pcm, fl.channels, 32, device), fl.sample_rate
        fl.close()
    else:
        raw, sr, ch, code = parse_wav(data)
        x = to_device_mono(raw, ch, code, device)
    if target_sr is not None and sr != target_sr:
        key = (sr, int(target_sr))
        if key not in _RESAMPLERS:
            _RESAMPLERS[key] = Resampler(sr, int(target_sr))
        x, sr = _RESAMPLERS[key](x), int(target_sr)
    return x, sr
