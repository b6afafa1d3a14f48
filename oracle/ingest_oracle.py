"""CPU oracle for the ingest row: channel averaging and torchaudio's sinc resampler, restated in NumPy.

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED: ``torchaudio`` (2.11.0 in the reference's uv.lock) is not installed and the reference
holds no golden for its resampler; this follows torchaudio's documented algorithm (``torchaudio.functional.resample`` ->
``_get_sinc_resample_kernel`` + strided convolution) as the reference calls it with defaults (augmentations.py:274-276).
"""
from __future__ import annotations

import math

import numpy as np


def pcm_to_mono(raw: np.ndarray, channels: int, fmt: int) -> np.ndarray:
    b = np.ascontiguousarray(raw).view(np.uint8).reshape(-1)
    if fmt == 16:
        x = b.view("<i2").astype(np.float32) / np.float32(32768.0)
    elif fmt == 32:
        x = b.view("<i4").astype(np.float32) / np.float32(2147483648.0)
    elif fmt == 24:
        t = b.reshape(-1, 3).astype(np.int32)
        v = t[:, 0] | (t[:, 1] << 8) | (t[:, 2] << 16)
        v = np.where(v >= 1 << 23, v - (1 << 24), v)
        x = v.astype(np.float32) / np.float32(8388608.0)
    elif fmt == 8:
        x = (b.astype(np.float32) - np.float32(128.0)) / np.float32(128.0)
    elif fmt == 64:
        x = b.view("<f8").astype(np.float32)
    else:
        x = b.view("<f4").copy()
    x = x.reshape(-1, channels)
    return x[:, 0] if channels == 1 else (x.sum(axis=1, dtype=np.float32) / np.float32(channels)).astype(np.float32)


def resample(x: np.ndarray, orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99, beta: float = 0.0) -> np.ndarray:
    """``torchaudio.functional.resample(x, orig_freq, new_freq)`` for ``x [..., T]`` (fp32 result, kernel built in fp64 like torchaudio)."""
    x = np.asarray(x, np.float32)
    if orig_freq == new_freq:
        return x
    g = math.gcd(orig_freq, new_freq)
    orig, new = orig_freq // g, new_freq // g
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    idx = np.arange(-width, width + orig, dtype=np.float64)[None, :] / orig
    t = (np.arange(0, -new, -1, dtype=np.float64)[:, None] / new + idx) * base
    t = np.clip(t, -lowpass_filter_width, lowpass_filter_width)
    if beta > 0:
        win = np.i0(beta * np.sqrt(1.0 - (t / lowpass_filter_width) ** 2)) / np.i0(beta)
    else:
        win = np.cos(t * math.pi / lowpass_filter_width / 2.0) ** 2
    t = t * math.pi
    kern = (np.where(t == 0, 1.0, np.sin(t) / np.where(t == 0, 1.0, t)) * win * (base / orig)).astype(np.float32)     # [new, taps]
    T = x.shape[-1]
    lead = x.shape[:-1]
    xp = np.pad(x.reshape(-1, T), ((0, 0), (width, width + orig)))
    n_fr = (xp.shape[1] - kern.shape[1]) // orig + 1
    win_idx = np.arange(n_fr)[:, None] * orig + np.arange(kern.shape[1])[None, :]
    out = np.einsum("bfk,pk->bfp", xp[:, win_idx].astype(np.float64), kern.astype(np.float64)).reshape(xp.shape[0], -1)
    tgt = math.ceil(new * T / orig)
    return out[:, :tgt].astype(np.float32).reshape(*lead, tgt)
