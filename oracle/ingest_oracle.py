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


# resampy's published filters: (num_zeros, precision, rolloff, Kaiser beta)
RESAMPY_FILTERS = {"kaiser_best": (64, 9, 0.9475937167399596, 14.769656459379492),
                   "kaiser_fast": (16, 9, 0.85, 8.555504641634386)}


def resampy_filter(res_type: str = "kaiser_best"):
    """``resampy.filters.sinc_window(num_zeros, precision, window=kaiser(beta), rolloff)``: the right half of a Kaiser-windowed sinc,
    ``num_zeros * 2**precision + 1`` points (what resampy ships precomputed as ``data/kaiser_best.npz``).  UNPINNED (resampy absent)."""
    nz, prec, rolloff, beta = RESAMPY_FILTERS[res_type]
    num_bits = 2 ** prec
    n = num_bits * nz
    sinc_win = rolloff * np.sinc(rolloff * np.linspace(0, nz, num=n + 1, endpoint=True))
    taper = np.kaiser(2 * n + 1, beta)[n:]
    return taper * sinc_win, num_bits


def resample_librosa(x: np.ndarray, orig_sr: int, target_sr: int, res_type: str = "kaiser_best", scale: bool = True) -> np.ndarray:
    """``librosa.resample(y=x, orig_sr=, target_sr=, scale=, res_type="kaiser_best")`` for a mono fp32 signal, as the reference calls
    it (birdset_train_splits.py:190-196): resampy's ``resample_f`` loop restated sample by sample (two wings over the half-window
    table, weights linearly interpolated between table entries, window scaled by the ratio when decimating), ``int(T * ratio)``
    samples zero-padded to ``ceil(T * ratio)`` (librosa's ``fix_length``), divided by ``sqrt(ratio)`` when ``scale``.
    PARITY UNPINNED: neither librosa nor resampy is in the image; restated from their published source."""
    x = np.asarray(x, np.float32)
    if orig_sr == target_sr:
        return x
    ratio = float(target_sr) / orig_sr
    interp_win, num_table = resampy_filter(res_type)
    if ratio < 1:
        interp_win = interp_win * ratio
    interp_delta = np.zeros_like(interp_win)
    interp_delta[:-1] = np.diff(interp_win)
    sc = min(1.0, ratio)
    index_step = int(sc * num_table)
    nwin, n_orig = interp_win.shape[0], x.shape[0]
    n_res = int(n_orig * ratio)
    t_out = np.arange(n_res) * (1.0 / ratio)
    n = t_out.astype(np.int64)
    xd = x.astype(np.float64)
    y = np.zeros(n_res, np.float64)
    frac = sc * (t_out - n)
    for wing in (0, 1):
        if wing:
            frac = sc - frac
        index_frac = frac * num_table
        offset = index_frac.astype(np.int64)
        eta = index_frac - offset
        lim = (nwin - offset) // index_step
        cnt = np.minimum(n + 1, lim) if wing == 0 else np.minimum(n_orig - n - 1, lim)
        for i in range(int(cnt.max()) if cnt.size else 0):
            m = i < cnt
            k = offset[m] + i * index_step
            src = (n[m] - i) if wing == 0 else (n[m] + i + 1)
            y[m] += (interp_win[k] + eta[m] * interp_delta[k]) * xd[src]
    n_samples = int(math.ceil(n_orig * ratio))
    out = np.zeros(n_samples, np.float64)
    out[:n_res] = y
    if scale:
        out /= np.sqrt(ratio)
    return out.astype(np.float32)
