"""Thin torch-tensor front ends over the C ABI (device pointers + current stream in, tensors out).

PyTorch is plumbing here: it owns HBM allocations and streams; all arithmetic happens in
libavexhip.so.  Every function raises ``AvexHipError`` if the library or a GPU is missing.
"""
from __future__ import annotations

import ctypes as C
import logging
import math
import os
import warnings
from typing import Dict, List, Mapping, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _capi
from ._capi import AvexHipError, BeatsConfig, FbankConfig, GemmArgs, MelspecConfig, Tensor, check, dtype_code, lib

logger = logging.getLogger(__name__)

F32_EPS = 1.1920929e-07


def _stream() -> int:
    return int(torch.cuda.current_stream().cuda_stream)


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else int(t.data_ptr())


def half_torch_dtype(code: int) -> torch.dtype:
    return torch.float16 if code == _capi.F16 else torch.bfloat16


def _need_cuda(*ts: Optional[torch.Tensor]) -> None:
    _capi.require_gpu()
    for t in ts:
        if t is not None and not t.is_cuda:
            raise AvexHipError("avex_amd kernels need CUDA(HIP) tensors; got a CPU tensor (no CPU fallback)")


# ---------------------------------------------------------------------------------------
# Frontend
# ---------------------------------------------------------------------------------------
def povey_window(win_length: int = 400) -> np.ndarray:
    """``hann(win, periodic=False) ** 0.85`` (reference: beats.py:75), fp32."""
    return np.power(hann_window(win_length), np.float32(0.85)).astype(np.float32)


def hann_window(win_length: int = 400) -> np.ndarray:
    """``torch.hann_window(win, periodic=False)`` op for op in fp32 (arange * 2pi/(N-1) -> cos -> * -0.5 + 0.5)."""
    n = np.arange(win_length, dtype=np.float32)
    c = np.cos(n * np.float32(math.pi * 2 / (win_length - 1))).astype(np.float32)
    return (c * np.float32(-0.5) + np.float32(0.5)).astype(np.float32)


def kaldi_mel_filterbank(n_mels: int = 128, n_fft: int = 512, sample_rate: float = 16000.0,
                         low_freq: float = 20.0, high_freq: float = 0.0) -> np.ndarray:
    """Triangular kaldi mel bank ``[n_fft//2+1, n_mels]`` (reference: beats.py:82-118), fp32."""
    if high_freq <= 0.0:
        high_freq = sample_rate / 2.0 + high_freq
    nb = n_fft // 2
    mel_low = 1127.0 * math.log(1.0 + low_freq / 700.0)
    mel_high = 1127.0 * math.log(1.0 + high_freq / 700.0)
    delta = np.float32((mel_high - mel_low) / (n_mels + 1))
    idx = np.arange(n_mels, dtype=np.float32)[:, None]
    left = np.float32(mel_low) + idx * delta
    center = np.float32(mel_low) + (idx + np.float32(1.0)) * delta
    right = np.float32(mel_low) + (idx + np.float32(2.0)) * delta
    freqs = np.float32(sample_rate / n_fft) * np.arange(nb, dtype=np.float32)
    mel = (np.float32(1127.0) * np.log(np.float32(1.0) + freqs / np.float32(700.0)))[None, :].astype(np.float32)
    fb = np.maximum(np.float32(0.0), np.minimum((mel - left) / (center - left), (right - mel) / (right - center)))
    fb = np.pad(fb.astype(np.float32), ((0, 0), (0, 1)))
    return np.ascontiguousarray(fb.T)


class FbankPlan:
    """Fused log-mel frontend (reference: ``_BatchedFbank`` + ``BEATs.preprocess``)."""

    def __init__(self, *, win_length: int = 400, hop_length: int = 160, n_mels: int = 128,
                 input_scale: float = 32768.0, preemph: float = 0.97, remove_dc: bool = True,
                 log_floor: float = F32_EPS, norm_mean: float = 0.0, norm_div: float = 1.0,
                 window: Optional[np.ndarray] = None, mel_fb: Optional[np.ndarray] = None) -> None:
        _capi.require_gpu()
        self.win_length, self.hop_length, self.n_mels = win_length, hop_length, n_mels
        window = povey_window(win_length) if window is None else np.ascontiguousarray(window, np.float32)
        mel_fb = kaldi_mel_filterbank(n_mels) if mel_fb is None else np.ascontiguousarray(mel_fb, np.float32)
        if window.shape != (win_length,) or mel_fb.shape != (257, n_mels):
            raise ValueError(f"window must be [{win_length}] and mel_fb [257,{n_mels}]")
        cfg = FbankConfig(win_length, hop_length, n_mels, input_scale, preemph, int(remove_dc), log_floor,
                          norm_mean, norm_div)
        self._h = lib().avexhip_fbank_plan_create(C.byref(cfg), window.ctypes.data, mel_fb.ctypes.data)
        if not self._h:
            raise AvexHipError(f"fbank_plan_create failed: {_capi.last_error()}")

    def num_frames(self, T: int) -> int:
        return int(lib().avexhip_fbank_num_frames(self._h, T))

    def __call__(self, wav: torch.Tensor) -> torch.Tensor:
        _need_cuda(wav)
        if wav.dim() != 2 or wav.dtype != torch.float32:
            raise ValueError("wav must be a [B, T] float32 tensor")
        if wav.stride(1) != 1:
            wav = wav.contiguous()
        B, T = wav.shape
        frames = self.num_frames(T)
        out = torch.empty((B, frames, self.n_mels), dtype=torch.float32, device=wav.device)
        if B == 0 or frames == 0:
            return out
        check(lib().avexhip_fbank_forward(self._h, _ptr(wav), B, T, wav.stride(0), _ptr(out), _stream()), "fbank_forward")
        return out

    def padded(self, wav: torch.Tensor, out_frames: int, remove_clip_mean: bool = False) -> torch.Tensor:
        """``[B, T]`` -> ``[B, out_frames, n_mels]``: rows past the last frame hold the normalised zero padding, frames past
        ``out_frames`` are cut; ``remove_clip_mean`` subtracts each clip's mean at load (EAT, eat/audio_processor.py:107)."""
        _need_cuda(wav)
        if wav.dim() != 2 or wav.dtype != torch.float32:
            raise ValueError("wav must be a [B, T] float32 tensor")
        if out_frames <= 0:
            raise ValueError("out_frames must be positive")
        if wav.stride(1) != 1:
            wav = wav.contiguous()
        B, T = wav.shape
        out = torch.empty((B, out_frames, self.n_mels), dtype=torch.float32, device=wav.device)
        if B == 0:
            return out
        off = None
        if remove_clip_mean:
            off = torch.empty((B,), dtype=torch.float32, device=wav.device)
            check(lib().avexhip_clip_mean(_ptr(wav), B, T, wav.stride(0), _ptr(off), _stream()), "clip_mean")
        check(lib().avexhip_fbank_forward_padded(self._h, _ptr(wav), B, T, wav.stride(0), _ptr(off) if off is not None else None,
                                                 out_frames, _ptr(out), _stream()), "fbank_forward_padded")
        return out

    def patches(self, wav: torch.Tensor, out_frames: int = 0, patch: int = 16, remove_clip_mean: bool = False, dtype="f16") -> torch.Tensor:
        """``[B, T]`` -> half ``[B * (out_frames/patch) * (n_mels/patch), patch*patch]``: the log-mel image cut into the rows a
        patch-embedding GEMM reads (token = t * n_mels/patch + f), straight from the filterbank kernel."""
        _need_cuda(wav)
        if wav.dim() != 2 or wav.dtype != torch.float32:
            raise ValueError("wav must be a [B, T] float32 tensor")
        if wav.stride(1) != 1:
            wav = wav.contiguous()
        B, T = wav.shape
        code = dtype_code(dtype)
        frames = out_frames if out_frames > 0 else self.num_frames(T)
        nt, nf = frames // patch, self.n_mels // patch
        out = torch.empty((B * nt * nf, patch * patch), dtype=half_torch_dtype(code), device=wav.device)
        off = None
        if remove_clip_mean:
            off = torch.empty((B,), dtype=torch.float32, device=wav.device)
            check(lib().avexhip_clip_mean(_ptr(wav), B, T, wav.stride(0), _ptr(off), _stream()), "clip_mean")
        check(lib().avexhip_fbank_forward_patches(self._h, _ptr(wav), B, T, wav.stride(0), _ptr(off), out_frames, patch, _ptr(out), code,
                                                  _stream()), "fbank_forward_patches")
        return out

    def __del__(self) -> None:
        try:
            if getattr(self, "_h", None):
                lib().avexhip_fbank_plan_destroy(self._h)
                self._h = None
        except Exception:  # noqa: BLE001
            pass


# ---------------------------------------------------------------------------------------
# Building blocks (used by the parity tests; the encoder handle calls the same launchers in C++)
# ---------------------------------------------------------------------------------------
def stft_window(kind: str, win_length: int) -> np.ndarray:
    """``torch.hann_window(N)`` / ``torch.hamming_window(N)`` (periodic), fp32 (reference: audio_utils.py:159-164)."""
    n = np.arange(win_length, dtype=np.float64)
    if kind == "hann":
        return (0.5 - 0.5 * np.cos(2.0 * math.pi * n / win_length)).astype(np.float32)
    if kind == "hamming":
        return (0.54 - 0.46 * np.cos(2.0 * math.pi * n / win_length)).astype(np.float32)
    raise ValueError(f"Unknown window type: {kind}")


def htk_mel_filterbank(n_freqs: int, n_mels: int, sample_rate: int, f_min: float = 0.0, f_max: Optional[float] = None) -> np.ndarray:
    """``torchaudio.functional.melscale_fbanks(n_freqs, f_min, f_max, n_mels, sample_rate, norm=None, mel_scale="htk")``
    = the ``fb`` buffer of ``torchaudio.transforms.MelScale`` as the reference builds it (audio_utils.py:97-101): ``[n_freqs, n_mels]``."""
    f_max = float(sample_rate // 2) if f_max is None else f_max
    all_freqs = np.linspace(0.0, sample_rate // 2, n_freqs)
    hz2mel = lambda f: 2595.0 * np.log10(1.0 + f / 700.0)
    mel2hz = lambda m: 700.0 * (10.0 ** (m / 2595.0) - 1.0)
    f_pts = mel2hz(np.linspace(hz2mel(f_min), hz2mel(f_max), n_mels + 2))
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return np.maximum(0.0, np.minimum(down, up)).astype(np.float32)


class MelspecPlan:
    """STFT power / mel spectrogram of the reference's ``AudioProcessor`` (audio_utils.py:77-172) on the GPU."""

    def __init__(self, *, n_fft: int, hop_length: int, win_length: Optional[int] = None, window: str = "hann", n_mels: int = 128,
                 sample_rate: int = 16000, mel: bool = True, center: bool = True, normalize: bool = True) -> None:
        _capi.require_gpu()
        win_length = win_length or n_fft
        self.n_fft, self.hop_length, self.n_bins = n_fft, hop_length, (n_mels if mel else n_fft // 2 + 1)
        w = stft_window(window, win_length)
        fb = htk_mel_filterbank(n_fft // 2 + 1, n_mels, sample_rate) if mel else None
        cfg = MelspecConfig(n_fft, hop_length, win_length, n_mels if mel else 0, int(center), int(normalize))
        self._normalize = normalize
        self._h = lib().avexhip_melspec_plan_create(C.byref(cfg), w.ctypes.data, fb.ctypes.data if fb is not None else None)
        if not self._h:
            raise AvexHipError(f"melspec_plan_create failed: {_capi.last_error()}")

    def num_frames(self, T: int) -> int:
        return int(lib().avexhip_melspec_num_frames(self._h, T))

    def __call__(self, wav: torch.Tensor) -> torch.Tensor:
        _need_cuda(wav)
        if wav.dim() != 2 or wav.dtype != torch.float32:
            raise ValueError("wav must be a [B, T] float32 tensor")
        wav = wav.contiguous()
        B, T = wav.shape
        frames = self.num_frames(T)
        out = torch.empty((B, self.n_bins, frames), dtype=torch.float32, device=wav.device)
        mm = torch.empty((B, 2), dtype=torch.int32, device=wav.device) if self._normalize else None
        check(lib().avexhip_melspec_forward(self._h, _ptr(wav), B, T, wav.stride(0), _ptr(out), _ptr(mm), _stream()), "melspec_forward")
        return out

    def __del__(self) -> None:
        try:
            if getattr(self, "_h", None):
                lib().avexhip_melspec_plan_destroy(self._h)
                self._h = None
        except Exception:  # noqa: BLE001
            pass


def effnet_stem(img: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, dtype="f16", want_raw: bool = False):
    """``[B, H, W]`` fp32 image -> NHWC half ``[B, Ho, Wo, Cp]`` (3x3 stride-2 stem with folded BatchNorm + SiLU); ``w [9, Cp]``."""
    _need_cuda(img, w, bias)
    code = dtype_code(dtype)
    img = img.contiguous()
    B, H, W = img.shape
    Cp = w.shape[1]
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    out = torch.empty((B, Ho, Wo, Cp), dtype=half_torch_dtype(code), device=img.device)
    raw = torch.empty((B, Ho, Wo, Cp), dtype=torch.float32, device=img.device) if want_raw else None
    check(lib().avexhip_effnet_stem(_ptr(img), B, H, W, _ptr(w.contiguous()), _ptr(bias), Cp, _ptr(out), _ptr(raw), code, _stream()), "effnet_stem")
    return (out, raw) if want_raw else out


def effnet_dwconv(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, k: int, stride: int, want_pool: bool = True):
    """Depthwise ``k x k`` conv (+ folded BN + SiLU) on NHWC half ``[B, H, W, Cp]``; returns (out, channel sums ``[B, Cp]`` fp32)."""
    _need_cuda(x, w, bias)
    code = _capi.F16 if x.dtype == torch.float16 else _capi.BF16
    x = x.contiguous()
    B, H, W, Cp = x.shape
    pad = (k - 1) // 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    out = torch.empty((B, Ho, Wo, Cp), dtype=x.dtype, device=x.device)
    pool = torch.empty((B, Cp), dtype=torch.float32, device=x.device) if want_pool else None
    nbytes = int(lib().avexhip_effnet_dwconv_part_bytes(B, H, W, Cp, k, stride)) if want_pool else 0
    part = torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=x.device) if want_pool else None     # one row of partial sums per workgroup
    check(lib().avexhip_effnet_dwconv(_ptr(x), B, H, W, Cp, k, stride, _ptr(w.contiguous()), _ptr(bias), _ptr(out), _ptr(pool), _ptr(part), nbytes, code, _stream()), "effnet_dwconv")
    return out, pool


def effnet_se(x: torch.Tensor, pool: torch.Tensor, C: int, w1: torch.Tensor, b1: torch.Tensor, w2: torch.Tensor, b2: torch.Tensor) -> torch.Tensor:
    """Squeeze-excitation on NHWC half ``x`` IN PLACE from its channel sums; returns the ``[B, Cp]`` scales."""
    _need_cuda(x, pool, w1, w2)
    code = _capi.F16 if x.dtype == torch.float16 else _capi.BF16
    B, H, W, Cp = x.shape
    scale = torch.empty((B, Cp), dtype=torch.float32, device=x.device)
    check(lib().avexhip_effnet_se(_ptr(pool), B, H * W, C, Cp, w1.shape[0], _ptr(w1.contiguous()), _ptr(b1), _ptr(w2.contiguous()), _ptr(b2),
                                  _ptr(scale), _ptr(x), code, _stream()), "effnet_se")
    return scale


def wavconv0(wav: torch.Tensor, w: torch.Tensor, gn_w: torch.Tensor, gn_b: torch.Tensor, frames_pad: int, slack_rows: int = 8,
             eps: float = 1e-5, dtype="f16") -> torch.Tensor:
    """wav2vec2 conv layer 0 (Conv1d(1,512,10,5) + GroupNorm over time + GELU) -> half ``[B * frames_pad + slack_rows, 512]``
    (``[clip][frame][channel]`` rows; rows past a clip's last frame are zero, the slack rows are uninitialised)."""
    _need_cuda(wav, w, gn_w, gn_b)
    code = dtype_code(dtype)
    wav = wav.contiguous()
    B, T = wav.shape
    out = torch.empty((B * frames_pad + slack_rows, 512), dtype=half_torch_dtype(code), device=wav.device)
    stats = torch.empty((int(lib().avexhip_wavconv0_stats_floats(B, T)),), dtype=torch.float32, device=wav.device)
    check(lib().avexhip_wavconv0(_ptr(wav), B, T, wav.stride(0), _ptr(w.contiguous()), _ptr(gn_w), _ptr(gn_b), eps, _ptr(stats),
                                 _ptr(out), frames_pad, code, _stream()), "wavconv0")
    return out


PROBE_ACT = {None: 0, "none": 0, "relu": 1, "gelu": 2, "tanh": 3}


def seq_interp_linear(x: torch.Tensor, t_out: int) -> torch.Tensor:
    """``[B, Tin, C]`` fp32 -> ``[B, t_out, C]``: ``F.interpolate(mode="linear", align_corners=False)`` along the sequence."""
    _need_cuda(x)
    x = x.float().contiguous()
    B, Tin, Cc = x.shape
    out = torch.empty((B, int(t_out), Cc), dtype=torch.float32, device=x.device)
    check(lib().avexhip_seq_interp_linear(_ptr(x), B, Tin, Cc, int(t_out), _ptr(out), _stream()), "seq_interp_linear")
    return out


def layer_mix(taps: Sequence[torch.Tensor], layer_weights: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``sum_l softmax(layer_weights)_l * taps[l]`` (weights of 1.0 without ``layer_weights``), the reference's
    ``_BaseProbe._sum`` (base_probes.py:197-206), in one pass over the taps."""
    import ctypes as C
    taps = [t.contiguous() for t in taps]
    _need_cuda(*taps)
    if not 1 <= len(taps) <= 16 or any(t.shape != taps[0].shape or t.dtype != torch.float32 for t in taps):
        raise ValueError("layer_mix: need 1..16 fp32 taps of one shape")
    if layer_weights is not None:
        _need_cuda(layer_weights)
        if layer_weights.numel() != len(taps):
            raise ValueError("layer_mix: one weight per tap")
        layer_weights = layer_weights.contiguous().float()
    out = torch.empty_like(taps[0])
    arr = (C.c_void_p * len(taps))(*[t.data_ptr() for t in taps])
    check(lib().avexhip_layer_mix(arr, len(taps), _ptr(layer_weights), out.numel(), _ptr(out), _stream()), "layer_mix")
    return out


def dense_f32(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, act: Optional[str] = None,
              resid: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fp32 ``act(x @ w.T + bias) (+ resid)`` over the last dim of ``x`` (nn.Linear semantics) on the fp32 matrix core."""
    _need_cuda(x, w)
    lead = x.shape[:-1]
    x2 = x.reshape(-1, x.shape[-1]).contiguous()
    w = w.contiguous()
    M, K = x2.shape
    N = w.shape[0]
    if w.shape[1] != K or x2.dtype != torch.float32 or w.dtype != torch.float32:
        raise ValueError(f"dense_f32: x [..., {K}] fp32 against w {tuple(w.shape)}")
    r2 = None
    if resid is not None:
        r2 = resid.reshape(-1, N).contiguous()
        if r2.shape[0] != M:
            raise ValueError("dense_f32: residual shape")
    out = torch.empty((M, N), dtype=torch.float32, device=x.device)
    check(lib().avexhip_dense_f32(_ptr(x2), K, _ptr(w), K, _ptr(bias.contiguous() if bias is not None else None), _ptr(r2), N, M, N, K,
                                  PROBE_ACT[act], _ptr(out), N, _stream()), "dense_f32")
    return out.reshape(*lead, N)


def lstm_layer(xg: torch.Tensor, w_hh_t: torch.Tensor, out: torch.Tensor, col: int = 0, reverse: bool = False) -> None:
    """One direction of one ``nn.LSTM`` layer: ``xg [B, T, 4H]`` (input half of the gates, biases included), ``w_hh_t [H, 4H]`` (W_hh
    transposed) -> ``out[:, :, col:col + H]`` (``out [B, T, W]`` contiguous, ``W >= col + H``)."""
    _need_cuda(xg, w_hh_t, out)
    B, T, G = xg.shape
    H = G // 4
    if w_hh_t.shape != (H, 4 * H) or out.shape[:2] != (B, T) or out.shape[2] < col + H or not (xg.is_contiguous() and w_hh_t.is_contiguous() and out.is_contiguous()):
        raise ValueError("lstm_layer: shapes")
    if not all(t.dtype == torch.float32 for t in (xg, w_hh_t, out)):      # (the column offset below is in 4-byte elements)
        raise ValueError("lstm_layer: float32 tensors only")
    check(lib().avexhip_lstm_layer(_ptr(xg), _ptr(w_hh_t), B, T, H, int(bool(reverse)), int(out.data_ptr()) + 4 * col, out.shape[2], _stream()), "lstm_layer")


def lstm_layer_pair(xg: torch.Tensor, w_hh_t: torch.Tensor, xg_rev: torch.Tensor, w_hh_t_rev: torch.Tensor, out: torch.Tensor) -> None:
    """Both directions of a bidirectional ``nn.LSTM`` layer in one launch: forward -> ``out[:, :, :H]``, backward -> ``out[:, :, H:2H]``
    (``out [B, T, 2H]`` contiguous).  Same arithmetic as two :func:`lstm_layer` calls."""
    _need_cuda(xg, w_hh_t, xg_rev, w_hh_t_rev, out)
    B, T, G = xg.shape
    H = G // 4
    ok = (xg_rev.shape == xg.shape and w_hh_t.shape == (H, 4 * H) and w_hh_t_rev.shape == (H, 4 * H) and out.shape == (B, T, 2 * H)
          and all(t.is_contiguous() and t.dtype == torch.float32 for t in (xg, xg_rev, w_hh_t, w_hh_t_rev, out)))
    if not ok:
        raise ValueError("lstm_layer_pair: shapes")
    check(lib().avexhip_lstm_layer_pair(_ptr(xg), _ptr(w_hh_t), _ptr(xg_rev), _ptr(w_hh_t_rev), B, T, H, _ptr(out), int(out.data_ptr()) + 4 * H,
                                        2 * H, _stream()), "lstm_layer_pair")


def mha_f32(qkv: torch.Tensor, num_heads: int, key_pad: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Attention core of ``nn.MultiheadAttention`` (eval, self attention): ``qkv [B, T, 3E]`` -> ``[B, T, E]``."""
    _need_cuda(qkv)
    qkv = qkv.contiguous()
    B, T, E3 = qkv.shape
    E = E3 // 3
    kp = None
    if key_pad is not None:
        kp = key_pad.to(device=qkv.device, dtype=torch.uint8).contiguous()
        if kp.shape != (B, T):
            raise ValueError("mha_f32: key_pad must be [B, T]")
    out = torch.empty((B, T, E), dtype=torch.float32, device=qkv.device)
    check(lib().avexhip_mha_f32(_ptr(qkv), B, T, E, num_heads, _ptr(kp), _ptr(out), _stream()), "mha_f32")
    return out


def to_half(x: torch.Tensor, dtype="f16") -> torch.Tensor:
    _need_cuda(x)
    code = dtype_code(dtype)
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=half_torch_dtype(code), device=x.device)
    check(lib().avexhip_cast_f32_to_half(_ptr(x), _ptr(out), x.numel(), code, _stream()), "cast_f32_to_half")
    return out


def to_f32(x: torch.Tensor) -> torch.Tensor:
    _need_cuda(x)
    code = _capi.F16 if x.dtype == torch.float16 else _capi.BF16
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    check(lib().avexhip_cast_half_to_f32(_ptr(x), _ptr(out), x.numel(), code, _stream()), "cast_half_to_f32")
    return out


def gemm(a: torch.Tensor, w: torch.Tensor, *, bias: Optional[torch.Tensor] = None,
         resid: Optional[torch.Tensor] = None, resid_half: Optional[torch.Tensor] = None, alpha: float = 1.0,
         gelu: bool = False, silu: bool = False,
         out_f32: bool = True, out_half: bool = False, out_raw: bool = False, variant: int = 0,
         ln_rows: Optional[torch.Tensor] = None, ln_s: Optional[torch.Tensor] = None,
         lnr_y: Optional[torch.Tensor] = None, lnr_rows: Optional[torch.Tensor] = None,
         lnr_gamma: Optional[torch.Tensor] = None, lnr_beta: Optional[torch.Tensor] = None, stats_out: bool = False,
         lda: Optional[int] = None, rows: Optional[int] = None, kdim: Optional[int] = None, slack_rows: int = 0,
         overflow: Optional[torch.Tensor] = None, pool_rows: int = 0, pool_mode: str = "mean", splitk: bool = False,
         rows_eps: Optional[float] = None) -> Dict[str, torch.Tensor]:
    """``epi(a @ w.T)`` with ``a [M,K]`` and ``w [N,K]`` half tensors (see avexhip_gemm).  ``ln_rows``/``ln_s`` fold a
    LayerNorm of the A rows into the epilogue, ``lnr_*`` apply LayerNorm(lnr_y) as the residual (the ``*_rows`` tensors come from
    :func:`ln_rowstats`), ``stats_out`` returns the per-row partial statistics ``[M, N/64, 2]`` of the output under ``"stats"``;
    ``overflow`` is an optional ``uint32``/``int32`` device scalar the f16 range alarm adds to (include/avexhip.h); ``pool_rows`` = T
    treats the rows as clips of T rows and returns under ``"pooled"`` the per-clip mean of the raw output (bias added, before residual /
    activation) without materialising it (``pool_part`` + ``avexhip_pool_reduce``); ``pool_mode`` "max" / "cls_token" return the per-clip
    maximum / first row instead; ``rows_eps`` returns under ``"rows"`` the finished row statistics ``[M (+1 if odd), 2]`` = (rstd, -mean rstd)
    of the output (what :func:`ln_rowstats` makes of ``"stats"``, same bits; the full-row kernel -- ``variant=8``, N = 768 -- writes them
    from its epilogue)."""
    _need_cuda(a, w)
    if a.dtype != w.dtype or a.dtype not in (torch.float16, torch.bfloat16):
        raise ValueError("a and w must both be float16 or bfloat16")
    code = _capi.F16 if a.dtype == torch.float16 else _capi.BF16
    a, w = a.contiguous(), w.contiguous()
    if lda is not None:
        # strided, possibly overlapping rows of a flat buffer: row m = a.flatten()[m * lda : m * lda + kdim]
        # (a Conv1d over [frame][channel] activations is this GEMM with lda = stride * C and kdim = kernel * C)
        M, K = int(rows), int(kdim)
        if a.numel() < (M - 1) * lda + K:
            raise ValueError("gemm: buffer too small for the strided view")
    else:
        M, K = a.shape
        lda = K
    N = w.shape[0]
    res: Dict[str, torch.Tensor] = {}
    args = GemmArgs()
    args.A, args.lda, args.W, args.ldw = _ptr(a), lda, _ptr(w), K
    args.M, args.N, args.K = M, N, K
    args.bias = _ptr(bias)
    if resid is not None:
        resid = resid.contiguous()
        args.resid, args.ldr = _ptr(resid), N
    elif resid_half is not None:
        resid_half = resid_half.contiguous()
        args.resid_half, args.ldrh = _ptr(resid_half), N
    args.alpha, args.gelu, args.variant = alpha, (2 if silu else int(gelu)), variant
    if out_f32:
        res["f32"] = torch.empty((M, N), dtype=torch.float32, device=a.device)
        args.out_f32, args.ldo = _ptr(res["f32"]), N
    if out_half:
        res["half"] = torch.empty((M + slack_rows, N), dtype=a.dtype, device=a.device)   # slack rows (uninitialised) for strided readers
        args.out_half, args.ldh = _ptr(res["half"]), N
    if out_raw:
        res["raw"] = torch.empty((M, N), dtype=torch.float32, device=a.device)
        args.out_raw, args.ldraw = _ptr(res["raw"]), N
    if ln_rows is not None:
        ln_rows = _padded_rows(ln_rows, M)
        args.ln_rows, args.ln_s = _ptr(ln_rows), _ptr(ln_s)
    if lnr_y is not None:
        lnr_y, lnr_rows = lnr_y.contiguous(), lnr_rows.contiguous()
        args.lnr_y, args.ldy, args.lnr_rows = _ptr(lnr_y), N, _ptr(lnr_rows)
        args.lnr_gamma, args.lnr_beta = _ptr(lnr_gamma), _ptr(lnr_beta)
    if overflow is not None:
        if overflow.numel() != 1 or overflow.element_size() != 4 or not overflow.is_cuda:
            raise ValueError("overflow must be a 4-byte device scalar")
        args.overflow_count = _ptr(overflow)
    sk_ws = None
    if splitk:      # lend the 128-tile kernel scratch for split-K (used when the product is few tiles of a long contraction)
        sk_ws = torch.empty((8 * M * N,), dtype=torch.float32, device=a.device)
        args.splitk_ws, args.splitk_bytes = _ptr(sk_ws), sk_ws.numel() * 4
    part = None
    if pool_rows:
        if M % pool_rows:
            raise ValueError("pool_rows must divide the number of rows")
        mode = {"mean": 0, "max": 1, "cls_token": 2}[pool_mode]
        if mode == 2:
            part = torch.full((M // pool_rows, N), float("nan"), dtype=torch.float32, device=a.device)
        else:
            part = torch.full(((M + 63) // 64, 2, N), float("nan"), dtype=torch.float32, device=a.device)      # NaN: an unwritten slot that is read shows
        args.pool_part, args.pool_rows, args.pool_mode = _ptr(part), int(pool_rows), mode
    if stats_out:
        res["stats"] = torch.zeros((M, N // 64, 2), dtype=torch.float32, device=a.device)
        args.stats_out = _ptr(res["stats"])
    if rows_eps is not None:
        res["rows"] = torch.zeros((M + (M & 1), 2), dtype=torch.float32, device=a.device)
        args.rows_out, args.rows_eps = _ptr(res["rows"]), float(rows_eps)
        if not stats_out:      # scratch for the kernels that cannot finish the statistics themselves
            scratch = torch.empty((M, N // 64, 2), dtype=torch.float32, device=a.device)
            args.stats_out = _ptr(scratch)
    check(lib().avexhip_gemm(C.byref(args), code, _stream()), "gemm")
    if part is not None and args.pool_mode == 2:
        res["pooled"] = part
    elif part is not None:
        res["pooled"] = torch.empty((M // pool_rows, N), dtype=torch.float32, device=a.device)
        check(lib().avexhip_pool_reduce_mode(_ptr(part), M // pool_rows, int(pool_rows), N, _ptr(res["pooled"]), N, int(args.pool_mode), _stream()), "pool_reduce")
    return res


def _padded_rows(rows: torch.Tensor, M: int) -> torch.Tensor:
    """(rstd, shift) pairs readable up to an even number of rows (the kernel fetches them two rows at a time)."""
    rows = rows.contiguous()
    if rows.shape[0] >= M + (M & 1):
        return rows
    out = torch.zeros((M + (M & 1), 2), dtype=torch.float32, device=rows.device)
    out[:rows.shape[0]] = rows
    return out


def ln_rowstats(stats: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """Partial statistics ``[M, nseg, 2]`` (``gemm(..., stats_out=True)["stats"]``) -> ``[M (+1 if odd), 2]`` = (rstd, -mean * rstd)."""
    _need_cuda(stats)
    stats = stats.contiguous()
    M, nseg, _ = stats.shape
    rows = torch.zeros((M + (M & 1), 2), dtype=torch.float32, device=stats.device)
    check(lib().avexhip_ln_rowstats(_ptr(stats), M, nseg, eps, _ptr(rows), _stream()), "ln_rowstats")
    return rows


def layernorm(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float = 1e-5, half_dtype="f16",
              want_f32: bool = True, want_half: bool = True) -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor]]:
    """LayerNorm of an fp32 or half ``[M, C]`` tensor -> (fp32 copy, half copy); either may be skipped."""
    _need_cuda(x, weight, bias)
    x = x.contiguous()
    M, Cc = x.shape
    if x.dtype == torch.float32:
        code = dtype_code(half_dtype)
        pin, pinh = _ptr(x), None
    else:
        code = _capi.F16 if x.dtype == torch.float16 else _capi.BF16
        pin, pinh = None, _ptr(x)
    o32 = torch.empty((M, Cc), dtype=torch.float32, device=x.device) if want_f32 else None
    oh = torch.empty((M, Cc), dtype=half_torch_dtype(code), device=x.device) if want_half else None
    check(lib().avexhip_layernorm(pin, pinh, Cc, _ptr(weight), _ptr(bias), eps, M, Cc, _ptr(o32), Cc, _ptr(oh), Cc,
                                  code, _stream()), "layernorm")
    return o32, oh


def attention(qkv: torch.Tensor, B: int, T: int, H: int, bias_tab: Optional[torch.Tensor],
              grep_w: Optional[torch.Tensor], grep_b: Optional[torch.Tensor], grep_a: Optional[torch.Tensor],
              key_pad: Optional[torch.Tensor] = None) -> torch.Tensor:
    _need_cuda(qkv)
    code = _capi.F16 if qkv.dtype == torch.float16 else _capi.BF16
    qkv = qkv.contiguous()
    out = torch.empty((B * T, H * 64), dtype=qkv.dtype, device=qkv.device)
    check(lib().avexhip_attention(_ptr(qkv), B, T, H, _ptr(bias_tab), _ptr(grep_w), _ptr(grep_b), _ptr(grep_a),
                                  _ptr(key_pad), _ptr(out), code, _stream()), "attention")
    return out


def attention_hd(qkv: torch.Tensor, B: int, T: int, H: int, head_dim: int, key_pad: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Plain multi-head self-attention on half ``[B*T, 3*H*head_dim]`` rows (q | k | v) for head widths 32 / 64 / 96 / 128: what
    ``torch.nn.MultiheadAttention`` computes inside the reference's sequence probes.  ``key_pad``: ``[B, T]`` uint8, 1 = padded key."""
    _need_cuda(qkv)
    code = _capi.F16 if qkv.dtype == torch.float16 else _capi.BF16
    qkv = qkv.contiguous()
    out = torch.empty((B * T, H * head_dim), dtype=qkv.dtype, device=qkv.device)
    check(lib().avexhip_attention_hd(_ptr(qkv), B, T, H, head_dim, _ptr(key_pad), _ptr(out), code, _stream()), "attention_hd")
    return out


def posconv_pack(g: torch.Tensor, v: torch.Tensor, groups: int, half_dtype="f16") -> torch.Tensor:
    _need_cuda(g, v)
    code = dtype_code(half_dtype)
    E, cg, K = v.shape
    out = torch.empty((E * cg * K,), dtype=half_torch_dtype(code), device=v.device)
    check(lib().avexhip_posconv_pack(_ptr(g.contiguous()), _ptr(v.contiguous()), E, groups, K, _ptr(out), code,
                                     _stream()), "posconv_pack")
    return out


def posconv(x_half: torch.Tensor, x_f32: Optional[torch.Tensor], w_packed: torch.Tensor, bias: torch.Tensor,
            groups: int, K: int = 128, half_out: bool = False) -> torch.Tensor:
    """``x + gelu(conv(x_half) + bias)``; the residual is ``x_f32`` when given, else ``x_half``."""
    _need_cuda(x_half, w_packed, bias)
    code = _capi.F16 if x_half.dtype == torch.float16 else _capi.BF16
    B, T, E = x_half.shape
    x_half = x_half.contiguous()
    out = torch.empty((B, T, E), dtype=x_half.dtype if half_out else torch.float32, device=x_half.device)
    check(lib().avexhip_posconv(_ptr(x_half), None if x_f32 is None else _ptr(x_f32.contiguous()), _ptr(w_packed),
                                _ptr(bias), B, T, E, groups, K, None if half_out else _ptr(out),
                                _ptr(out) if half_out else None, code, _stream()), "posconv")
    return out


def token_embed_ln(patches: torch.Tensor, pos: torch.Tensor, cls: torch.Tensor, ln_w: torch.Tensor, ln_b: torch.Tensor, eps: float,
                   B: int, want_f32: bool = False) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """Class token + (patch + position) rows, LayerNorm'ed: half ``[B * n, C]`` -> (half, fp32 or None) ``[B * (n + 1), C]``."""
    _need_cuda(patches, pos, cls, ln_w, ln_b)
    code = _capi.F16 if patches.dtype == torch.float16 else _capi.BF16
    M, Cc = patches.shape
    n = M // B
    outh = torch.empty((B * (n + 1), Cc), dtype=patches.dtype, device=patches.device)
    outf = torch.empty((B * (n + 1), Cc), dtype=torch.float32, device=patches.device) if want_f32 else None
    check(lib().avexhip_token_embed_ln(_ptr(patches.contiguous()), _ptr(pos.contiguous()), _ptr(cls.contiguous()), _ptr(ln_w), _ptr(ln_b), eps,
                                       B, n, Cc, _ptr(outh), _ptr(outf), code, _stream()), "token_embed_ln")
    return outh, outf


def mean_pool(x: torch.Tensor) -> torch.Tensor:
    _need_cuda(x)
    B, T, Cc = x.shape
    out = torch.empty((B, Cc), dtype=torch.float32, device=x.device)
    check(lib().avexhip_mean_pool(_ptr(x.contiguous()), B, T, Cc, _ptr(out), _stream()), "mean_pool")
    return out


def rel_bucket(rel: int, num_buckets: int = 320, max_distance: int = 800) -> int:
    """Host-only helper (no GPU needed)."""
    return int(lib().avexhip_rel_bucket(int(rel), int(num_buckets), int(max_distance)))


# ---------------------------------------------------------------------------------------
# Encoder handle
# ---------------------------------------------------------------------------------------
_CFG_FIELDS = ("input_patch_size", "embed_dim", "encoder_layers", "encoder_embed_dim", "encoder_ffn_embed_dim",
               "encoder_attention_heads", "conv_pos", "conv_pos_groups", "num_buckets", "max_distance")


RESIDUAL_CODES = {"f32": 0, "fp32": 0, "float32": 0, "half": 1, "f16": 1, "bf16": 1, "operand": 1}
RESIDUAL_BATCH_INVARIANT = 2      # AVEXHIP_RESIDUAL_BATCH_INVARIANT: OR-ed into residual_dtype (include/avexhip.h)


def residual_code(residual, batch_invariant: bool = False) -> int:
    """``residual_dtype`` of the handle configs: "f32" / "half" (+ the batch-invariance bit: a clip's outputs are then bit-identical
    whatever batch it arrives in -- LayerNorm fold at every size, no split-K, one final LayerNorm + pool path; the reference's fp32 path
    is batch-independent, beats_model.py:279-429)."""
    try:
        code = RESIDUAL_CODES[str(residual).lower()]
    except KeyError as e:
        raise ValueError(f"residual must be 'f32' or 'half', got {residual!r}") from e
    return code | (RESIDUAL_BATCH_INVARIANT if batch_invariant else 0)


def make_beats_config(cfg: Mapping[str, object], operand_dtype="f16", max_chunk_clips: int = 0,
                      residual="half", batch_invariant: bool = False, hidden_shift: int = 0) -> BeatsConfig:
    act = str(cfg.get("activation_fn", "gelu"))
    if act not in _capi.FFN_CODES:
        raise RuntimeError(f"--activation-fn {act} not supported")      # the reference's own error (modules.py:237)
    if bool(cfg.get("layer_norm_first", False)) and bool(cfg.get("deep_norm", True)):
        raise AssertionError("deep_norm and layer_norm_first exclude each other (beats.py:275)")
    c = BeatsConfig()
    c.layer_norm_first = int(bool(cfg.get("layer_norm_first", False)))
    c.activation_fn = _capi.FFN_CODES[act]
    c.conv_bias = int(bool(cfg.get("conv_bias", False)))
    for f in _CFG_FIELDS:
        setattr(c, f, int(cfg[f]))
    if not bool(cfg.get("relative_position_embedding", True)):
        c.num_buckets = 0
    c.gru_rel_pos = int(bool(cfg.get("gru_rel_pos", True)))
    c.deep_norm = int(bool(cfg.get("deep_norm", True)))
    c.num_mel_bins = int(cfg.get("num_mel_bins", 128))
    c.sample_frequency = float(cfg.get("sample_frequency", 16000.0))
    c.frame_length_ms = float(cfg.get("frame_length", 25.0))
    c.frame_shift_ms = float(cfg.get("frame_shift", 10.0))
    c.fbank_mean = float(cfg.get("fbank_mean", 15.41663))
    c.fbank_std = float(cfg.get("fbank_std", 6.55582))
    c.operand_dtype = dtype_code(operand_dtype)
    c.max_chunk_clips = int(max_chunk_clips)
    c.residual_dtype = residual_code(residual, batch_invariant)
    c.hidden_shift = int(hidden_shift)
    return c


def tensor_table(state: Mapping[str, object]):
    """A state dict (torch tensors or numpy arrays, host or device) as the ``avexhip_tensor`` table the ``*_create`` entry points take:
    ``(array, count, keep_alive)``; the library copies every tensor during the call, ``keep_alive`` must live until it returns."""
    keep = []
    entries = []
    for name, val in state.items():
        if isinstance(val, torch.Tensor):
            t = val.detach()
            if t.dtype != torch.float32 or not t.is_contiguous():
                t = t.float().contiguous()
            keep.append(t)
            entries.append((name.encode(), int(t.data_ptr()), t.numel()))
        else:
            a = np.ascontiguousarray(val, np.float32)
            keep.append(a)
            entries.append((name.encode(), int(a.ctypes.data), a.size))
    arr = (Tensor * len(entries))()
    for i, (n, p, k) in enumerate(entries):
        arr[i].name, arr[i].data, arr[i].numel = n, p, k
    return arr, len(entries), keep


def handle_profile(fn, h) -> List[Tuple[str, float, float]]:
    """``*_last_profile`` of an encoder handle as ``[(stage, ms, flops)]``."""
    names = C.POINTER(C.c_char_p)()
    ms = C.POINTER(C.c_float)()
    fl = C.POINTER(C.c_double)()
    n = C.c_int(0)
    check(fn(h, C.byref(names), C.byref(ms), C.byref(fl), C.byref(n)), "last_profile")
    return [(names[i].decode(), float(ms[i]), float(fl[i])) for i in range(n.value)]


def pool_code(hook_pooled) -> int:
    """``hook_pooled`` of the encoder forwards as the C ABI's code: False / 0 full taps, True / 1 / "mean", 2 / "max", 3 / "cls_token"
    (the aggregations of ``extract_embeddings``, beats_model.py:403-417)."""
    if isinstance(hook_pooled, str):
        try:
            return {"none": 0, "mean": 1, "max": 2, "cls_token": 3}[hook_pooled]
        except KeyError as e:
            raise ValueError(f"Unsupported aggregation method: {hook_pooled}") from e
    code = int(hook_pooled)
    if not 0 <= code <= 3:
        raise ValueError(f"hook_pooled must be 0..3, got {hook_pooled!r}")
    return code


class BeatsGraph:
    """A recorded forward of a ``BeatsEncoder`` at one input shape.  ``wav`` (and ``frame_pad`` when asked for) are the graph's static
    inputs: write the next batch INTO them (``g.wav.copy_(x)``), call ``replay()``, read ``features`` / ``pooled`` / ``hooks[i]`` --
    the same tensors every time, overwritten by the next replay.  Replays are ordered on the current stream like any kernel."""

    def __init__(self, enc: "BeatsEncoder", B: int, T: int, hook_layers, hook_pooled, want_features, want_pooled, with_frame_pad, dev) -> None:
        self._enc = enc          # keeps the handle alive
        Tt = enc.num_tokens(T)
        if B <= 0 or Tt <= 0:
            raise AvexHipError(f"input too short: {T} samples give {Tt} tokens")
        self.tokens = Tt
        self.wav = torch.zeros((B, T), dtype=torch.float32, device=dev)
        self.frame_pad = torch.zeros((B, Tt), dtype=torch.uint8, device=dev) if with_frame_pad else None
        need = int(lib().avexhip_beats_workspace_bytes(enc._h, B, T))
        self._ws = torch.empty((need,), dtype=torch.uint8, device=dev)      # the graph's own: the encoder's may be reallocated
        self.hooks: Dict[int, torch.Tensor] = {}
        self._ptrs = (C.c_void_p * (enc.L + 1))()
        mask = 0
        code = pool_code(hook_pooled)      # the buffers are sized by the CODE the library gets ("none" is a true string and code 0)
        for i in sorted(set(int(x) for x in hook_layers)):
            if not 0 <= i <= enc.L:
                raise ValueError(f"hook layer {i} out of range 0..{enc.L}")
            self.hooks[i] = torch.empty((B, enc.E) if code else (B, Tt, enc.E), dtype=torch.float32, device=dev)
            self._ptrs[i] = int(self.hooks[i].data_ptr())
            mask |= 1 << i
        self.features = torch.empty((B, Tt, enc.E), dtype=torch.float32, device=dev) if want_features else None
        self.pooled = torch.empty((B, enc.E), dtype=torch.float32, device=dev) if want_pooled else None
        # the default stream cannot be captured: record on a side stream that follows / is followed by the current one
        side = torch.cuda.Stream(device=dev)
        cur = torch.cuda.current_stream(dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            self._g = lib().avexhip_beats_graph_capture(enc._h, _ptr(self.wav), B, T, T, _ptr(self.frame_pad), mask, self._ptrs, code,
                                                        _ptr(self.features), _ptr(self.pooled), _ptr(self._ws), self._ws.numel(), _stream())
        cur.wait_stream(side)
        if not self._g:
            raise AvexHipError(f"beats_graph_capture failed: {_capi.last_error()}")
        self.nodes = int(lib().avexhip_beats_graph_nodes(self._g))

    def replay(self) -> "BeatsGraph":
        """Launch the recorded forward.  The encoder's ``on_overflow`` policy applies to replays as to eager forwards -- ``"warn"`` (one call
        late, no synchronisation) and ``"raise"`` (synchronises) -- except ``"retry"``: a graph replays the one mode it recorded and cannot
        climb the retry ladder, so a clipped f16 value raises there as well, with the eager path's message."""
        check(lib().avexhip_beats_graph_launch(self._g, _stream()), "beats_graph_launch")
        enc = self._enc
        if enc.on_overflow != "ignore":
            new = enc._new_overflow(sync=enc.on_overflow in ("raise", "retry"))
            if new:
                msg = (f"avex_amd: {new} lane(s) clipped a value to the f16 range (+-65504) inside a replayed BEATs forward: the result is not the "
                       "reference's.  Use residual='f32' (out-of-range sums), hidden_shift=8 (out-of-range hidden activations), "
                       "operand_dtype='bf16' with residual='f32' (fp32's exponent range); on_overflow='retry' is not available for graphs "
                       "(a graph replays one recorded mode): run the batch through forward().")
                if enc.on_overflow in ("raise", "retry"):
                    raise AvexHipError(msg)
                warnings.warn(msg, RuntimeWarning, stacklevel=2)
        return self

    def close(self) -> None:
        if getattr(self, "_g", None):
            torch.cuda.synchronize()
            lib().avexhip_beats_graph_destroy(self._g)
            self._g = None

    def __del__(self) -> None:
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


class BeatsEncoder:
    """Owns an ``avexhip_beats`` handle built from an fp32 state dict (torch tensors or numpy arrays,
    host or device).  ``forward`` runs the whole path wav -> features / taps / pooled on the current stream."""

    # The rungs on_overflow="retry" climbs when the f16 range alarm fires, in order.  f16 operands keep the default mode's accuracy (3e-4 of
    # the reference, inside north_star's 1e-3); only the last rung gives that up (bf16 operands: 2e-3):
    #   1. f16 operands + fp32 residual stream: the pre-LayerNorm sums never pass through f16 (an out-of-range SUM is the common case);
    #   2./3. the same + hidden activations stored x 2^-8 / 2^-14 with the inverse folded into fc2's weights (exact powers of two:
    #      ``hidden_shift``, avexhip_beats_config): fc1 outputs up to 1.6e7 / 1e9 fit;
    #   4. bf16 operands + fp32 residual stream: fp32's exponent range everywhere.
    # Precision of the shifted rungs: a stored hidden activation h x 2^-k is an f16 NORMAL (11 significant bits, exact scaling) for
    # |h| >= 2^(k-14) and an f16 SUBNORMAL below that, where the absolute step is 2^(k-24): rung 2 (k = 8) rounds |h| < 2^-6 to multiples of
    # 2^-16, rung 3 (k = 14) rounds every |h| < 1 to multiples of 2^-10 (the modes rely on f16 denormals not being flushed: MODE.FP_DENORM
    # keeps them on gfx950, and the MFMA reads them).  GELU outputs below 1 then carry up to 5e-4 absolute error each -- measured on the
    # clean checkpoint the pooled embedding stays at the default mode's 3e-4 (tests/test_gpu_overflow.py::test_hidden_shift_is_exact_
    # scaling_on_a_clean_checkpoint), because fc2 averages 3072 of them -- but rung 3 is "holds 1e9" first and "exact" second; the
    # ladder tries it only after rung 2 alarmed.
    # A checkpoint whose WEIGHTS do not fit f16 (weights_fit, csrc/handle_core.h) cannot be served by any f16 rung: with on_overflow="retry"
    # the constructor then builds rung 4 as the serving handle (``served_by`` names it) instead of raising.
    RETRY_LADDER = (("f16 operands, fp32 residual stream", dict(operand_dtype="f16", residual="f32", hidden_shift=0)),
                    ("f16 operands, fp32 residual stream, hidden activations x 2^-8", dict(operand_dtype="f16", residual="f32", hidden_shift=8)),
                    ("f16 operands, fp32 residual stream, hidden activations x 2^-14", dict(operand_dtype="f16", residual="f32", hidden_shift=14)),
                    ("bf16 operands, fp32 residual stream", dict(operand_dtype="bf16", residual="f32", hidden_shift=0)))

    def __init__(self, cfg: Mapping[str, object], state: Mapping[str, object], operand_dtype="f16",
                 max_chunk_clips: int = 0, residual="half", on_overflow: Optional[str] = None, batch_invariant: bool = False,
                 hidden_shift: int = 0) -> None:
        """``on_overflow``: what to do when an f16 conversion inside the forward clipped a value to +-65504 (the handle's sticky range
        alarm, ``avexhip_beats_overflow_count``; the reference computes in fp32 and has no such limit, backbone.py:350-375):
        ``"warn"`` (default; checked without synchronising, so the warning may come one call late), ``"raise"``, ``"retry"`` (the
        batch is run again up ``RETRY_LADDER`` until a rung's own alarm stays quiet, and that result is returned; ``last_rung`` names
        the rung that served the last forward, ``None`` = the handle itself; both synchronise after every forward) or ``"ignore"``.
        Environment default: ``AVEX_AMD_ON_OVERFLOW``.  ``hidden_shift``: see ``RETRY_LADDER`` (0 = off)."""
        _capi.require_gpu()
        self.cfg = dict(cfg)
        self.on_overflow = (on_overflow or os.environ.get("AVEX_AMD_ON_OVERFLOW") or "warn").lower()
        if self.on_overflow not in ("warn", "raise", "retry", "ignore"):
            raise ValueError(f"on_overflow must be 'warn', 'raise', 'retry' or 'ignore', got {self.on_overflow!r}")
        self._overflow_seen = 0
        self.last_rung: Optional[str] = None
        self._rungs: Dict[str, Optional["BeatsEncoder"]] = {}      # retry ladder, built on first need (each is a second copy of the weights on the device)
        self._mode = (dtype_code(operand_dtype), residual_code(residual) & 1, int(hidden_shift))
        self._fallback_args = (dict(cfg), state, max_chunk_clips) if self.on_overflow == "retry" and dtype_code(operand_dtype) == _capi.F16 else None
        self.batch_invariant = bool(batch_invariant)      # a clip's outputs do not depend on the batch it arrives in (bit for bit); see residual_code
        self.ccfg = make_beats_config(cfg, operand_dtype, max_chunk_clips, residual, self.batch_invariant, hidden_shift)
        self.E = int(cfg["encoder_embed_dim"])
        self.L = int(cfg["encoder_layers"])
        arr, n, keep = tensor_table(state)
        self._h = lib().avexhip_beats_create(C.byref(self.ccfg), arr, n)
        self.served_by: Optional[str] = None      # set when the constructor itself had to climb the ladder (weights outside the f16 range)
        if not self._h and self.on_overflow == "retry" and dtype_code(operand_dtype) == _capi.F16 and "do not fit the f16 range" in _capi.last_error():
            # the retry contract is "a result is always returned": no f16 rung can hold these weights (the shifted ones scale fc2's up), so the
            # handle that serves every batch is the ladder's last rung
            name, mode = self.RETRY_LADDER[-1]
            logger.warning("avex_amd: %s  on_overflow='retry': serving with '%s'.", _capi.last_error(), name)
            self.ccfg = make_beats_config(cfg, mode["operand_dtype"], max_chunk_clips, mode["residual"], self.batch_invariant, mode["hidden_shift"])
            self._mode = (dtype_code(mode["operand_dtype"]), residual_code(mode["residual"]) & 1, mode["hidden_shift"])
            self._fallback_args = None      # nothing wider exists
            self.served_by = name
            self._h = lib().avexhip_beats_create(C.byref(self.ccfg), arr, n)
        del keep
        if not self._h:
            raise AvexHipError(f"beats_create failed: {_capi.last_error()}")
        self._ws: Optional[torch.Tensor] = None

    def _retry(self, msg: str, wav, kw) -> Dict[str, object]:
        """Climb ``RETRY_LADDER``: the first rung wider than this handle whose own alarm stays quiet serves the batch."""
        fcfg, fstate, fchunk = self._fallback_args
        for name, mode in self.RETRY_LADDER:
            key = (dtype_code(mode["operand_dtype"]), residual_code(mode["residual"]) & 1, mode["hidden_shift"])
            # skip rungs that are not wider than the handle: the same mode, or a narrower residual stream / shift
            if key == self._mode or (key[0] == self._mode[0] and (key[1] > self._mode[1] or key[2] < self._mode[2])):
                continue
            if name not in self._rungs:
                try:
                    self._rungs[name] = BeatsEncoder(fcfg, fstate, max_chunk_clips=fchunk, on_overflow="ignore", batch_invariant=self.batch_invariant, **mode)
                except AvexHipError as e:      # e.g. a shift that takes fc2's weights out of range, or a GLU feed-forward
                    logger.warning("avex_amd: retry rung '%s' is not available for this checkpoint: %s", name, e)
                    self._rungs[name] = None
            enc = self._rungs[name]
            if enc is None:
                continue
            out = enc.forward(wav, **kw)
            if enc._new_overflow(sync=True) == 0:
                logger.warning(msg + f"  Re-ran the batch with {name}.")
                self.last_rung = name
                return out
        raise AvexHipError(msg + "  No rung of the retry ladder could hold the values.")      # (unreachable for f16 handles: bf16 cannot clip)

    def num_tokens(self, T: int) -> int:
        return int(lib().avexhip_beats_num_tokens(self._h, T))

    def _workspace(self, B: int, T: int, device) -> torch.Tensor:
        need = int(lib().avexhip_beats_workspace_bytes(self._h, B, T))
        if self._ws is None or self._ws.numel() < need or self._ws.device != device:
            self._ws = None
            self._ws = torch.empty((need,), dtype=torch.uint8, device=device)
        return self._ws

    def forward(self, wav: torch.Tensor, *, hook_layers: Sequence[int] = (), hook_pooled=False,
                want_features: bool = True, want_pooled: bool = False, frame_pad: Optional[torch.Tensor] = None
                ) -> Dict[str, object]:
        """``hook_pooled``: False -> taps ``[B, T', E]``; True / "mean", "max", "cls_token" -> ``[B, E]``, reduced over the tokens on the
        device (inside the GEMM epilogue that produces the tap when clips have >= 64 tokens)."""
        _need_cuda(wav)
        if wav.dim() != 2:
            raise ValueError("wav must be [B, T]")
        if wav.dtype != torch.float32:
            wav = wav.float()
        if wav.stride(1) != 1:
            wav = wav.contiguous()
        B, T = wav.shape
        Tt = self.num_tokens(T)
        if B == 0 or Tt <= 0:
            raise AvexHipError(f"input too short: {T} samples give {Tt} tokens")
        dev = wav.device
        ws = self._workspace(B, T, dev)
        hooks: Dict[int, torch.Tensor] = {}
        ptrs = (C.c_void_p * (self.L + 1))()
        mask = 0
        code = pool_code(hook_pooled)      # sizes the tap buffers AND goes to the library: never the truthiness of the argument
        for i in sorted(set(int(x) for x in hook_layers)):
            if not 0 <= i <= self.L:
                raise ValueError(f"hook layer {i} out of range 0..{self.L}")
            shape = (B, self.E) if code else (B, Tt, self.E)
            hooks[i] = torch.empty(shape, dtype=torch.float32, device=dev)
            ptrs[i] = int(hooks[i].data_ptr())
            mask |= 1 << i
        feats = torch.empty((B, Tt, self.E), dtype=torch.float32, device=dev) if want_features else None
        pooled = torch.empty((B, self.E), dtype=torch.float32, device=dev) if want_pooled else None
        pad = None
        if frame_pad is not None:
            pad = frame_pad.to(device=dev, dtype=torch.uint8).contiguous()
            if pad.shape != (B, Tt):
                raise ValueError(f"frame_pad must be [B={B}, T'={Tt}], got {tuple(pad.shape)}")
        check(lib().avexhip_beats_forward(self._h, _ptr(wav), B, T, wav.stride(0), _ptr(pad), mask, ptrs,
                                          code, _ptr(feats), _ptr(pooled), _ptr(ws), ws.numel(), _stream()),
              "beats_forward")
        self.last_rung = None
        if self.on_overflow != "ignore":
            new = self._new_overflow(sync=self.on_overflow in ("raise", "retry"))
            if new:
                msg = (f"avex_amd: {new} lane(s) clipped a value to the f16 range (+-65504) inside the BEATs forward: the result is not the "
                       "reference's.  Use residual='f32' (out-of-range sums), hidden_shift=8 (out-of-range hidden activations), "
                       "operand_dtype='bf16' with residual='f32' (fp32's exponent range), or on_overflow='retry' (tries them in that order).")
                if self.on_overflow == "raise":
                    raise AvexHipError(msg)
                if self.on_overflow == "retry" and self._fallback_args is not None:
                    return self._retry(msg, wav, dict(hook_layers=hook_layers, hook_pooled=hook_pooled, want_features=want_features,
                                                      want_pooled=want_pooled, frame_pad=frame_pad))
                warnings.warn(msg, RuntimeWarning, stacklevel=2)
        return {"features": feats, "pooled": pooled, "hooks": hooks, "tokens": Tt}

    def capture(self, batch: int, samples: int, *, hook_layers: Sequence[int] = (), hook_pooled: bool = False,
                want_features: bool = True, want_pooled: bool = False, with_frame_pad: bool = False,
                device: Optional[torch.device] = None) -> "BeatsGraph":
        """Record the forward for ``[batch, samples]`` inputs as a hipGraph (``avexhip_beats_graph_capture``) with its own static input,
        output and workspace tensors; see ``BeatsGraph``.  For small batches, where the ~95 launches of a forward cost more than the
        kernels: one clip 1.7 ms -> see profiles/."""
        return BeatsGraph(self, batch, samples, hook_layers, hook_pooled, want_features, want_pooled, with_frame_pad,
                          device or torch.device("cuda", torch.cuda.current_device()))

    def overflow_events(self, sync: bool = True) -> int:
        """The handle's sticky range-alarm count (0 = no f16 conversion ever clipped); ``sync`` waits for the current stream first."""
        n = C.c_uint32(0)
        check(lib().avexhip_beats_overflow_count(self._h, C.byref(n), _stream(), int(bool(sync))), "beats_overflow_count")
        return int(n.value)

    def reset_overflow(self) -> None:
        check(lib().avexhip_beats_overflow_reset(self._h, _stream()), "beats_overflow_reset")
        torch.cuda.current_stream().synchronize()
        self._overflow_seen = 0

    def _new_overflow(self, sync: bool) -> int:
        n = self.overflow_events(sync=sync)
        new = n - self._overflow_seen
        self._overflow_seen = n
        return max(new, 0)

    def set_profiling(self, enabled: bool) -> None:
        check(lib().avexhip_beats_set_profiling(self._h, int(enabled)), "set_profiling")

    def last_profile(self) -> List[Tuple[str, float, float]]:
        names = C.POINTER(C.c_char_p)()
        ms = C.POINTER(C.c_float)()
        fl = C.POINTER(C.c_double)()
        n = C.c_int(0)
        check(lib().avexhip_beats_last_profile(self._h, C.byref(names), C.byref(ms), C.byref(fl), C.byref(n)), "last_profile")
        return [(names[i].decode(), float(ms[i]), float(fl[i])) for i in range(n.value)]

    def close(self) -> None:
        for enc in list(getattr(self, "_rungs", {}).values()):
            if enc is not None:
                enc.close()
        self._rungs = {}
        if getattr(self, "_h", None):
            lib().avexhip_beats_destroy(self._h)
            self._h = None
        self._ws = None

    def __del__(self) -> None:
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass
