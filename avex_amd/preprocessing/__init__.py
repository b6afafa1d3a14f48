"""Frontend namespace: fused log-mel filterbank on the GPU.

The reference's ``avex/preprocessing`` package is empty (three 0-byte files); its real BEATs
frontend is ``_BatchedFbank`` + ``BEATs.preprocess`` (avex/models/beats/beats.py:39-163,304-323).
This module exposes the HIP implementation of that arithmetic under the name BASELINE.json uses.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from ..kernels import F32_EPS, FbankPlan, hann_window, kaldi_mel_filterbank, povey_window

__all__ = ["BatchedFbank", "beats_preprocess", "povey_window", "hann_window", "kaldi_mel_filterbank"]


class BatchedFbank:
    """``[B, T]`` waveform (already scaled, e.g. ``x * 2**15``) -> ``[B, frames, n_mels]`` log-mel,
    the contract of the reference's ``_BatchedFbank.forward`` (kaldi defaults: 25 ms / 10 ms frames,
    per-frame DC removal, pre-emphasis 0.97, Povey window, power spectrum, log with fp32-eps floor)."""

    def __init__(self, num_mel_bins: int = 128, sample_frequency: float = 16000.0, frame_length_ms: float = 25.0,
                 frame_shift_ms: float = 10.0, preemphasis_coefficient: float = 0.97, low_freq: float = 20.0,
                 high_freq: float = 0.0, window: Optional[np.ndarray] = None, mel_fb: Optional[np.ndarray] = None,
                 input_scale: float = 1.0, norm_mean: float = 0.0, norm_div: float = 1.0) -> None:
        self.win_length = int(sample_frequency * frame_length_ms / 1000.0)
        self.hop_length = int(sample_frequency * frame_shift_ms / 1000.0)
        self.n_fft = 512
        if self.win_length > self.n_fft:
            raise ValueError("frame length above 512 samples is not built (n_fft fixed at 512)")
        if mel_fb is None:
            mel_fb = kaldi_mel_filterbank(num_mel_bins, self.n_fft, sample_frequency, low_freq, high_freq)
        self._plan = FbankPlan(win_length=self.win_length, hop_length=self.hop_length, n_mels=num_mel_bins,
                               input_scale=input_scale, preemph=preemphasis_coefficient, remove_dc=True,
                               log_floor=F32_EPS, norm_mean=norm_mean, norm_div=norm_div, window=window, mel_fb=mel_fb)

    def num_frames(self, T: int) -> int:
        return self._plan.num_frames(T)

    def __call__(self, waveforms: torch.Tensor) -> torch.Tensor:
        return self._plan(waveforms)

    forward = __call__


def beats_preprocess(source: torch.Tensor, fbank_mean: float = 15.41663, fbank_std: float = 6.55582,
                     _cache: dict = {}) -> torch.Tensor:
    """``BEATs.preprocess``: ``(fbank(x * 2**15) - mean) / (2 * std)`` in one kernel."""
    key = (float(fbank_mean), float(fbank_std))
    if key not in _cache:
        _cache[key] = BatchedFbank(input_scale=32768.0, norm_mean=fbank_mean, norm_div=2.0 * fbank_std)
    return _cache[key](source.float())
