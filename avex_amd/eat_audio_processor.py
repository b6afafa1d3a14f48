"""EAT frontend on the GPU: the contract of avex's ``EATAudioProcessor`` (avex/models/eat/audio_processor.py:19-143).

The reference loops over the batch on the CPU (``torchaudio.compliance.kaldi.fbank`` per clip, then pad/truncate to
``target_length`` frames and ``(mel - norm_mean) / (2 norm_std)``), moving tensors host<->device around it.  Here one launch of
the fused HIP filterbank kernel (``avex_amd/csrc/fbank.hip``, the same kernel as the BEATs frontend with a Hann window and no
2**15 scale) produces the whole ``[B, target_length, n_mels]`` batch on the device; a second tiny kernel supplies the per-clip
mean the reference removes first (``mono - mono.mean()``, :107).  Only the EAT *frontend* is built: the EAT encoder is
third-party remote code that is not part of the reference tree.
"""
from __future__ import annotations

from typing import Union

import numpy as np
import torch

from . import _capi
from .kernels import F32_EPS, FbankPlan, hann_window, kaldi_mel_filterbank

__all__ = ["EATAudioProcessor"]


class EATAudioProcessor:
    """Raw waveforms -> Mel FBanks ``(B, target_length, n_mels)`` float32, same constructor and call contract as the
    reference class (``sample_rate, target_length, n_mels, norm_mean, norm_std, frame_shift_ms, window_type``)."""

    def __init__(self, *, sample_rate: int = 16_000, target_length: int = 1024, n_mels: int = 128, norm_mean: float = -4.268,
                 norm_std: float = 4.569, frame_shift_ms: int = 10, window_type: str = "hanning") -> None:
        if window_type != "hanning":
            raise ValueError(f"window_type={window_type!r} is not built (the reference default 'hanning' is)")
        self.sample_rate = sample_rate
        self.target_length = target_length
        self.n_mels = n_mels
        self.norm_mean = norm_mean
        self.norm_std = norm_std
        self.frame_shift_ms = frame_shift_ms
        self.window_type = window_type
        self.hop_length: int = int(round(sample_rate * frame_shift_ms / 1_000))      # audio_processor.py:67
        self._win_length = int(sample_rate * 25 / 1000)                              # kaldi default frame_length 25 ms
        self._plan = None

    def _get_plan(self) -> FbankPlan:
        if self._plan is None:
            per_sample = self.norm_mean == 0.0 and self.norm_std == 1.0
            self._plan = FbankPlan(win_length=self._win_length, hop_length=self.hop_length, n_mels=self.n_mels, input_scale=1.0,
                                   preemph=0.97, remove_dc=True, log_floor=F32_EPS,
                                   norm_mean=0.0 if per_sample else float(self.norm_mean),
                                   norm_div=1.0 if per_sample else 2.0 * float(self.norm_std),
                                   window=hann_window(self._win_length),
                                   mel_fb=kaldi_mel_filterbank(self.n_mels, 512, float(self.sample_rate), 20.0, 0.0))
        return self._plan

    def __call__(self, wav: Union[torch.Tensor, np.ndarray]) -> torch.Tensor:
        if not isinstance(wav, torch.Tensor):
            wav = torch.as_tensor(wav, dtype=torch.float32)
        if wav.dim() == 1:
            wav = wav.unsqueeze(0)
        original_device = wav.device
        _capi.require_gpu()
        dev = wav.device if wav.is_cuda else torch.device("cuda", torch.cuda.current_device())
        x = wav.to(device=dev, dtype=torch.float32)
        mel = self._get_plan().padded(x, self.target_length, remove_clip_mean=True)
        if self.norm_mean == 0.0 and self.norm_std == 1.0:
            # per-sample statistics over the padded log-mel (audio_processor.py:131-134; unbiased std like torch.std)
            mean = mel.mean(dim=(1, 2), keepdim=True)
            std = mel.std(dim=(1, 2), keepdim=True)
            std = torch.where(std > 0, std, torch.ones_like(std))
            mel = (mel - mean) / (std * 2)
        return mel.to(original_device)
