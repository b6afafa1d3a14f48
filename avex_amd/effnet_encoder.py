"""EfficientNet-B0 ``features`` stack on the MI355X kernels.

The reference (avex/models/efficientnet.py:55-66, 116-137, 208) feeds a mel spectrogram, repeated to three channels, to
``torchvision.models.efficientnet_b0().features``.  Here:

* activations are NHWC in the operand type, channels padded to a multiple of 128 (padding channels are exactly zero in every
  layer: zero weights, zero bias, SiLU(0) = 0), so every 1x1 convolution (expand, project, head: all of the FLOPs that matter)
  is an ``avexhip_gemm`` over ``[B*H*W, C]`` rows with eval-mode BatchNorm folded into weight and bias, SiLU (``gelu = 2``) or
  the residual add in the epilogue;
* the three identical input channels collapse the stem into a 1-channel 3x3 convolution with channel-summed weights;
* depthwise convolutions, the squeeze-excitation pool (accumulated by the depthwise kernel on its way out), its two tiny
  fully connected layers and the channel rescale are the bandwidth-bound HIP kernels of ``csrc/effnet.hip``.

PARITY UNPINNED: torchvision is third-party, absent from the reference tree and both machines; the checker is
``oracle/effnet_oracle.py`` on synthetic weights.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Iterable, List, Mapping, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _capi
from . import kernels as K
from .synth import EFFNET_B0_STAGES

__all__ = ["EfficientNetB0Encoder"]


class EfficientNetB0Encoder:
    """``[B, n_mels, frames]`` fp32 mel images on the GPU -> ``features [B, 1280, H', W']`` (and the reference's hook taps).
    A thin wrapper over the ``avexhip_effnet`` handle (csrc/effnet_handle.cpp): the library owns the (BatchNorm-folded) weights, this
    class the output tensors and the workspace."""

    def __init__(self, state: Mapping[str, np.ndarray], operand_dtype: str = "f16", prefix: str = "model.", stages: Sequence = EFFNET_B0_STAGES,
                 bn_eps: float = 1e-5, max_chunk_clips: int = 0) -> None:
        _capi.require_gpu()
        self.dtype = operand_dtype
        self.stages = [tuple(int(v) for v in s) for s in stages]
        if len(self.stages) > 8:
            raise K.AvexHipError("EfficientNet: at most 8 stages")
        c = _capi.EffnetConfig()
        c.n_stages = len(self.stages)
        for i, st in enumerate(self.stages):
            for j in range(6):
                c.stage[i][j] = st[j]
        c.stem_channels, c.head_channels, c.bn_eps = 32, 1280, float(bn_eps)
        c.operand_dtype, c.max_chunk_clips = _capi.dtype_code(operand_dtype), int(max_chunk_clips)
        sub = {k[len(prefix):]: v for k, v in state.items() if k.startswith(prefix)} if prefix else dict(state)
        arr, n, keep = K.tensor_table(sub)
        self._h = _capi.lib().avexhip_effnet_create(C.byref(c), arr, n)
        del keep
        if not self._h:
            raise K.AvexHipError(f"effnet_create failed: {_capi.last_error()}")
        self._ws: Optional[torch.Tensor] = None
        # hookable convolutions in the reference's order (efficientnet.py:82-114): stem, every block's projection (blocks with an expansion), head
        names = [prefix + "features.0.0"]
        for si, (er, _k, _s, _cin, _cout, n_rep) in enumerate(self.stages, start=1):
            if er != 1:
                names += [prefix + f"features.{si}.{j}.block.3.0" for j in range(n_rep)]
        names.append(prefix + f"features.{len(self.stages) + 1}.0")
        self._tap_names = names
        if len(names) != int(_capi.lib().avexhip_effnet_num_taps(self._h)):
            raise K.AvexHipError("EfficientNet: tap list of the wrapper and of the library disagree")
        self.stem_name, self.head_name = names[0], names[-1]

    def tap_names(self) -> List[str]:
        return list(self._tap_names)

    def _shape(self, tap: int, H: int, W: int) -> Tuple[int, int, int]:
        cc, ho, wo = C.c_int(0), C.c_int(0), C.c_int(0)
        _capi.check(_capi.lib().avexhip_effnet_tap_shape(self._h, tap, H, W, C.byref(cc), C.byref(ho), C.byref(wo)), "effnet_tap_shape")
        return cc.value, ho.value, wo.value

    @torch.no_grad()
    def forward(self, mel: torch.Tensor, hook_layers: Iterable[str] = (), want_features: bool = True, want_pooled: bool = False) -> Dict[str, object]:
        """``hook_layers``: names out of ``tap_names()``; their raw (pre-BatchNorm) convolution outputs come back as fp32 NCHW."""
        if mel.dim() != 3 or mel.dtype != torch.float32 or not mel.is_cuda:
            raise ValueError("mel must be a [B, n_mels, frames] float32 CUDA tensor")
        mel = mel.contiguous()
        B, H, W = mel.shape
        dev = mel.device
        need = int(_capi.lib().avexhip_effnet_workspace_bytes(self._h, B, H, W))
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = None
            self._ws = torch.empty((need,), dtype=torch.uint8, device=dev)
        hooks: Dict[str, torch.Tensor] = {}
        ptrs = (C.c_void_p * len(self._tap_names))()
        mask = 0
        for name in set(hook_layers):
            if name not in self._tap_names:
                raise ValueError(f"{name!r} is not a hookable layer: {self._tap_names}")
            i = self._tap_names.index(name)
            hooks[name] = torch.empty((B,) + self._shape(i, H, W), dtype=torch.float32, device=dev)
            ptrs[i] = int(hooks[name].data_ptr())
            mask |= 1 << i
        ch, ho, wo = self._shape(-1, H, W)
        feats = torch.empty((B, ch, ho, wo), dtype=torch.float32, device=dev) if want_features else None
        pooled = torch.empty((B, ch), dtype=torch.float32, device=dev) if want_pooled else None
        _capi.check(_capi.lib().avexhip_effnet_forward(self._h, K._ptr(mel), B, H, W, mask, ptrs, K._ptr(feats), K._ptr(pooled), K._ptr(self._ws),
                                                       self._ws.numel(), K._stream()), "effnet_forward")
        out: Dict[str, object] = {"hooks": hooks}
        if want_features:
            out["features"] = feats                                  # (B, C, H, W) like the reference
        if want_pooled:
            out["pooled"] = pooled
        return out

    def overflow_events(self, sync: bool = True) -> int:
        n = C.c_uint32(0)
        _capi.check(_capi.lib().avexhip_effnet_overflow_count(self._h, C.byref(n), K._stream(), int(bool(sync))), "effnet_overflow_count")
        return int(n.value)

    def set_profiling(self, enabled: bool) -> None:
        _capi.check(_capi.lib().avexhip_effnet_set_profiling(self._h, int(enabled)), "effnet_set_profiling")

    def last_profile(self):
        return K.handle_profile(_capi.lib().avexhip_effnet_last_profile, self._h)

    def close(self) -> None:
        if getattr(self, "_h", None):
            _capi.lib().avexhip_effnet_destroy(self._h)
            self._h = None
        self._ws = None

    def __del__(self) -> None:
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass
