"""EfficientNet-B0 model class for the plugin registry, backed by the HIP kernels (registry key ``"efficientnet"``).

Counterpart of the reference wrapper (avex/models/efficientnet.py:21-322): ``process_audio`` runs the configured
``AudioProcessor`` (mel spectrogram on the GPU, avex_amd/csrc/melspec.hip; the reference then repeats the image to three
channels, :133-135 -- here the stem convolution sums its weights over the three identical channels instead);
``forward`` returns ``model.features(x)`` as ``(B, 1280, H', W')`` in features mode or the classifier logits (:159-215);
hookable layers are ``model.features.0.0``, every ``*.block.3.0`` and ``model.features.8.0`` (:82-114) and deliver the raw
convolution outputs before their BatchNorm; ``extract_embeddings`` reduces 4-D taps over the time axis and flattens (:217-322).
The sub-module tree only HOLDS parameters under torchvision's ``efficientnet_b0`` names (``model.*``), so a reference checkpoint
loads with ``load_state_dict``.  Inference only, eval-mode BatchNorm, no CPU path; only variant ``"b0"`` is built.
PARITY UNPINNED against torchvision (see oracle/effnet_oracle.py).
"""
from __future__ import annotations

import logging
from typing import Any, Dict, List, Optional, Union

import torch
import torch.nn as nn

from ._capi import AvexHipError
from .base_model import ModelBase
from .configs import AudioConfig
from .effnet_encoder import EfficientNetB0Encoder
from .synth import EFFNET_B0_STAGES, EFFNET_STAGES

logger = logging.getLogger(__name__)


def _cna(cin: int, cout: int, k: int, s: int, groups: int = 1, act: bool = True) -> nn.Sequential:
    mods: List[nn.Module] = [nn.Conv2d(cin, cout, k, s, (k - 1) // 2, groups=groups, bias=False), nn.BatchNorm2d(cout)]
    if act:
        mods.append(nn.SiLU())
    return nn.Sequential(*mods)


class _SE(nn.Module):
    def __init__(self, c: int, cs: int) -> None:
        super().__init__()
        self.fc1 = nn.Conv2d(c, cs, 1)
        self.fc2 = nn.Conv2d(cs, c, 1)


class _MBConv(nn.Module):
    def __init__(self, er: int, k: int, s: int, cin: int, cout: int) -> None:
        super().__init__()
        ce = cin * er
        layers: List[nn.Module] = []
        if er != 1:
            layers.append(_cna(cin, ce, 1, 1))
        layers.append(_cna(ce, ce, k, s, groups=ce))
        layers.append(_SE(ce, max(1, cin // 4)))
        layers.append(_cna(ce, cout, 1, 1, act=False))
        self.block = nn.Sequential(*layers)


class EfficientNetParameters(nn.Module):
    """Parameter tree with torchvision ``EfficientNet``'s names (features / classifier)."""

    def __init__(self, num_outputs: int = 1000, stages=EFFNET_B0_STAGES) -> None:
        super().__init__()
        feats: List[nn.Module] = [_cna(3, stages[0][3], 3, 2)]
        for (er, k, s, cin, cout, n) in stages:
            feats.append(nn.Sequential(*[_MBConv(er, k, s if j == 0 else 1, cin if j == 0 else cout, cout) for j in range(n)]))
        feats.append(_cna(stages[-1][4], 1280, 1, 1))
        self.features = nn.Sequential(*feats)
        self.classifier = nn.Sequential(nn.Dropout(0.2), nn.Linear(1280, num_outputs))


class Model(ModelBase):
    """EfficientNet-B0 on the MI355X HIP path."""

    name = "efficientnet"

    def __init__(self, num_classes: Optional[int] = None, pretrained: bool = False, device: str = "cuda",
                 audio_config: Optional[Union[AudioConfig, Dict[str, Any]]] = None, return_features_only: bool = False,
                 efficientnet_variant: str = "b0", operand_dtype: str = "f16") -> None:
        super().__init__(device=device, audio_config=audio_config)
        if efficientnet_variant not in EFFNET_STAGES:                     # efficientnet.py:61-68: b0 and b1
            raise ValueError(f"Unsupported EfficientNet variant: {efficientnet_variant}")
        self.efficientnet_variant = efficientnet_variant
        self._stages = EFFNET_STAGES[efficientnet_variant]
        if pretrained:
            raise FileNotFoundError("pretrained=True needs torchvision's IMAGENET1K_V1 weights (efficientnet.py:56), which are not "
                                    "reachable offline; load a local state dict with load_state_dict() / checkpoint_path=")
        if num_classes is None:
            return_features_only = True
        self.num_classes = num_classes
        self.return_features_only = return_features_only
        self.gradient_checkpointing = False
        self.audio_config = audio_config
        self.operand_dtype = operand_dtype
        self.model = EfficientNetParameters(1000 if return_features_only else int(num_classes), self._stages)
        self._encoder: Optional[EfficientNetB0Encoder] = None
        self._weights_dirty = True
        try:
            self.to(device)
        except (RuntimeError, AssertionError) as e:
            logger.warning("could not move EfficientNet parameters to %s (%s); they stay on CPU until .to() succeeds", device, e)
        self.eval()

    def _apply(self, fn, *a, **k):
        self._weights_dirty = True
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        self._weights_dirty = True
        if state_dict and not any(k.startswith("model.") for k in state_dict):
            state_dict = {"model." + k: v for k, v in state_dict.items()}
        return super().load_state_dict(state_dict, strict=strict, assign=assign)

    def _ensure_encoder(self) -> EfficientNetB0Encoder:
        if self._encoder is None or self._weights_dirty:
            p = next(self.parameters())
            if not p.is_cuda:
                raise AvexHipError(f"EfficientNet parameters live on {p.device}; the avex_amd path runs on a GPU only (no CPU fallback)")
            with torch.cuda.device(p.device):
                state = {k: v.detach().cpu().numpy() for k, v in self.state_dict().items() if k.startswith("model.features.")}
                self._encoder = EfficientNetB0Encoder(state, operand_dtype=self.operand_dtype, stages=self._stages)
            self._weights_dirty = False
        return self._encoder

    def _discover_embedding_layers(self) -> None:
        if len(self._layer_names) == 0:
            names = []
            for name, _m in self.named_modules():
                if name == "model.features.0.0" or name == "model.features.8.0" or (name.endswith(".block.3.0") and "model.features." in name):
                    names.append(name)
            self._layer_names = names

    def process_audio(self, x: torch.Tensor) -> torch.Tensor:
        """float32 cast + AudioProcessor (efficientnet.py:116-137); the 3-channel repeat is folded into the stem weights."""
        if x.dtype != torch.float32:
            x = x.to(torch.float32)
        dev = next(self.parameters()).device
        x = x.to(dev)                                  # the spectrogram runs on the GPU: move the waveform, not the image
        return super().process_audio(x)

    def enable_gradient_checkpointing(self) -> None:
        self.gradient_checkpointing = True             # inference path: nothing to checkpoint, kept for API parity

    def forward(self, x: torch.Tensor, padding_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``(B, T)`` waveform (or an already computed ``(B, n_mels, frames)`` image) -> ``(B, 1280, H', W')`` or logits."""
        x = self.process_audio(x)
        if x.dim() == 4:                               # (B, 3, F, T) as the reference would pass on: the channels are copies
            x = x[:, 0]
        if x.dim() != 3:
            raise ValueError(f"EfficientNet expects a (batch, freq, time) image after audio processing, got {tuple(x.shape)}")
        enc = self._ensure_encoder()
        named = dict(self.named_modules())
        hooked = [n for n in enc.tap_names() if len(named[n]._forward_hooks) > 0]
        with torch.cuda.device(x.device):
            r = enc.forward(x.contiguous(), hook_layers=hooked, want_features=True)
        for n in hooked:
            self._fire_forward_hooks(named[n], r["hooks"][n])
        features = r["features"]
        if self.return_features_only:
            return features
        pooled = features.mean(dim=(2, 3))             # model.avgpool + flatten (efficientnet.py:212-214)
        return self.model.classifier[1](pooled)

    def extract_embeddings(self, x: Union[torch.Tensor, Dict[str, torch.Tensor]], *, padding_mask: Optional[torch.Tensor] = None,
                           aggregation: str = "none", freeze_backbone: bool = True) -> Union[torch.Tensor, List[torch.Tensor]]:
        if not self._hooks:
            raise ValueError("No hooks are registered in the model.")
        self._clear_hook_outputs()
        try:
            wav = x["raw_wav"] if isinstance(x, dict) else x
            with torch.no_grad():
                self.forward(wav, padding_mask)
            order = self._hook_layers if self._hook_layers else list(self._hook_outputs.keys())
            embeddings = [self._hook_outputs[n] for n in order]
            if not embeddings:
                raise ValueError("No outputs were captured from registered hooks.")
            if aggregation == "none":
                return embeddings[0] if len(embeddings) == 1 else embeddings
            for i in range(len(embeddings)):
                if embeddings[i].dim() == 2:
                    continue
                if aggregation == "mean":
                    embeddings[i] = embeddings[i].mean(dim=-1)
                elif aggregation == "max":
                    embeddings[i] = embeddings[i].max(dim=-1)[0]
                elif aggregation == "cls_token":
                    embeddings[i] = embeddings[i][:, 0, :]
                else:
                    raise ValueError(f"Unsupported aggregation method: {aggregation}")
                if embeddings[i].dim() == 3:
                    embeddings[i] = embeddings[i].reshape(embeddings[i].shape[0], -1)
                else:
                    raise ValueError(f"Unexpected embedding dimension: {embeddings[i].dim()}. Expected 2, 3, or 4.")
            return torch.cat(embeddings, dim=1)
        finally:
            self._clear_hook_outputs()
