"""AVES model class for the plugin registry, backed by the HIP kernels (registry key ``"aves"``).

Counterpart of the reference wrapper (avex/models/aves_model.py:62-262): raw-waveform input, ``forward`` returns the last
transformer layer's features ``(B, T', 768)`` (:149-150), hookable layers are the transformer blocks'
``model.encoder.transformer.layers.{i}.feed_forward.output_dense`` (:101-120), ``extract_embeddings`` gathers every
registered hook in registration order (:152-262).  The sub-module tree only HOLDS parameters under torchaudio's
``wav2vec2_model`` names (``model.*``), so a torchaudio AVES state dict loads with ``load_state_dict``; the arithmetic runs in
libavexhip.so through ``avex_amd.aves_encoder.AvesEncoder``.

Differences from the reference: the constructor does not download ``birdaves-biox-base.torchaudio.pt`` (:87-90; neither machine
has a network) -- weights arrive through ``load_state_dict`` / ``load_model(checkpoint_path=...)``; inference only; no CPU path.
PARITY UNPINNED against torchaudio (see oracle/aves_oracle.py).
"""
from __future__ import annotations

import os

import logging
from typing import Any, Dict, List, Optional, Union

import torch
import torch.nn as nn

from ._capi import AvexHipError
from .aves_encoder import AvesEncoder
from .base_model import ModelBase
from .configs import AudioConfig
from .synth import AVES_BASE_CFG

logger = logging.getLogger(__name__)


class AVESConfig:
    """AVESConfig of the reference (aves_model.py:19-59): wav2vec2-base layout."""

    def __init__(self, cfg: Optional[dict] = None) -> None:
        self.extractor_mode = "group_norm"
        self.extractor_conv_layer_config = [list(c) for c in AVES_BASE_CFG["extractor_conv_layer_config"]]
        self.extractor_conv_bias = False
        self.encoder_embed_dim = 768
        self.encoder_projection_dropout = 0.1
        self.encoder_pos_conv_kernel = 128
        self.encoder_pos_conv_groups = 16
        self.encoder_num_layers = 12
        self.encoder_num_heads = 12
        self.encoder_attention_dropout = 0.1
        self.encoder_ff_interm_features = 3072
        self.encoder_ff_interm_dropout = 0.0
        self.encoder_dropout = 0.1
        self.encoder_layer_norm_first = False
        self.encoder_layer_drop = 0.05
        if cfg is not None:
            self.update(cfg)

    def update(self, cfg: dict) -> None:
        self.__dict__.update(cfg)

    def to_dict(self) -> dict:
        return self.__dict__


class _ConvBlock(nn.Module):
    def __init__(self, cin: int, cout: int, k: int, s: int, group_norm: bool) -> None:
        super().__init__()
        self.layer_norm = nn.GroupNorm(cout, cout) if group_norm else None
        self.conv = nn.Conv1d(cin, cout, k, s, bias=False)


class _PosConv(nn.Module):
    def __init__(self, E: int, cg: int, K: int) -> None:
        super().__init__()
        self.conv = nn.Module()
        self.conv.bias = nn.Parameter(torch.zeros(E))
        self.conv.parametrizations = nn.Module()
        self.conv.parametrizations.weight = nn.Module()
        self.conv.parametrizations.weight.original0 = nn.Parameter(torch.ones(1, 1, K))
        self.conv.parametrizations.weight.original1 = nn.Parameter(torch.zeros(E, cg, K))


class _Attn(nn.Module):
    def __init__(self, E: int) -> None:
        super().__init__()
        self.k_proj = nn.Linear(E, E); self.v_proj = nn.Linear(E, E); self.q_proj = nn.Linear(E, E); self.out_proj = nn.Linear(E, E)


class _FF(nn.Module):
    def __init__(self, E: int, F: int) -> None:
        super().__init__()
        self.intermediate_dense = nn.Linear(E, F)
        self.output_dense = nn.Linear(F, E)


class _Layer(nn.Module):
    def __init__(self, E: int, F: int) -> None:
        super().__init__()
        self.attention = _Attn(E)
        self.layer_norm = nn.LayerNorm(E)
        self.feed_forward = _FF(E, F)
        self.final_layer_norm = nn.LayerNorm(E)


class Wav2Vec2Parameters(nn.Module):
    """Parameter tree with torchaudio ``Wav2Vec2Model``'s names (feature_extractor / encoder.feature_projection / encoder.transformer)."""

    def __init__(self, c: AVESConfig) -> None:
        super().__init__()
        self.feature_extractor = nn.Module()
        blocks, cin = [], 1
        for i, (cout, k, s) in enumerate(c.extractor_conv_layer_config):
            blocks.append(_ConvBlock(cin, cout, k, s, group_norm=(i == 0 and c.extractor_mode == "group_norm")))
            cin = cout
        self.feature_extractor.conv_layers = nn.ModuleList(blocks)
        E, F = c.encoder_embed_dim, c.encoder_ff_interm_features
        self.encoder = nn.Module()
        self.encoder.feature_projection = nn.Module()
        self.encoder.feature_projection.layer_norm = nn.LayerNorm(cin)
        self.encoder.feature_projection.projection = nn.Linear(cin, E)
        self.encoder.transformer = nn.Module()
        self.encoder.transformer.pos_conv_embed = _PosConv(E, E // c.encoder_pos_conv_groups, c.encoder_pos_conv_kernel)
        self.encoder.transformer.layer_norm = nn.LayerNorm(E)
        self.encoder.transformer.layers = nn.ModuleList([_Layer(E, F) for _ in range(c.encoder_num_layers)])


class Model(ModelBase):
    """AVES backbone on the MI355X HIP path (features mode only, like the reference wrapper's ``forward``)."""

    name = "aves"

    def __init__(self, *, num_classes: Optional[int] = None, pretrained: bool = False, device: str = "cuda",
                 audio_config: Optional[Union[AudioConfig, Dict[str, Any]]] = None, operand_dtype: str = "f16",
                 init_config: Optional[Dict[str, Any]] = None, batch_invariant: Optional[bool] = None, residual: Optional[str] = None) -> None:
        super().__init__(device=device, audio_config=audio_config)
        # inter-kernel residual stream, BEATs' policy: "auto" (default) = fp32 stream for calls that hand frames back (every call of this
        # wrapper does: forward() features and un-pooled taps, aves_model.py:129-150), "half" / "f32" force one (AvesEncoder)
        self.residual = (residual or os.environ.get("AVEX_AMD_RESIDUAL") or "auto").lower()
        # a clip's outputs bit-identical whatever batch it arrives in (kernels.residual_code; environment: AVEX_AMD_BATCH_INVARIANT=1)
        self.batch_invariant = bool(batch_invariant) if batch_invariant is not None else os.environ.get("AVEX_AMD_BATCH_INVARIANT", "0") not in ("", "0")
        if pretrained:
            raise FileNotFoundError("pretrained=True needs birdaves-biox-base.torchaudio.pt (aves_model.py:87-90), which is not "
                                    "reachable offline; load a local state dict with load_state_dict() / checkpoint_path=")
        self.num_classes = num_classes
        self.config = AVESConfig(init_config)
        self.operand_dtype = operand_dtype
        self.model = Wav2Vec2Parameters(self.config)
        self._encoder: Optional[AvesEncoder] = None
        self._weights_dirty = True
        try:
            self.to(device)
        except (RuntimeError, AssertionError) as e:
            logger.warning("could not move AVES parameters to %s (%s); they stay on CPU until .to() succeeds", device, e)

    def _apply(self, fn, *a, **k):
        self._weights_dirty = True
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        self._weights_dirty = True
        # torchaudio checkpoints come without the wrapper's "model." prefix (aves_model.py:91: self.model.load_state_dict)
        if state_dict and not any(k.startswith("model.") for k in state_dict):
            state_dict = {"model." + k: v for k, v in state_dict.items()}
        # old-style weight_norm names
        ren = {"model.encoder.transformer.pos_conv_embed.conv.weight_g": "model.encoder.transformer.pos_conv_embed.conv.parametrizations.weight.original0",
               "model.encoder.transformer.pos_conv_embed.conv.weight_v": "model.encoder.transformer.pos_conv_embed.conv.parametrizations.weight.original1"}
        state_dict = {ren.get(k, k): v for k, v in state_dict.items()}
        return super().load_state_dict(state_dict, strict=strict, assign=assign)

    def _ensure_encoder(self) -> AvesEncoder:
        if self._encoder is None or self._weights_dirty:
            p = next(self.parameters())
            if not p.is_cuda:
                raise AvexHipError(f"AVES parameters live on {p.device}; the avex_amd path runs on a GPU only (no CPU fallback)")
            with torch.cuda.device(p.device):
                state = {k: v.detach().float().cpu().numpy() for k, v in self.state_dict().items()}
                self._encoder = AvesEncoder(self.config.to_dict(), state, operand_dtype=self.operand_dtype, batch_invariant=self.batch_invariant, residual=self.residual)
            self._weights_dirty = False
        return self._encoder

    def _discover_embedding_layers(self) -> None:
        if len(self._layer_names) == 0:
            self._layer_names = [name for name, _ in self.named_modules()
                                 if name.endswith(".feed_forward.output_dense") and "model.encoder.transformer.layers." in name]

    def _prep_input(self, inputs: torch.Tensor) -> torch.Tensor:
        if inputs.ndim == 1:
            inputs = inputs.unsqueeze(0)
        return inputs.to(next(self.parameters()).device)

    def forward(self, x: torch.Tensor, padding_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``(B, T)`` raw waveform -> last-layer features ``(B, T', 768)`` (the reference ignores ``padding_mask`` here too)."""
        if x is None:
            raise ValueError("Input tensor cannot be None")
        x = self._prep_input(x).to(torch.float32)
        enc = self._ensure_encoder()
        taps = [layer.feed_forward.output_dense for layer in self.model.encoder.transformer.layers]
        hooked = [i for i, m in enumerate(taps) if len(m._forward_hooks) > 0]
        with torch.cuda.device(x.device):
            r = enc.forward(x.contiguous(), hook_layers=hooked, want_features=True)
        for i in hooked:
            self._fire_forward_hooks(taps[i], r["hooks"][i])
        return r["features"]

    def extract_embeddings(self, x: Union[torch.Tensor, Dict[str, torch.Tensor]], *, padding_mask: Optional[torch.Tensor] = None,
                           aggregation: str = "none", freeze_backbone: bool = True) -> Union[torch.Tensor, List[torch.Tensor]]:
        """All registered hooks' outputs (aves_model.py:152-262): list / tensor for ``"none"``, else ``(B, 768 * n)``."""
        if not self._hooks:
            raise ValueError("No hooks are registered in the model.")
        self._clear_hook_outputs()
        try:
            wav = x["raw_wav"] if isinstance(x, dict) else x
            mask = x.get("padding_mask") if isinstance(x, dict) else padding_mask
            with torch.no_grad():
                self.forward(wav, mask)
            embeddings = list(self._hook_outputs.values())
            if not embeddings:
                raise ValueError(f"No layers found matching: {self._hook_outputs.keys()}")
            return self._aggregate(embeddings, aggregation)
        finally:
            self._clear_hook_outputs()
