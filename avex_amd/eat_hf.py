"""EAT model class for the plugin registry, backed by the HIP kernels (registry key ``"eat_hf"``).

Counterpart of the reference wrapper ``EATHFModel`` (avex/models/eat_hf.py:111-419): same constructor keywords
(``model_name, num_classes, device, audio_config, target_length, pooling, fairseq_weights_path, norm_mean, norm_std,
return_features_only``), the same errors, ``forward`` = waveform -> 128-bin Mel FBanks (``EATAudioProcessor``) ->
``backbone.extract_features`` -> ``(B, 513, 768)`` unpooled patch embeddings in features mode, CLS / mean pooling + ``classifier``
otherwise (:241-289); hookable layers are ``backbone.model.blocks.{i}.attn.proj`` (:220-236); ``extract_embeddings`` has the
extra ``pooling`` keyword (:294-405).

The reference obtains ``backbone`` from ``transformers.AutoModel.from_pretrained("worstchan/EAT-base_epoch30_pretrain",
trust_remote_code=True)`` (:201): HF Hub remote code that neither machine can reach.  Here ``backbone`` is a parameter tree with that
remote model's names (``backbone.model.local_encoder.proj``, ``.extra_tokens``, ``.fixed_positional_encoder.positions``,
``.pre_norm``, ``.blocks.{i}.{attn.qkv, attn.proj, norm1, mlp.fc1, mlp.fc2, norm2}``), so an avex EAT checkpoint loads with
``load_state_dict``; the arithmetic runs in libavexhip.so through ``avex_amd.eat_encoder.EatEncoder``.  PARITY UNPINNED against
the remote code (see oracle/eat_oracle.py).  ``fairseq_weights_path`` applies the reference's key renaming (:55-73) to a local
fairseq checkpoint.  Inference only; no CPU path.
"""
from __future__ import annotations

import os

import logging
from collections import OrderedDict
from typing import Any, Dict, List, Optional, Union

import torch
import torch.nn as nn

from ._capi import AvexHipError
from .base_model import ModelBase
from .eat_audio_processor import EATAudioProcessor
from .eat_encoder import EatEncoder
from .synth import EAT_BASE_CFG

logger = logging.getLogger(__name__)


def rename_fairseq_key(key: str) -> str:
    """fairseq -> HF parameter names, as the reference's ``load_fairseq_weights._rename_key`` does (eat_hf.py:55-73)."""
    if key == "modality_encoders.IMAGE.context_encoder.norm.weight":
        return "model.pre_norm.weight"
    if key == "modality_encoders.IMAGE.context_encoder.norm.bias":
        return "model.pre_norm.bias"
    img_prefix = "modality_encoders.IMAGE."
    if key.startswith(img_prefix):
        key = "model." + key[len(img_prefix):]
    elif not key.startswith("model."):
        key = "model." + key
    return key


class _Attn(nn.Module):
    def __init__(self, E: int) -> None:
        super().__init__()
        self.qkv = nn.Linear(E, 3 * E)
        self.proj = nn.Linear(E, E)


class _Mlp(nn.Module):
    def __init__(self, E: int, F: int) -> None:
        super().__init__()
        self.fc1 = nn.Linear(E, F)
        self.fc2 = nn.Linear(F, E)


class _Block(nn.Module):
    def __init__(self, E: int, F: int, eps: float) -> None:
        super().__init__()
        self.attn = _Attn(E)
        self.norm1 = nn.LayerNorm(E, eps=eps)
        self.mlp = _Mlp(E, F)
        self.norm2 = nn.LayerNorm(E, eps=eps)


class _PositionTable(nn.Module):
    def __init__(self, rows: int, E: int) -> None:
        super().__init__()
        self.positions = nn.Parameter(torch.zeros(1, rows, E), requires_grad=False)


class EATBackboneParameters(nn.Module):
    """Parameter tree with the HF remote ``EATModel``'s names (held under ``.model`` like ``AutoModel`` returns it)."""

    def __init__(self, cfg: Dict[str, Any]) -> None:
        super().__init__()
        E, L, P = int(cfg["embed_dim"]), int(cfg["depth"]), int(cfg["patch_size"])
        F = E * int(cfg["mlp_ratio"])
        eps = float(cfg.get("norm_eps", 1e-6))
        m = nn.Module()
        m.local_encoder = nn.Module()
        m.local_encoder.proj = nn.Conv2d(1, E, P, stride=P)
        m.extra_tokens = nn.Parameter(torch.zeros(1, 1, E))
        m.fixed_positional_encoder = _PositionTable(int(cfg["max_length"]) * (int(cfg["n_mels"]) // P), E)
        m.pre_norm = nn.LayerNorm(E, eps=eps)
        m.blocks = nn.ModuleList([_Block(E, F, eps) for _ in range(L)])
        self.model = m


class EATHFModel(ModelBase):
    """EAT backbone on the MI355X HIP path."""

    name = "eat_hf"

    def __init__(self, *, model_name: str = "worstchan/EAT-base_epoch30_pretrain", num_classes: Optional[int] = None, device: str = "cuda",
                 audio_config: Optional[Dict[str, Any]] = None, target_length: int = 1024, pooling: str = "cls",
                 fairseq_weights_path: Optional[str] = None, norm_mean: float = -4.268, norm_std: float = 4.569,
                 return_features_only: bool = False, operand_dtype: str = "f16", init_config: Optional[Dict[str, Any]] = None,
                 batch_invariant: Optional[bool] = None, residual: Optional[str] = None) -> None:
        super().__init__(device=device, audio_config=audio_config)
        if num_classes is None:
            num_classes = 0
        # inter-kernel residual stream, BEATs' policy (beats_model.py here): "auto" (default) = fp32 stream for every call that hands back
        # un-averaged rows -- features, taps, the class token (this wrapper's default pooling, eat_hf.py:149,281-282) -- and the operand
        # type for token means only; "half" / "f32" force one (EatEncoder)
        self.residual = (residual or os.environ.get("AVEX_AMD_RESIDUAL") or "auto").lower()
        if not return_features_only and num_classes == 0:                        # eat_hf.py:175-176
            raise ValueError("num_classes must be > 0 when return_features_only=False")
        self.pooling = pooling
        self.num_classes = num_classes
        self.return_features_only = return_features_only
        self.audio_config = audio_config
        self.model_name = model_name
        self.operand_dtype = operand_dtype
        # a clip's outputs bit-identical whatever batch it arrives in (kernels.residual_code; environment: AVEX_AMD_BATCH_INVARIANT=1)
        self.batch_invariant = bool(batch_invariant) if batch_invariant is not None else os.environ.get("AVEX_AMD_BATCH_INVARIANT", "0") not in ("", "0")
        self.norm_mean, self.norm_std = norm_mean, norm_std
        self.config = dict(EAT_BASE_CFG, target_length=target_length)
        if init_config:
            self.config.update(init_config)
        # the dedicated frontend replaces whatever audio_config says (eat_hf.py:185-195)
        self.audio_processor = EATAudioProcessor(sample_rate=16_000, target_length=target_length, n_mels=128, norm_mean=norm_mean,
                                                 norm_std=norm_std)
        self.backbone = EATBackboneParameters(self.config)
        self.classifier = nn.Linear(int(self.config["embed_dim"]), num_classes) if (not return_features_only and num_classes > 0) else None
        self._encoder: Optional[EatEncoder] = None
        self._weights_dirty = True
        if fairseq_weights_path is not None:
            self.load_fairseq_weights(fairseq_weights_path)
        try:
            self.to(device)
        except (RuntimeError, AssertionError) as e:
            logger.warning("could not move EAT parameters to %s (%s); they stay on CPU until .to() succeeds", device, e)

    # ------------------------------------------------------------------ weights
    def _apply(self, fn, *a, **k):
        self._weights_dirty = True
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        self._weights_dirty = True
        return super().load_state_dict(state_dict, strict=strict, assign=assign)

    def load_fairseq_weights(self, weights_path: str) -> None:
        """A local fairseq EAT checkpoint (``{"model": state_dict}``) into the HF-named tree (eat_hf.py:44-108)."""
        from .weights import load_checkpoint_file
        ckpt = load_checkpoint_file(weights_path)
        alt = ckpt["model"] if isinstance(ckpt, dict) and "model" in ckpt else ckpt
        have = set(self.backbone.state_dict().keys())
        mapped: "OrderedDict[str, torch.Tensor]" = OrderedDict()
        for k, v in alt.items():
            if k.startswith("_ema"):
                continue
            nk = rename_fairseq_key(k)
            if nk in have:
                mapped[nk] = v
            else:
                logger.info("[skip] %s -> %s (not in the EAT parameter tree)", k, nk)
        missing, unexpected = self.backbone.load_state_dict(mapped, strict=False)
        if missing:
            logger.warning("EAT: missing keys after loading the fairseq checkpoint: %s", missing)
        if unexpected:
            logger.warning("EAT: unexpected keys after loading the fairseq checkpoint: %s", unexpected)
        self._weights_dirty = True

    def _ensure_encoder(self) -> EatEncoder:
        if self._encoder is None or self._weights_dirty:
            p = next(self.backbone.parameters())
            if not p.is_cuda:
                raise AvexHipError(f"EAT parameters live on {p.device}; the avex_amd path runs on a GPU only (no CPU fallback)")
            with torch.cuda.device(p.device):
                state = {"backbone." + k: v for k, v in self.backbone.state_dict().items()}
                self._encoder = EatEncoder(self.config, state, operand_dtype=self.operand_dtype, norm_mean=self.norm_mean, norm_std=self.norm_std,
                                           batch_invariant=self.batch_invariant, residual=self.residual)
            self._weights_dirty = False
        return self._encoder

    # ------------------------------------------------------------------ layers
    def _discover_embedding_layers(self) -> None:
        """Only ``backbone.model.blocks.{i}.attn.proj`` (eat_hf.py:220-236)."""
        if len(self._layer_names) == 0:
            self._layer_names = [n for n, _ in self.named_modules() if n.endswith("attn.proj") and "backbone.model.blocks." in n]

    # ------------------------------------------------------------------ forward
    def forward(self, x: torch.Tensor, padding_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Raw waveform ``(B, T)`` -> unpooled ``(B, 513, 768)`` in features mode (or without a classifier), else logits
        ``(B, num_classes)``; ``padding_mask`` is unused, as in the reference."""
        if x is None:
            raise ValueError("Input tensor cannot be None")
        if x.dim() == 1:
            x = x.unsqueeze(0)
        enc = self._ensure_encoder()
        dev = next(self.backbone.parameters()).device
        wav = x.to(device=dev, dtype=torch.float32).contiguous()
        taps = [blk.attn.proj for blk in self.backbone.model.blocks]
        hooked = [i for i, m in enumerate(taps) if len(m._forward_hooks) > 0]
        with torch.cuda.device(dev):
            r = enc.forward(wav, hook_layers=hooked, want_features=True)
        for i in hooked:
            self._fire_forward_hooks(taps[i], r["hooks"][i])
        feats = r["features"]
        if self.return_features_only or self.classifier is None:
            return feats
        if self.pooling == "cls":
            pooled = feats[:, 0]
        elif self.pooling == "mean":
            pooled = feats.mean(dim=1)
        else:
            raise ValueError("pooling must be 'cls' or 'mean'")
        from . import kernels as K                                    # the classification head on the device (fp32 MFMA)
        return K.dense_f32(pooled.contiguous(), self.classifier.weight.detach(), self.classifier.bias.detach())

    # ------------------------------------------------------------------ embeddings
    def extract_embeddings(self, x: Union[torch.Tensor, Dict[str, torch.Tensor]], *, padding_mask: Optional[torch.Tensor] = None,
                           pooling: str = "cls", aggregation: str = "none", freeze_backbone: bool = True
                           ) -> Union[torch.Tensor, List[torch.Tensor]]:  # type: ignore[override]
        self.ensure_hooks_registered()
        if not self._hooks:
            raise ValueError("No hooks are registered in the model.")
        self._clear_hook_outputs()
        try:
            wav = x["raw_wav"] if isinstance(x, dict) else x
            expected = wav.shape[0]
            prev = self.pooling
            self.pooling = pooling
            try:
                with torch.no_grad():
                    self.forward(wav, padding_mask)
            finally:
                self.pooling = prev
            order = self._hook_layers if self._hook_layers else list(self._hook_outputs.keys())
            embeddings = [self._hook_outputs[n] for n in order]
            if not embeddings:
                raise ValueError(f"No layers found matching: {self._hook_outputs.keys()}")
            embeddings = [e if e.shape[0] == expected else e.transpose(0, 1) for e in embeddings]
            return self._aggregate(embeddings, aggregation)
        finally:
            self._clear_hook_outputs()


Model = EATHFModel
