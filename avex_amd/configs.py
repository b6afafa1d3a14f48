"""Boundary config types of the plugin API: ``AudioConfig`` and ``ModelSpec``.

Same field names, defaults and strictness (``extra="forbid"``) as the reference's pydantic
schemas for the fields the embedding path consumes (reference: avex/configs.py:170-228
``AudioConfig``, :231-372 ``ModelSpec``), so a reference YAML ``model_spec:`` block validates
unchanged.  Training / evaluation / probe schemas are out of scope.
"""
from __future__ import annotations

from typing import Any, ClassVar, Literal, Optional

from pydantic import BaseModel, ConfigDict, Field, field_validator


class AudioConfig(BaseModel):
    sample_rate: int = 16000
    n_fft: int = 2048
    hop_length: Optional[int] = None
    win_length: Optional[int] = None
    window: Literal["hann", "hamming"] = "hann"
    n_mels: int = 128
    representation: Literal["spectrogram", "mel_spectrogram", "raw"] = "mel_spectrogram"
    normalize: bool = True
    target_length_seconds: Optional[int] = None
    window_selection: Literal["random", "center"] = "random"
    center: bool = True
    extra_config: Optional[dict[str, Any]] = None

    model_config = ConfigDict(extra="forbid")

    @field_validator("sample_rate", "n_fft", "hop_length", "win_length", "n_mels", "target_length_seconds")
    @classmethod
    def _positive(cls, v: Optional[int]) -> Optional[int]:
        if v is not None and v <= 0:
            raise ValueError(f"Value must be positive, got {v}")
        return v


class ModelSpec(BaseModel):
    """Everything needed to instantiate a registered model class."""

    name: str
    pretrained: bool = True
    device: str = "cuda"
    audio_config: Optional[AudioConfig] = None
    text_model_name: Optional[str] = None
    projection_dim: Optional[int] = None
    temperature: Optional[float] = None
    eat_cfg: Optional[dict[str, Any]] = None
    extra_config: Optional[dict[str, Any]] = None
    pretraining_mode: Optional[bool] = None
    handle_padding: Optional[bool] = None
    fairseq_weights_path: Optional[str] = None
    eat_norm_mean: Optional[float] = -4.268
    eat_norm_std: Optional[float] = 4.569
    efficientnet_variant: Literal["b0", "b1"] = "b0"
    use_naturelm: Optional[bool] = None
    fine_tuned: Optional[bool] = None
    init_config: Optional[dict[str, Any]] = Field(None, description="BEATsConfig mapping for checkpoints without one")
    language: Optional[str] = None
    model_id: Optional[str] = "worstchan/EAT-base_epoch30_pretrain"

    model_config = ConfigDict(extra="forbid")

    @field_validator("device")
    @classmethod
    def _device(cls, v: str) -> str:
        base = v.split(":")[0]
        if base not in ("cpu", "cuda", "mps", "hip"):
            raise ValueError(f"Invalid device '{v}'")
        return v

    # spec fields forwarded to a model class __init__ when it declares a parameter of that name
    # (reference: avex/models/utils/factory.py:30-46)
    FORWARDED: ClassVar[tuple] = ("text_model_name", "projection_dim", "temperature", "eat_cfg", "pretraining_mode",
                        "handle_padding", "fairseq_weights_path", "eat_norm_mean", "eat_norm_std",
                        "efficientnet_variant", "use_naturelm", "fine_tuned", "init_config", "language", "model_id")
