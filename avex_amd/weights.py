"""Checkpoint ingest for the plugin loader: local ``.safetensors`` / ``.pt`` files -> flat state dict.

Format behaviour follows the reference (avex/utils/utils.py:350-469 ``universal_torch_load``,
:509-570 ``_process_state_dict``; avex/models/utils/load.py:521-570 ``_load_checkpoint``):
safetensors files are flat tensor dicts, ``.pt`` files may wrap the weights under
``"model_state_dict"`` or ``"model"``; ``module.`` / ``model.`` prefixes are stripped, classifier
keys dropped unless asked for, and the ``backbone.`` prefix is added or removed to match the target.
Remote URIs (``hf://``, ``gs://``, ``s3://``, ``r2://``) are out of scope: neither box has a network.
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import torch

_REMOTE = ("hf://", "gs://", "s3://", "r2://", "http://", "https://")
_CLASSIFIER_TERMS = ("classifier", "head", "classification", "classification_head")


def is_remote(path: str) -> bool:
    return str(path).startswith(_REMOTE)


def load_checkpoint_file(path: str, map_location: str = "cpu") -> Dict[str, object]:
    path = str(path)
    if is_remote(path):
        raise FileNotFoundError(
            f"Checkpoint not found: {path} (remote checkpoint URIs are not reachable offline; "
            "download the file and pass checkpoint_path=<local file>)")
    if not os.path.exists(path):
        raise FileNotFoundError(f"Checkpoint not found: {path}")
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return {"model_state_dict": load_file(path, device="cpu")}
    return torch.load(path, map_location=map_location, weights_only=False)


def process_state_dict(state_dict: Dict[str, object], keep_classifier: bool = False,
                       drop_model_prefix: bool = True) -> Dict[str, torch.Tensor]:
    if "model_state_dict" in state_dict:
        state_dict = state_dict["model_state_dict"]
    elif "model" in state_dict and isinstance(state_dict["model"], dict):
        state_dict = state_dict["model"]
    out: Dict[str, torch.Tensor] = {}
    for key, value in state_dict.items():
        if key.startswith("module."):
            key = key[len("module."):]
        elif drop_model_prefix and key.startswith("model."):
            key = key[len("model."):]
        if not keep_classifier and any(t in key.lower() for t in _CLASSIFIER_TERMS):
            continue
        out[key] = value
    return out


def classifier_out_features(state_dict: Dict[str, object]) -> Optional[int]:
    """``num_classes`` implied by a checkpoint's classifier weight, if it has one
    (reference: load.py:314-420)."""
    sd = state_dict.get("model_state_dict", state_dict.get("model", state_dict))
    if not isinstance(sd, dict):
        return None
    for key in ("classifier.weight", "module.classifier.weight", "model.classifier.weight", "model.classifier.1.weight"):
        if key in sd and hasattr(sd[key], "shape"):
            return int(sd[key].shape[0])
    return None


def load_into(model: torch.nn.Module, checkpoint_path: str, keep_classifier: bool = False) -> None:
    ckpt = load_checkpoint_file(checkpoint_path)
    target = list(model.state_dict().keys())
    sd = process_state_dict(ckpt, keep_classifier=keep_classifier,
                            drop_model_prefix=not any(k.startswith("model.") for k in target))
    tgt_bb = any(k.startswith("backbone.") for k in target)
    src_bb = any(k.startswith("backbone.") for k in sd)
    if tgt_bb and not src_bb:
        sd = {f"backbone.{k}": v for k, v in sd.items()}
    elif src_bb and not tgt_bb:
        sd = {k[len("backbone."):] if k.startswith("backbone.") else k: v for k, v in sd.items()}
    model.load_state_dict(sd, strict=False)
