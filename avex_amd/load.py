"""``load_model`` / ``build_model`` / ``build_model_from_spec`` / ``load_label_mapping``.

Counterpart of avex/models/utils/load.py:35-311,521-570 and factory.py:56-166: a registered id, a
YAML path or a ``ModelSpec`` -> model instance with weights.  Decision order kept from the
reference: explicit ``checkpoint_path`` beats the registry default; a checkpoint forces
``pretrained=False``; feature mode is offered to a class iff its ``__init__`` has a parameter
literally named ``return_features_only``; without ``return_features_only`` the classifier width
comes from the checkpoint; constructor kwargs are filtered by the class signature.
"""
from __future__ import annotations

import inspect
import json
import logging
import os
from pathlib import Path
from typing import Optional, Union

from . import registry
from .base_model import ModelBase, coerce_audio_config
from .configs import ModelSpec
from .weights import classifier_out_features, is_remote, load_checkpoint_file, load_into

logger = logging.getLogger(__name__)


def build_model_from_spec(model_spec: ModelSpec, device: str, **kwargs: object) -> ModelBase:
    cls = registry.get_model_class(model_spec.name)
    if cls is None:
        raise KeyError(f"Model class '{model_spec.name}' is not registered. "
                       f"Available classes: {registry.list_model_classes()}")
    audio_config = coerce_audio_config(model_spec.audio_config)     # a reference ModelSpec carries the reference's own AudioConfig
    init = {"device": device, "audio_config": audio_config, **kwargs}
    for field in ModelSpec.FORWARDED:
        value = getattr(model_spec, field, None)
        if value is not None and value != "":
            init[field] = value
    accepted = set(inspect.signature(cls.__init__).parameters)
    return cls(**{k: v for k, v in init.items() if k in accepted})


def build_model(model_name: str, device: str, **kwargs: object) -> ModelBase:
    """Build a registered model id (architecture only, no checkpoint)."""
    spec = registry.get_model_spec(model_name)
    if spec is None:
        raise KeyError(f"Model '{model_name}' is not registered. Available models: {list(registry._MODEL_REGISTRY)}")
    return build_model_from_spec(spec.model_copy(deep=True), device, **kwargs)


def load_label_mapping(model_or_path: Union[str, Path]) -> Optional[dict]:
    """``{"label_to_index": ..., "index_to_label": ...}`` from a local JSON file or a registered id whose
    class-mapping path is local; ``None`` when unavailable (remote URIs are unreachable offline)."""
    path = str(model_or_path)
    if path in registry._MODEL_REGISTRY:
        path = registry.get_class_mapping_path(path) or ""
    if not path or is_remote(path) or not os.path.exists(path):
        return None
    with open(path) as f:
        label_to_index = json.load(f)
    if "label_to_index" in label_to_index:
        return label_to_index
    return {"label_to_index": label_to_index, "index_to_label": {v: k for k, v in label_to_index.items()}}


def _load_from_modelspec(spec: ModelSpec, device: str, checkpoint_path: Optional[str],
                         registry_key: Optional[str] = None, return_features_only: bool = False) -> ModelBase:
    spec = spec.model_copy(deep=True)
    spec.device = device
    if not checkpoint_path:
        key = registry_key
        if key is None:
            key = next((k for k, s in registry._MODEL_REGISTRY.items() if s.name == spec.name and s == spec), None)
        if key is not None:
            checkpoint_path = registry.get_checkpoint_path(key)
    if checkpoint_path:
        spec.pretrained = False
    cls = registry.get_model_class(spec.name)
    supports_features = cls is not None and "return_features_only" in inspect.signature(cls.__init__).parameters
    kwargs: dict = {}
    if return_features_only and supports_features:
        kwargs["return_features_only"] = True
    if checkpoint_path and not return_features_only:
        n = classifier_out_features(load_checkpoint_file(checkpoint_path))
        if n is None and registry_key is not None:
            mapping = load_label_mapping(registry_key)
            if mapping and "label_to_index" in mapping:
                n = len(mapping["label_to_index"])
        if n is not None:
            kwargs["num_classes"] = n
        elif supports_features:
            return_features_only = True
            kwargs["return_features_only"] = True
    if spec.pretrained and not checkpoint_path and supports_features and not return_features_only:
        return_features_only = True
        kwargs["return_features_only"] = True
    if not checkpoint_path and not return_features_only and not spec.pretrained and not supports_features:
        raise ValueError("load_model() without a checkpoint no longer creates new classifier heads. "
                         "Build a backbone with build_model()/build_model_from_spec() and attach a probe head instead.")
    model = build_model_from_spec(spec, device, **kwargs)
    if not return_features_only and registry_key is not None:
        mapping = load_label_mapping(registry_key)
        if mapping:
            model.label_mapping = mapping
    if checkpoint_path:
        load_into(model, checkpoint_path, keep_classifier=not return_features_only)
    return model.to(device)


def load_model(model: Union[str, Path, ModelSpec], device: str = "cpu", checkpoint_path: Optional[str] = None,
               return_features_only: bool = False) -> ModelBase:
    """Architecture + weights for a registered id, a YAML config path or a ``ModelSpec``."""
    if isinstance(model, Path):
        model = str(model)
    if isinstance(model, str):
        spec = registry.get_model_spec(model)
        if spec is not None:
            return _load_from_modelspec(spec, device, checkpoint_path, registry_key=model,
                                        return_features_only=return_features_only)
        if model.endswith((".yml", ".yaml")) or Path(model).exists():
            spec = registry.load_model_spec_from_yaml(model)
            key = Path(model).stem
            registry.register_model(key, spec)
            return _load_from_modelspec(spec, device, checkpoint_path, registry_key=key,
                                        return_features_only=return_features_only)
        raise ValueError(f"Unknown model identifier: '{model}'. Available models: {list(registry._MODEL_REGISTRY)}. "
                         "Or provide a path to a YAML config file.")
    if isinstance(model, ModelSpec):
        return _load_from_modelspec(model, device, checkpoint_path, return_features_only=return_features_only)
    raise TypeError(f"Unsupported model type: {type(model)}. Expected str, Path, or ModelSpec.")
