"""avex_amd — MI355X-native embedding-extraction path behind the AVEX plugin API.

Public surface mirrors ``import avex`` (reference: avex/__init__.py:11-51) for the hot path:
``load_model`` -> model with ``forward`` / ``extract_embeddings`` / ``register_hooks_for_layers``.
The arithmetic lives in ``avex_amd/lib/libavexhip.so`` (hand-written HIP for gfx950, C ABI in
``include/avexhip.h``); importing this package does not need a GPU, running a model does.
"""
from .configs import AudioConfig, ModelSpec
from .load import build_model, build_model_from_spec, load_label_mapping, load_model
from .registry import (describe_model, get_checkpoint_path, get_model_class, get_model_spec, list_model_classes,
                       list_model_layers, list_models, register_model, register_model_class)
from .base_model import ModelBase

__version__ = "0.1.0"

__all__ = [
    "load_model", "register_model", "get_model_spec", "list_models", "describe_model", "list_model_layers",
    "register_model_class", "get_model_class", "list_model_classes", "build_model", "build_model_from_spec",
    "get_checkpoint_path", "load_label_mapping", "ModelBase", "ModelSpec", "AudioConfig",
]
