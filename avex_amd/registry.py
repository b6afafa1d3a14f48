"""Model registry: named ``ModelSpec`` s and model classes (the plugin mechanism).

Counterpart of avex/models/utils/registry.py: ``register_model`` / ``get_model_spec`` /
``list_models`` / ``describe_model`` / ``get_checkpoint_path`` for specs, ``register_model_class``
/ ``get_model_class`` / ``list_model_classes`` for classes (key = ``cls.name`` or the lower-cased class
name, registry.py:600-621), ``list_model_layers`` for probe targets.  Module-global dicts, populated
at import; like the reference there is no locking.
"""
from __future__ import annotations

import logging
from pathlib import Path
from typing import Dict, Optional, Type, Union

import yaml

from .base_model import ModelBase
from .configs import ModelSpec
from .official_models import OFFICIAL_MODELS

logger = logging.getLogger(__name__)

_MODEL_REGISTRY: Dict[str, ModelSpec] = {}
_MODEL_CLASSES: Dict[str, Type[ModelBase]] = {}
_CHECKPOINT_PATHS: Dict[str, Optional[str]] = {}
_CLASS_MAPPING_PATHS: Dict[str, Optional[str]] = {}


# ------------------------------------------------------------------ specs
def load_model_spec_from_yaml(yaml_path: Union[str, Path]) -> ModelSpec:
    """Read ``model_spec:`` (or a bare spec mapping) from a YAML file."""
    path = Path(yaml_path)
    if not path.exists():
        raise FileNotFoundError(f"Config file not found: {path}")
    with open(path) as f:
        data = yaml.safe_load(f) or {}
    if not isinstance(data, dict):
        raise ValueError(f"{path}: expected a mapping at the top level")
    spec = data.get("model_spec", data)
    if not isinstance(spec, dict) or "name" not in spec:
        raise ValueError(f"{path}: no model_spec with a 'name' found")
    return ModelSpec(**spec)


def register_model(name: str, model_spec: ModelSpec, checkpoint_path: Optional[str] = None,
                   class_mapping_path: Optional[str] = None) -> None:
    if name in _MODEL_REGISTRY:
        logger.warning("Model '%s' is already registered. Overwriting with new configuration.", name)
    _MODEL_REGISTRY[name] = model_spec
    _CHECKPOINT_PATHS[name] = checkpoint_path
    _CLASS_MAPPING_PATHS[name] = class_mapping_path


def get_model_spec(name: str) -> Optional[ModelSpec]:
    return _MODEL_REGISTRY.get(name)


def get_checkpoint_path(name: str) -> Optional[str]:
    if name not in _MODEL_REGISTRY:
        raise KeyError(f"Model '{name}' is not registered")
    return _CHECKPOINT_PATHS.get(name) or None


def get_class_mapping_path(name: str) -> Optional[str]:
    if name not in _MODEL_REGISTRY:
        raise KeyError(f"Model '{name}' is not registered")
    return _CLASS_MAPPING_PATHS.get(name) or None


def _summary(name: str, spec: ModelSpec) -> dict:
    variant = ""
    if spec.name == "beats":
        variant = "NatureLM" if spec.use_naturelm else ("fine-tuned" if spec.fine_tuned else "ssl")
    return {
        "description": f"{spec.name} ({variant})" if variant else spec.name,
        "model_type": spec.name,
        "has_trained_classifier": bool(_CLASS_MAPPING_PATHS.get(name)),
        "checkpoint_path": _CHECKPOINT_PATHS.get(name),
        "class_available": spec.name in _MODEL_CLASSES,
    }


def list_models(verbose: bool = False) -> Dict[str, dict]:
    """Registered model ids -> summary dict (``verbose`` prints a table like the reference does)."""
    out = {n: _summary(n, s) for n, s in sorted(_MODEL_REGISTRY.items())}
    if verbose:
        print(f"{'Model Name':<36}{'Type':<16}{'Classifier':<12}Built here")
        for n, d in out.items():
            print(f"{n:<36}{d['model_type']:<16}{'yes' if d['has_trained_classifier'] else 'no':<12}"
                  f"{'yes' if d['class_available'] else 'no'}")
    return out


def describe_model(name: str, verbose: bool = False) -> dict:
    spec = get_model_spec(name)
    if spec is None:
        raise KeyError(f"Model '{name}' is not registered")
    d = _summary(name, spec)
    d["model_spec"] = spec.model_dump()
    d["class_mapping_path"] = _CLASS_MAPPING_PATHS.get(name)
    if verbose:
        print(f"{name}: {d['description']}\n  checkpoint: {d['checkpoint_path']}\n  audio: {spec.audio_config}")
    return d


# ------------------------------------------------------------------ classes
def register_model_class(cls: Type) -> Type:
    """Register a ``ModelBase`` subclass (usable as a decorator)."""
    key = getattr(cls, "name", None) or cls.__name__.lower()
    if key in _MODEL_CLASSES:
        logger.warning("Model class '%s' is already registered, overwriting.", key)
    _MODEL_CLASSES[key] = cls
    return cls


def get_model_class(name: str) -> Optional[Type[ModelBase]]:
    return _MODEL_CLASSES.get(name)


def list_model_classes() -> list[str]:
    return list(_MODEL_CLASSES)


def list_model_layers(model: Union[str, ModelBase], device: str = "cpu") -> Dict[str, Union[list, str]]:
    if isinstance(model, str):
        if get_model_spec(model) is None:
            raise ValueError(f"Model '{model}' not found in registry. Available models: {list(_MODEL_REGISTRY)}")
        from .load import build_model_from_spec
        inst = build_model_from_spec(get_model_spec(model).model_copy(deep=True), device, return_features_only=True)
    else:
        inst = model
    if not isinstance(inst, ModelBase):
        raise ValueError(f"Model must be an instance of ModelBase, got {type(inst)}")
    inst._discover_embedding_layers()
    layers = list(inst._layer_names)
    last = inst._get_last_non_classification_layer() or (layers[-1] if layers else "")
    return {"layers": layers, "last_layer": last, "all": layers, "special_options": ["last_layer", "all"]}


# ------------------------------------------------------------------ bootstrap
def initialize_registry() -> None:
    for name, entry in OFFICIAL_MODELS.items():
        register_model(name, ModelSpec(**entry["model_spec"]), entry.get("checkpoint_path"),
                       entry.get("class_mapping_path"))
    from .beats_model import Model as BeatsModel
    _MODEL_CLASSES.setdefault("beats", BeatsModel)
    from .aves_model import Model as AvesModel
    _MODEL_CLASSES.setdefault("aves", AvesModel)
    from .efficientnet import Model as EfficientNetModel
    _MODEL_CLASSES.setdefault("efficientnet", EfficientNetModel)
    from .eat_hf import EATHFModel
    _MODEL_CLASSES.setdefault("eat_hf", EATHFModel)        # key = cls.name; the reference's module scan also lists the lower-cased
    _MODEL_CLASSES.setdefault("eathf", EATHFModel)         # class name ("eat_hf, eathf")


initialize_registry()
