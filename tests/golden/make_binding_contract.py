"""Drive INTEGRATION.md section 2 VERBATIM against the REAL reference (development container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_binding_contract.py [--check]

The reference-side module of INTEGRATION.md section 2 is cut out of that file and executed as is; then the reference's OWN
registry / factory / loader / probe factory are run on the four HIP classes:

    register_model_class            avex/models/utils/registry.py:600-621
    build_model_from_spec           avex/models/utils/factory.py:108-166   (passes avex.configs.AudioConfig)
    load_model(spec, checkpoint_path=<local .safetensors>)   load.py:35-311, _load_checkpoint :521-570
    list_model_layers(model)        registry.py:660-710                     (isinstance check against the reference's ModelBase)
    register_hooks_for_layers([0, -1])
    build_probe_from_config(cfg, num_classes, device, base_model=m)   probes/utils/factory.py:56-186, base_probes.py:23-195

There is no GPU here, so the one thing that cannot happen is the encoder forward: ``extract_embeddings`` of each model instance is
replaced by a recorder that notes the call (dummy shape, keywords) and answers zeros of the documented shape.  Everything else is
the real code on both sides.  The outcome is written to ``binding_contract.json`` (data only); ``tests/test_binding_contract.py``
asserts it (and re-runs this script when /root/reference is present), ``tests/test_gpu_binding.py`` replays the recorded call
sequence on the GPU with real forwards.  ``--check`` compares a fresh run with the committed fixture instead of writing it.
"""
import functools
import inspect
import json
import os
import re
import sys
import tempfile

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import numpy as np
import torch

from _ref_import import import_reference

FIXTURE = os.path.join(HERE, "binding_contract.json")


def integration_block() -> str:
    """The python block of INTEGRATION.md that starts with the reference-side module's path comment."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"```python\n(# avex/models/hip_backends\.py.*?)```", text, re.S)
    if not m:
        raise RuntimeError("INTEGRATION.md section 2: reference-side module block not found")
    return m.group(1)


def _jsonable(v):
    if hasattr(v, "model_dump"):
        return {"__type__": f"{type(v).__module__}.{type(v).__name__}", **v.model_dump()}
    if isinstance(v, dict):
        return {k: _jsonable(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_jsonable(x) for x in v]
    if isinstance(v, (str, int, float, bool)) or v is None:
        return v
    return repr(v)


def _record_init(cls, log):
    """Wrap cls.__init__ so the keywords the reference's factory passes are recorded (inspect.signature follows __wrapped__)."""
    orig = cls.__init__

    @functools.wraps(orig)
    def init(self, *a, **kw):
        log.append({k: _jsonable(v) for k, v in kw.items()})
        return orig(self, *a, **kw)
    cls.__init__ = init


def _save_safetensors(sd, path):
    from safetensors.numpy import save_file
    save_file({k: np.ascontiguousarray(v) for k, v in sd.items()}, path)


def _equal_tensors(model, sd):
    """How many checkpoint tensors sit, value for value, in the model's state_dict under the same key."""
    have = model.state_dict()
    n = 0
    for k, v in sd.items():
        if k in have and tuple(have[k].shape) == tuple(v.shape) and np.array_equal(have[k].cpu().numpy(), v):
            n += 1
    return n


def _probe_step(avex, model, probe_kwargs, embed_dim, tokens):
    """build_probe_from_config(..., base_model=model) with the model's forward stood in for by a recorder."""
    from avex.configs import ProbeConfig
    from avex.models.probes.utils.factory import build_probe_from_config
    calls = []

    def recorder(x, **kw):
        wav = x["raw_wav"] if isinstance(x, dict) else x
        calls.append({"input_shape": list(wav.shape), "input_dtype": str(wav.dtype).replace("torch.", ""),
                      "keywords": {k: _jsonable(v) for k, v in kw.items()}, "hooks_registered": list(model._hook_layers)})
        n = len(model._hook_layers)
        if kw.get("aggregation", "none") == "none":
            out = [torch.zeros(wav.shape[0], tokens, embed_dim) for _ in range(n)]
            return out[0] if n == 1 else out
        return torch.zeros(wav.shape[0], embed_dim * n)
    model.extract_embeddings = recorder                       # instance attribute: the class stays as it is
    cfg = ProbeConfig(**probe_kwargs)
    probe = build_probe_from_config(cfg, num_classes=5, device="cpu", base_model=model)
    first = len(calls)
    logits = probe(torch.zeros(2, calls[0]["input_shape"][1]))                 # _BaseProbe._get_embeddings path (base_probes.py:167-195)
    rec = {"probe_config": probe_kwargs, "probe_class": type(probe).__name__, "resolved_layers": list(probe.layers),
           "construction_calls": calls[:first], "forward_calls": calls[first:], "logits_shape": list(logits.shape),
           "model_training_after": bool(model.training),
           "requires_grad_any": any(p.requires_grad for p in model.parameters())}
    head = getattr(probe, "classifier", None)
    if isinstance(head, torch.nn.Linear):
        rec["inferred_dim"] = head.in_features
    del model.extract_embeddings
    model.deregister_all_hooks()
    return rec


def run() -> dict:
    avex = import_reference()
    import avex.models.utils.registry as ref_registry
    from avex.configs import AudioConfig as RefAudioConfig, ModelSpec as RefModelSpec
    from avex.models.base_model import ModelBase as RefModelBase
    from avex_amd import synth

    block = integration_block()
    ns = {"__name__": "avex.models.hip_backends"}
    exec(compile(block, "INTEGRATION.md#section2", "exec"), ns)              # noqa: S102  (the documented module, verbatim)

    out = {"reference_version": avex.__version__, "integration_block_sha256": __import__("hashlib").sha256(block.encode()).hexdigest(),
           "registered_keys": sorted(k for k in ref_registry.list_model_classes() if k.endswith("_hip")), "families": {}}

    beats_spec = avex.get_model_spec("esp_aves2_sl_beats_all").model_copy(deep=True)
    eat_spec = avex.get_model_spec("esp_aves2_eat_all").model_copy(deep=True)
    eff_spec = avex.get_model_spec("esp_aves2_effnetb0_all").model_copy(deep=True)
    aves_spec = RefModelSpec(name="aves", pretrained=False, device="cpu",
                             audio_config=RefAudioConfig(sample_rate=16000, representation="raw", normalize=False, target_length_seconds=10))
    families = [
        # key, spec, checkpoint generator, embed dim of a hooked layer, tokens of a 10 s clip, features-mode keyword accepted
        ("beats_hip", beats_spec, lambda: synth.beats_state_dict(synth.BEATS_BASE_CFG, seed=0), 768, 496),
        ("eat_hf_hip", eat_spec, lambda: synth.eat_state_dict(synth.EAT_BASE_CFG, seed=0), 768, 513),
        ("efficientnet_hip", eff_spec, lambda: synth.effnet_b0_state_dict(seed=0), None, None),
        ("aves_hip", aves_spec, lambda: synth.aves_state_dict(synth.AVES_BASE_CFG, seed=0), 768, 499),
    ]
    tmp = tempfile.mkdtemp(prefix="binding_contract_")
    for key, spec, make_sd, dim, tokens in families:
        cls = ref_registry.get_model_class(key)
        fam = {"class": f"{cls.__module__}.{cls.__qualname__}", "mro": [f"{c.__module__}.{c.__qualname__}" for c in cls.__mro__[:4]],
               "spec_source": spec.name}
        spec = spec.model_copy(deep=True)
        spec.name = key
        init_log = []
        _record_init(cls, init_log)
        supports = "return_features_only" in inspect.signature(cls.__init__).parameters        # what load.py:216-220 looks at
        fam["supports_return_features_only"] = supports
        runtime = {"return_features_only": True} if supports else {"num_classes": None}

        # ---- build_model_from_spec (factory.py:108-166)
        m = avex.build_model_from_spec(spec, "cpu", **runtime)
        fam["build"] = {"init_keywords": init_log[-1], "isinstance_reference_ModelBase": isinstance(m, RefModelBase),
                        "device_attr": m.device, "training": bool(m.training),
                        "audio_processor": type(m.audio_processor).__name__ if m.audio_processor is not None else None,
                        "state_dict_keys": len(m.state_dict())}
        ap = m.audio_processor
        fam["build"]["target_length_samples"] = (int(ap.target_length_seconds * ap.sr) if getattr(ap, "target_length_seconds", None)
                                                 else int(getattr(ap, "target_length", 0)) or None)

        # ---- load_model(spec, checkpoint_path=local) (load.py:35-311,521-570)
        sd = make_sd()
        ck = os.path.join(tmp, key + ".safetensors")
        _save_safetensors(sd, ck)
        lm = avex.load_model(spec.model_copy(deep=True), device="cpu", checkpoint_path=ck, return_features_only=True)
        fam["load"] = {"checkpoint_tensors": len(sd), "model_state_dict_keys": len(lm.state_dict()),
                       "tensors_equal_after_load": _equal_tensors(lm, sd), "init_keywords": init_log[-1],
                       "class": type(lm).__name__}
        os.remove(ck)

        # ---- list_model_layers / layer map / hook resolution
        info = ref_registry.list_model_layers(lm)
        fam["layers"] = {"layers": info["layers"], "last_layer": info["last_layer"], "special_options": info["special_options"]}
        fam["hooks_0_m1"] = lm.register_hooks_for_layers([0, -1])
        fam["hooks_all_count"] = len(lm.register_hooks_for_layers(["all"]))
        fam["hooks_last_layer"] = lm.register_hooks_for_layers(["last_layer"])
        lm.deregister_all_hooks()

        # ---- probes (probes/utils/factory.py:56-186; base_probes.py:23-195)
        if dim is None:                                     # EfficientNet: the last hooked conv gives (B, 1280, 4, 32) for a 10 s clip; "mean" is
            dim, tokens = 1280 * 4, 1                       # mean(dim=-1) then view(B, -1) (efficientnet.py:297-311): 1280 x 4 columns
        fam["probe_linear_mean"] = _probe_step(avex, lm, dict(probe_type="linear", target_layers=["last_layer"], aggregation="mean",
                                                             freeze_backbone=True), dim, tokens)
        if key in ("beats_hip", "aves_hip"):
            fam["probe_linear_two_layers"] = _probe_step(avex, lm, dict(probe_type="linear", target_layers=[fam["hooks_0_m1"][0], fam["hooks_0_m1"][1]],
                                                                       aggregation="mean", freeze_backbone=True), dim, tokens)
            fam["probe_attention_none"] = _probe_step(avex, lm, dict(probe_type="attention", target_layers=["last_layer"], aggregation="none",
                                                                    num_heads=4, attention_dim=128, num_layers=1, input_processing="sequence",
                                                                    freeze_backbone=True), dim, tokens)
        out["families"][key] = fam

    # ---- the replacing form: the HIP class under the built-in's key, an official id loaded through it
    cls = ref_registry.get_model_class("beats_hip")
    builtin = ref_registry.get_model_class("beats")
    Replacing = type("Model", (cls,), {"name": "beats"})
    ref_registry.register_model_class(Replacing)
    sd = synth.beats_state_dict(synth.BEATS_BASE_CFG, seed=0)
    ck = os.path.join(tmp, "official.safetensors")
    _save_safetensors(sd, ck)
    om = avex.load_model("esp_aves2_sl_beats_all", device="cpu", checkpoint_path=ck, return_features_only=True)
    out["official_id_through_replaced_class"] = {"id": "esp_aves2_sl_beats_all", "class_mro_has_hip": any(c.__module__ == "avex_amd.beats_model" for c in type(om).__mro__),
                                                 "tensors_equal_after_load": _equal_tensors(om, sd), "layers": ref_registry.list_model_layers(om)["layers"]}
    ref_registry._MODEL_CLASSES["beats"] = builtin            # leave the reference's registry as it was
    os.remove(ck)
    os.rmdir(tmp)
    return out


def main() -> int:
    out = run()
    text = json.dumps(out, indent=1, sort_keys=True) + "\n"
    if "--check" in sys.argv:
        want = open(FIXTURE).read()
        if text != want:
            sys.stderr.write("binding contract differs from the committed fixture\n")
            return 1
        print("binding contract reproduces the committed fixture")
        return 0
    with open(FIXTURE, "w") as f:
        f.write(text)
    print(f"wrote {FIXTURE}: {len(text)} bytes")
    return 0


if __name__ == "__main__":
    sys.exit(main())
