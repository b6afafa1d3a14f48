"""Generate golden vectors by running the REAL reference (development container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_goldens.py

Imports avex from /root/reference (see _ref_import.py), loads the closed-form synthetic
weights of ``avex_amd.synth`` into it and records inputs' seeds + the reference's outputs as
small ``.npz/.json`` fixtures in this directory.  Only data is written; no reference source
travels.  The oracle (oracle/beats_oracle.py) and the HIP path are both checked against
these files.
"""
import json
import os
import sys
import time

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import numpy as np
import torch

from _ref_import import import_reference
from avex_amd import synth

avex = import_reference()
from avex.models.beats.beats import BEATs, BEATsConfig, _BatchedFbank  # noqa: E402

torch.set_num_threads(8)
OUT = HERE


def to_torch_sd(sd):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}


def save(name, **arrays):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrays)
    print(f"wrote {name}: {os.path.getsize(path)/1024:.1f} KiB")


# ----------------------------------------------------------------------------- 1. frontend
def gen_fbank():
    fb = _BatchedFbank()
    out = {}
    out["window"] = fb.window.numpy()
    mel = fb.mel_fb.numpy()
    nz = np.nonzero(mel)
    out["mel_nz_rows"] = nz[0].astype(np.int16); out["mel_nz_cols"] = nz[1].astype(np.int16)
    out["mel_nz_vals"] = mel[nz]
    # (a) noise, T=16000, B=2 : full tensor
    x = synth.noise_clips(2, 16000, seed=0)
    out["noise16k"] = fb(torch.from_numpy(x) * 2 ** 15).numpy()
    # (b) tones, T=16000, B=6 : full tensor
    x = synth.tone_clips(16000)
    out["tone16k"] = fb(torch.from_numpy(x) * 2 ** 15).numpy()
    # (c) noise, T=160000, B=2 : every 37th frame + per-frame sums
    x = synth.noise_clips(2, 160000, seed=0)
    y = fb(torch.from_numpy(x) * 2 ** 15).numpy()
    out["noise160k_rows37"] = y[:, ::37]
    out["noise160k_framesum"] = y.sum(-1)
    # (d) other mel sizes / lengths exercised by the reference's own test (test_batched_fbank.py:52-91)
    for nm in (64, 256):
        fbm = _BatchedFbank(num_mel_bins=nm)
        x = synth.noise_clips(1, 4000, seed=7)
        out[f"noise4k_mel{nm}"] = fbm(torch.from_numpy(x) * 2 ** 15).numpy()
    # (e) impulse at sample 200 and DC input (Appendix B known answers)
    imp = np.zeros((1, 16000), np.float32); imp[0, 200] = 1.0
    out["impulse200"] = fb(torch.from_numpy(imp) * 2 ** 15).numpy()[:, :3]
    save("fbank.npz", **out)


# ----------------------------------------------------------------------------- 2. tiny config, per stage
def gen_tiny():
    cfg = BEATsConfig(**synth.BEATS_TINY_CFG)
    m = BEATs(cfg).eval()
    sd = synth.beats_state_dict(synth.BEATS_TINY_CFG, seed=1)
    missing, unexpected = m.load_state_dict({k[len("backbone."):]: v for k, v in to_torch_sd(sd).items()}, strict=False)
    assert not unexpected, unexpected
    assert set(missing) <= {"fbank.window", "fbank.mel_fb"}, missing
    x = synth.noise_clips(2, 16000, seed=3)
    caps = {}

    def hook(name):
        def fn(mod, inp, out):
            caps[name] = (out[0] if isinstance(out, tuple) else out).detach().numpy().copy()
        return fn
    m.layer_norm.register_forward_hook(hook("stage.patch_ln"))
    m.post_extract_proj.register_forward_hook(hook("backbone.post_extract_proj"))
    m.encoder.pos_conv.register_forward_hook(hook("stage.pos_conv"))
    m.encoder.layer_norm.register_forward_hook(hook("stage.enc_in"))
    for i, layer in enumerate(m.encoder.layers):
        layer.self_attn.register_forward_hook(hook(f"stage.attn{i}"))
        layer.fc2.register_forward_hook(hook(f"backbone.encoder.layers.{i}.fc2"))
        layer.register_forward_hook(hook(f"stage.layer{i}"))
    with torch.no_grad():
        fbk = m.preprocess(torch.from_numpy(x)).numpy()
        y, _ = m(torch.from_numpy(x))
        # padded variant: second half of clip 1 padded
        pm = torch.zeros(2, 16000, dtype=torch.bool); pm[1, 8000:] = True
        caps_nomask = dict(caps)
        ym, fm = m(torch.from_numpy(x), padding_mask=pm)
    out = {"features": y.numpy(), "stage.fbank": fbk, "features_masked": ym.numpy(), "frame_mask": fm.numpy()}
    for k, v in caps_nomask.items():
        if v.ndim == 3 and v.shape[0] != 2:       # (T,B,E) -> (B,T,E)
            v = v.transpose(1, 0, 2)
        if k == "stage.pos_conv":                  # (B,C,T) -> (B,T,C)
            v = v.transpose(0, 2, 1)
        out[k] = v
    save("tiny_stages.npz", **out)


# ----------------------------------------------------------------------------- 3-6. BEATs-base through the public API
def gen_base():
    spec = avex.get_model_spec("esp_aves2_sl_beats_all").model_copy(deep=True)
    model = avex.build_model_from_spec(spec, device="cpu", return_features_only=True).eval()
    sd = synth.beats_state_dict(synth.BEATS_BASE_CFG, seed=0)
    missing, unexpected = model.load_state_dict(to_torch_sd(sd), strict=False)
    assert not unexpected, unexpected
    assert set(missing) <= {"backbone.fbank.window", "backbone.fbank.mel_fb"}, missing
    model.disable_layerdrop = True

    meta = {}
    meta["layer_map"] = {str(k): v for k, v in model.get_model_layer_map().items()}
    meta["resolve_[0,-1]"] = model.register_hooks_for_layers([0, -1])
    meta["resolve_all"] = model.register_hooks_for_layers(["all"])
    meta["resolve_last_layer"] = model.register_hooks_for_layers(["last_layer"])
    meta["resolve_mixed"] = model.register_hooks_for_layers([3, "backbone.encoder.layers.2.fc2", "all", -1])
    errs = {}
    for label, arg in (("index_oob", [13]), ("neg_oob", [-14]), ("bool", [True]), ("unknown", ["backbone.nope"])):
        try:
            model.register_hooks_for_layers(arg)
            errs[label] = "no error"
        except Exception as e:  # noqa: BLE001
            errs[label] = type(e).__name__
    meta["errors"] = errs
    model.deregister_all_hooks()
    try:
        model.extract_embeddings(torch.zeros(1, 16000), aggregation="mean")
        meta["errors"]["no_hooks"] = "no error"
    except Exception as e:  # noqa: BLE001
        meta["errors"]["no_hooks"] = type(e).__name__

    out = {}
    names = model.register_hooks_for_layers(["all"])
    timings = {}
    for tag, B, T in (("b1", 1, 160000), ("b4", 4, 160000), ("odd", 2, 123457), ("short", 3, 16000)):
        x = torch.from_numpy(synth.noise_clips(B, T, seed=0))
        with torch.no_grad():
            t0 = time.time()
            feats = model(x)
            timings[tag] = time.time() - t0
            emb = model.extract_embeddings(x, aggregation="mean")
        out[f"{tag}.pooled"] = feats.mean(dim=1).numpy()
        out[f"{tag}.all_mean"] = emb.numpy()
        out[f"{tag}.feat_tok16"] = feats[:, ::16].numpy()
        print(tag, tuple(feats.shape), tuple(emb.shape), f"{timings[tag]:.2f}s")
    # tones (reference regression inputs), 1 s clips
    x = torch.from_numpy(synth.tone_clips(16000))
    with torch.no_grad():
        out["tone.pooled"] = model(x).mean(dim=1).numpy()
    # aggregation variants on [0,-1]
    model.register_hooks_for_layers([0, -1])
    x = torch.from_numpy(synth.noise_clips(2, 32000, seed=5))
    with torch.no_grad():
        for agg in ("mean", "max", "cls_token"):
            out[f"agg.{agg}"] = model.extract_embeddings(x, aggregation=agg).numpy()
        lst = model.extract_embeddings(x, aggregation="none")
        meta["agg_none_shapes"] = [list(t.shape) for t in lst]
        out["agg.none0_tok8"] = lst[0][:, ::8].numpy(); out["agg.none1_tok8"] = lst[1][:, ::8].numpy()
        # dict input with padding mask: second half of clip 1 padded (Collater schema, data/dataset.py:393-399)
        pm = torch.zeros(2, 32000, dtype=torch.bool); pm[1, 16000:] = True
        out["mask.mean"] = model.extract_embeddings({"raw_wav": x, "padding_mask": pm}, aggregation="mean").numpy()
        out["mask.features_tok8"] = model(x, pm)[:, ::8].numpy()
    # bucket function for T=496 (Toeplitz: bucket depends on j-i only)
    attn0 = model.backbone.encoder.layers[0].self_attn
    rel = torch.arange(-495, 496, dtype=torch.long)[None, :]
    out["bucket_rel_-495..495"] = attn0._relative_positions_bucket(rel, bidirectional=True).numpy()[0].astype(np.int16)
    meta["cpu_reference_seconds"] = timings
    meta["cpu_reference_host"] = {"threads": torch.get_num_threads(), "torch": torch.__version__}
    save("base_api.npz", **out)
    with open(os.path.join(OUT, "base_api.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print(json.dumps(meta["errors"]))

    # 7. load_model() with a local safetensors checkpoint == direct state-dict load
    try:
        from safetensors.numpy import save_file
        path = "/tmp/_synth_beats.safetensors"
        save_file({k: np.ascontiguousarray(v) for k, v in sd.items()}, path)
        m2 = avex.load_model("esp_aves2_sl_beats_all", device="cpu", checkpoint_path=path, return_features_only=True).eval()
        x = torch.from_numpy(synth.noise_clips(1, 160000, seed=0))
        with torch.no_grad():
            d = (m2(x).mean(1).numpy() - out["b1.pooled"])
        print("load_model path max|diff| vs state-dict path:", float(np.abs(d).max()))
        os.remove(path)
    except Exception as e:  # noqa: BLE001
        print("load_model check skipped:", repr(e))


if __name__ == "__main__":
    which = sys.argv[1:] or ["fbank", "tiny", "base"]
    if "fbank" in which:
        gen_fbank()
    if "tiny" in which:
        gen_tiny()
    if "base" in which:
        gen_base()
