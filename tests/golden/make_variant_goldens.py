"""Golden vectors for BEATsConfig options no official checkpoint uses (development container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_variant_goldens.py

Runs the REAL reference (`avex.models.beats.beats.BEATs`, imported from /root/reference through _ref_import.py) on the small
configurations of ``avex_amd.synth.BEATS_VARIANTS`` -- pre-LN blocks, the other FFN activations of ``get_activation_fn``, the gated
linear unit, a patch-embedding bias, no / ungated relative position bias, no post_extract_proj -- with the closed-form synthetic
weights, and stores the reference's outputs in variants.npz.  Only data is written.
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import numpy as np
import torch

from _ref_import import import_reference
from avex_amd import synth

avex = import_reference()
from avex.models.beats.beats import BEATs, BEATsConfig  # noqa: E402

torch.set_num_threads(8)
SEED_W, SEED_X = 3, 13
B, T = 2, 32000


def main():
    out = {}
    x = synth.noise_clips(B, T, seed=SEED_X)
    pm = np.zeros((B, T), bool)
    pm[1, T // 2:] = True
    for name, cfg in synth.BEATS_VARIANTS.items():
        m = BEATs(BEATsConfig(**cfg)).eval()
        sd = synth.beats_state_dict(cfg, seed=SEED_W)
        missing, unexpected = m.load_state_dict({k[len("backbone."):]: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()},
                                                strict=False)
        assert not unexpected, (name, unexpected)
        assert set(missing) <= {"fbank.window", "fbank.mel_fb"}, (name, missing)
        caps = {}

        def hook(key):
            def fn(mod, inp, o):
                caps[key] = o.detach().numpy().transpose(1, 0, 2).copy()      # (T, B, E) -> (B, T, E)
            return fn
        for i, layer in enumerate(m.encoder.layers):
            layer.fc2.register_forward_hook(hook(f"fc2.{i}"))
        with torch.no_grad():
            y, _ = m.extract_features(torch.from_numpy(x))
            taps = dict(caps)
            ym, _ = m.extract_features(torch.from_numpy(x), padding_mask=torch.from_numpy(pm))
        # every third token of the features, their mean over the tokens, every sixth token of the taps and their means
        out[f"{name}.features_tok3"] = y.numpy()[:, ::3]
        out[f"{name}.pooled"] = y.mean(dim=1).numpy()
        out[f"{name}.features_masked_tok3"] = ym.numpy()[:, ::3]
        for k, v in taps.items():
            out[f"{name}.{k}_tok6"] = v[:, ::6]
            out[f"{name}.{k}_mean"] = v.mean(axis=1)
        print(name, tuple(y.shape), "rms", float(y.pow(2).mean().sqrt()), "keys", len(sd))
    path = os.path.join(HERE, "variants.npz")
    np.savez_compressed(path, **out)
    print(f"wrote variants.npz: {os.path.getsize(path) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
