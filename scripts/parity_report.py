#!/usr/bin/env python3
"""Print measured parity numbers (GPU path vs committed reference goldens) for DESIGN.md."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from _util import rel_l2, max_abs
from avex_amd import synth, kernels as K

g = np.load(os.path.join(ROOT, "tests/golden/base_api.npz"))
fb = np.load(os.path.join(ROOT, "tests/golden/fbank.npz"))
out = {}
plan = K.FbankPlan()
y = plan(torch.from_numpy(synth.noise_clips(2, 16000, seed=0)).cuda()).cpu().numpy()
out["fbank.noise16k.max_abs"] = max_abs(y, fb["noise16k"])
y = plan(torch.from_numpy(synth.noise_clips(2, 160000, seed=0)).cuda()).cpu().numpy()
out["fbank.noise160k.max_abs"] = max_abs(y[:, ::37], fb["noise160k_rows37"])
sd = synth.beats_state_dict(synth.BEATS_BASE_CFG, seed=0)
for dt, residual in (("f16", "half"), ("bf16", "half"), ("f16", "f32")):
    enc = K.BeatsEncoder(synth.BEATS_BASE_CFG, sd, operand_dtype=dt, residual=residual)
    dt = dt if residual == "half" else dt + "_residual_f32"      # (the model classes run this mode for calls that return frames)
    for tag, B, T in (("b1", 1, 160000), ("b4", 4, 160000), ("odd", 2, 123457), ("short", 3, 16000)):
        r = enc.forward(torch.from_numpy(synth.noise_clips(B, T, seed=0)).cuda(), hook_layers=range(13), want_pooled=True)
        p = r["pooled"].cpu().numpy()
        out[f"{dt}.{tag}.pooled_rel_l2_max"] = max(rel_l2(p[b], g[f"{tag}.pooled"][b]) for b in range(B))
        out[f"{dt}.{tag}.frame_rel_l2"] = rel_l2(r["features"].cpu().numpy()[:, ::16], g[f"{tag}.feat_tok16"])
        am = np.concatenate([r["hooks"][i].cpu().numpy().mean(1) for i in range(13)], 1)
        out[f"{dt}.{tag}.all_hooks_mean_rel_l2"] = rel_l2(am, g[f"{tag}.all_mean"])
    r = enc.forward(torch.from_numpy(synth.tone_clips(16000)).cuda(), want_pooled=True)
    out[f"{dt}.tone.pooled_rel_l2"] = rel_l2(r["pooled"].cpu().numpy(), g["tone.pooled"])
    enc.close()
print(json.dumps(out, indent=1))
