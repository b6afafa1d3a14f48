#!/usr/bin/env python3
"""How much does a clip's pooled embedding depend on the batch it arrives in?  Per family: the same clips alone (a batch of 4) and at rows of
a large batch, default mode and batch-invariant mode (kernels.residual_code) -> JSON (profiles/r04_parity.json "batch_dependence").

    python scripts/batch_invariance.py
"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from _util import rel_l2
from avex_amd import synth, kernels as K
from avex_amd.eat_encoder import EatEncoder
from avex_amd.aves_encoder import AvesEncoder
from avex_amd.effnet_encoder import EfficientNetB0Encoder

out = {}
rows = (0, 1, 100, 255)


def report(name, small, big, t_default=None):
    a, b = small.float().cpu().numpy(), big.float().cpu().numpy()
    return {"rel_l2": float(f"{rel_l2(a, b):.3e}"), "bit_identical": bool(torch.equal(small, big))}


# BEATs-base, 256 x 10 s
sd = synth.beats_state_dict(synth.BEATS_BASE_CFG, seed=0)
x = torch.from_numpy(synth.noise_clips(256, 160000, seed=0)).cuda()
for mode in ("default", "batch_invariant"):
    for residual in ("half", "f32"):
        enc = K.BeatsEncoder(synth.BEATS_BASE_CFG, sd, operand_dtype="f16", residual=residual, batch_invariant=(mode == "batch_invariant"))
        big = enc.forward(x, want_features=False, want_pooled=True)["pooled"][list(rows)]
        small = enc.forward(x[list(rows)], want_features=False, want_pooled=True)["pooled"]
        one = enc.forward(x[100:101], want_features=False, want_pooled=True)["pooled"]
        r = report("beats", small, big)
        r["one_clip_vs_256"] = report("beats", one, big[2:3])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            enc.forward(x, want_features=False, want_pooled=True)
        torch.cuda.synchronize(); r["ms_per_256_clip_step"] = round((time.perf_counter() - t0) / 5 * 1e3, 2)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            enc.forward(x[:1], want_features=False, want_pooled=True)
        torch.cuda.synchronize(); r["ms_per_1_clip_step"] = round((time.perf_counter() - t0) / 20 * 1e3, 3)
        out[f"beats.f16.{residual}.{mode}"] = r
        enc.close()
del x
# EAT-base, 64 x 5 s
sd = synth.eat_state_dict()
x = torch.from_numpy(synth.noise_clips(64, 80000, seed=3)).cuda()
for mode in ("default", "batch_invariant"):
    enc = EatEncoder(synth.EAT_BASE_CFG, sd, operand_dtype="f16", batch_invariant=(mode == "batch_invariant"))
    big = enc.forward(x, want_features=False, pooling="mean")["pooled"][[0, 1, 40, 63]]
    small = enc.forward(x[[0, 1, 40, 63]], want_features=False, pooling="mean")["pooled"]
    out[f"eat.f16.half.{mode}"] = report("eat", small, big)
    enc.close()
del x
# AVES, 32 x 10 s
try:
    cfg = synth.AVES_BASE_CFG
    sd = synth.aves_state_dict(cfg)
    x = torch.from_numpy(synth.noise_clips(32, 160000, seed=4)).cuda()
    for mode in ("default", "batch_invariant"):
        enc = AvesEncoder(cfg, sd, operand_dtype="f16", batch_invariant=(mode == "batch_invariant"))
        big = enc.forward(x)["features"].mean(1)[[0, 1, 20, 31]]
        small = enc.forward(x[[0, 1, 20, 31]])["features"].mean(1)
        out[f"aves.f16.half.{mode}"] = report("aves", small, big)
    del x
except Exception as e:  # noqa: BLE001
    out["aves"] = {"error": repr(e)[:200]}
# EfficientNet-B0, 256 x 10 s (no LayerNorm fold, no split-K: one mode)
enc = EfficientNetB0Encoder(synth.effnet_b0_state_dict())
plan = K.MelspecPlan(n_fft=800, hop_length=160, n_mels=128, normalize=True)
x = torch.from_numpy(synth.noise_clips(256, 160000, seed=2)).cuda()
big = enc.forward(plan(x), want_features=False, want_pooled=True)["pooled"][list(rows)]
small = enc.forward(plan(x[list(rows)]), want_features=False, want_pooled=True)["pooled"]
out["effnet_b0.f16"] = report("effnet", small, big)
print(json.dumps(out, indent=1))
