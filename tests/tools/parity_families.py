#!/usr/bin/env python3
"""Measured parity of every model family against its CPU oracle, as one JSON document (run on the GPU box; the copy under
profiles/ is what DESIGN.md and the test tolerances quote).

    python tests/tools/parity_families.py > gpurun_out/r03_parity.json

BEATs is pinned to the real reference (tests/golden/); EAT / EfficientNet / AVES oracles are restatements of third-party code that
neither machine has ("parity unpinned"), so those rows measure HIP-vs-restatement on synthetic weights.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from avex_amd import kernels as K
from avex_amd import synth


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(f"{np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30):.3e}")


def beats():
    g = np.load(os.path.join(ROOT, "tests", "golden", "base_api.npz"))
    sd = synth.beats_state_dict(synth.BEATS_BASE_CFG, seed=0)
    out = {"reference": "tests/golden/base_api.npz (the real avex BEATs, fp32 CPU)"}
    for dt in ("f16", "bf16"):
        for res in ("half", "f32"):
            enc = K.BeatsEncoder(synth.BEATS_BASE_CFG, sd, operand_dtype=dt, residual=res)
            r = enc.forward(torch.from_numpy(synth.noise_clips(4, 160000, seed=0)).cuda(), hook_layers=range(13), want_pooled=True)
            am = np.concatenate([r["hooks"][i].cpu().numpy().mean(1) for i in range(13)], 1)
            out[f"{dt}.residual_{res}"] = {"pooled": rel(r["pooled"].cpu().numpy(), g["b4.pooled"]),
                                           "frame_level": rel(r["features"].cpu().numpy()[:, ::16], g["b4.feat_tok16"]),
                                           "all_13_hooks_mean": rel(am, g["b4.all_mean"]), "overflow_events": enc.overflow_events()}
            enc.close()
    return out


def eat():
    from avex_amd.eat_encoder import EatEncoder
    from oracle import eat_oracle as EO
    cfg = synth.EAT_BASE_CFG
    sd = synth.eat_state_dict(cfg)
    wav = synth.noise_clips(2, 80000, seed=12)
    ref, taps = EO.eat_forward(wav, sd, cfg)
    out = {"reference": "oracle/eat_oracle.py (parity unpinned), EAT-base, 2 clips x 5 s, 513 tokens"}
    out["policy"] = ("residual='auto' (the default): calls that return un-averaged rows -- features, the class token, un-pooled taps -- run on the fp32 "
                     "residual stream, token-mean-only calls (config C3's timed path: pooled_mean_only) on the operand-type stream")
    for dt in ("f16", "bf16"):
        for res in ("auto", "half", "f32"):
            enc = EatEncoder(cfg, sd, operand_dtype=dt, residual=res)
            r = enc.forward(torch.from_numpy(wav).cuda(), hook_layers=list(range(12)), pooling="mean")
            f = r["features"].cpu().numpy()
            pm = enc.forward(torch.from_numpy(wav).cuda(), want_features=False, pooling="mean")["pooled"].cpu().numpy()
            out[f"{dt}.residual_{res}"] = {"pooled_mean": rel(f.mean(1), ref.mean(1)), "pooled_mean_only": rel(pm, ref.mean(1)), "cls_token": rel(f[:, 0], ref[:, 0]), "frame_level": rel(f, ref),
                       "taps_mean_worst": max(rel(r["hooks"][i].cpu().numpy().mean(1), taps[f"backbone.model.blocks.{i}.attn.proj"].mean(1)) for i in range(12))}
            enc.close()
    return out


def effnet():
    from avex_amd.effnet_encoder import EfficientNetB0Encoder
    from oracle import effnet_oracle as EO
    out = {"reference": "oracle/effnet_oracle.py (parity unpinned), mel image 2 x 64 x 101"}
    for variant, stages, sdf in (("b0", synth.EFFNET_B0_STAGES, synth.effnet_b0_state_dict),):
        sd = sdf()
        mel = np.abs(synth.normal("emel", (2, 64, 101), 0.5)).astype(np.float32)
        ref, taps = EO.effnet_features(mel, sd, stages)
        for dt in ("f16", "bf16"):
            enc = EfficientNetB0Encoder(sd, operand_dtype=dt)
            names = enc.tap_names()
            r = enc.forward(torch.from_numpy(mel).cuda(), hook_layers=names, want_features=True, want_pooled=True)
            out[f"{variant}.{dt}"] = {"features": rel(r["features"].cpu().numpy(), ref), "pooled": rel(r["pooled"].cpu().numpy(), ref.mean((2, 3))),
                                      "taps_worst": max(rel(r["hooks"][n].cpu().numpy(), taps[n]) for n in names)}
    return out


def aves():
    from avex_amd.aves_encoder import AvesEncoder
    from oracle import aves_oracle as AO
    cfg = synth.AVES_BASE_CFG
    sd = synth.aves_state_dict(cfg)
    x = synth.noise_clips(2, 32000, seed=44)
    ref, taps = AO.aves_forward(x, sd, cfg)
    name = "model.encoder.transformer.layers.{}.feed_forward.output_dense"
    out = {"reference": "oracle/aves_oracle.py (parity unpinned), wav2vec2-base, 2 clips x 2 s, 12 layers",
           "policy": "residual='auto' (the default): fp32 residual stream for calls that return frames or un-pooled taps, operand type for token means only (pooled_only)"}
    for res in ("auto", "half", "f32"):
        enc = AvesEncoder(cfg, sd, residual=res)
        r = enc.forward(torch.from_numpy(x).cuda(), hook_layers=list(range(12)), want_features=True, want_pooled=True)
        po = enc.forward(torch.from_numpy(x).cuda(), want_features=False, want_pooled=True)["pooled"].cpu().numpy()
        out[f"f16.residual_{res}"] = {"pooled": rel(r["pooled"].cpu().numpy(), ref.mean(1)), "pooled_only": rel(po, ref.mean(1)), "frame_level": rel(r["features"].cpu().numpy(), ref),
                                      "taps_mean_worst": max(rel(r["hooks"][i].cpu().numpy().mean(1), taps[name.format(i)].mean(1)) for i in range(12))}
        enc.close()
    return out


def main():
    doc = {"metric": "relative L2 error, || hip - reference || / || reference ||", "beats": beats(), "eat": eat(), "efficientnet": effnet(), "aves": aves()}
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
