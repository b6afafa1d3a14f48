#!/usr/bin/env python3
"""Relative L2 error of every EfficientNet-B0 hook tap and of the final features against the NumPy restatement (synthetic weights)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from avex_amd import synth
from avex_amd.effnet_encoder import EfficientNetB0Encoder
from oracle import effnet_oracle as EO
rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
sd = synth.effnet_b0_state_dict()
for dt in ("f16", "bf16"):
    enc = EfficientNetB0Encoder(sd, operand_dtype=dt)
    mel = np.abs(synth.normal("emel", (2, 64, 101), 0.5)).astype(np.float32)
    ref, taps = EO.effnet_features(mel, sd, synth.EFFNET_B0_STAGES)
    names = enc.tap_names()
    r = enc.forward(torch.from_numpy(mel).cuda(), hook_layers=names, want_features=True, want_pooled=True)
    print(dt, "features", f"{rel(r['features'].cpu().numpy(), ref):.2e}", "pooled", f"{rel(r['pooled'].cpu().numpy(), ref.mean((2, 3))):.2e}")
    print("   taps:", " ".join(f"{rel(r['hooks'][n].cpu().numpy(), taps[n]):.1e}" for n in names))
