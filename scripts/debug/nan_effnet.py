#!/usr/bin/env python3
"""EfficientNet: does a NaN sample reach the pooled output (and only its clip's)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from avex_amd import kernels as K, synth
from avex_amd.effnet_encoder import EfficientNetB0Encoder
enc = EfficientNetB0Encoder(synth.effnet_b0_state_dict())
plan = K.MelspecPlan(n_fft=800, hop_length=160, n_mels=128, normalize=True)
x = synth.noise_clips(4, 160000, seed=5)
clean = enc.forward(plan(torch.from_numpy(x).cuda()), want_features=False, want_pooled=True)["pooled"].cpu().numpy()
bad = x.copy(); bad[2, 80000] = np.nan
p = enc.forward(plan(torch.from_numpy(bad).cuda()), want_features=False, want_pooled=True)["pooled"].cpu().numpy()
print("nan per clip", [int(np.isnan(p[i]).sum()) for i in range(4)], "other clips identical", [bool(np.array_equal(p[i], clean[i])) for i in (0, 1, 3)])
