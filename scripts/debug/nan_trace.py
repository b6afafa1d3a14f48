#!/usr/bin/env python3
"""Where does a NaN sample stop being NaN?  (tests/test_gpu_e2e.py::test_nan_input_stays_nan_and_stays_in_its_clip)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from avex_amd import kernels as K, synth
cfg = synth.BEATS_BASE_CFG; sd = synth.beats_state_dict(cfg, seed=0)
x = synth.noise_clips(4, 32000, seed=31); x[2, 17000] = np.nan
for res in ("f32", "half"):
    enc = K.BeatsEncoder(cfg, sd, operand_dtype="f16", residual=res, on_overflow="ignore")
    r = enc.forward(torch.from_numpy(x).cuda(), hook_layers=list(range(13)), want_features=True, want_pooled=True)
    print(res, "pooled nan per clip", [int(torch.isnan(r["pooled"][i]).sum()) for i in range(4)], "features nan", [int(torch.isnan(r["features"][i]).sum()) for i in range(4)])
    for i in range(13):
        print("  hook", i, [int(torch.isnan(r["hooks"][i][c]).sum()) for c in range(4)])
    enc.close()
plan = K.FbankPlan() if hasattr(K, "FbankPlan") else None
if plan is not None:
    fb = plan(torch.from_numpy(x).cuda())
    print("fbank out", tuple(fb.shape), "nan per clip", [int(torch.isnan(fb[i].float()).sum()) for i in range(4)])
