#!/usr/bin/env python3
"""Where a small batch's time goes: per-stage HIP-event times of one forward at batch 1 / 4 / 16, LayerNorm fold on (every layer GEMM
in the 256-tile streaming kernel) and off (128-tile kernel below 1024 rows), and the call time of both."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from avex_amd import synth, kernels as K

cfg = synth.BEATS_BASE_CFG
sd = synth.beats_state_dict(cfg, seed=0)
for fold in ("1", "0"):
    os.environ["AVEX_AMD_LN_FOLD"] = fold
    enc = K.BeatsEncoder(cfg, sd, operand_dtype="f16", residual="half")
    for B in (1, 4, 16):
        wav = (0.1 * torch.randn(B, 160000)).cuda()
        for _ in range(5):
            enc.forward(wav, want_features=False, want_pooled=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            enc.forward(wav, want_features=False, want_pooled=True)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 50 * 1e3
        enc.set_profiling(True)
        enc.forward(wav, want_features=False, want_pooled=True)
        pr = enc.last_profile()
        enc.set_profiling(False)
        print(f"fold {fold} batch {B:2d}: {ms:.3f} ms per call | " + "  ".join(f"{n} {t:.3f}" for n, t, _ in pr))
    enc.close()
