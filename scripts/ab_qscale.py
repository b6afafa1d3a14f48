#!/usr/bin/env python3
"""log2(e) of the attention's base-2 softmax folded into W_q / b_q in fp32 (default) against plain Q scaled inside the kernel by a
constant rounded to the operand type (AVEX_AMD_Q_LOG2E=0): parity against the reference golden for both operand types, and step time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from avex_amd import synth, kernels as K


def rel(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b.astype(np.float64)))


cfg = synth.BEATS_BASE_CFG
sd = synth.beats_state_dict(cfg, seed=0)
g = np.load(os.path.join(ROOT, "tests", "golden", "base_api.npz"))
x4 = torch.from_numpy(synth.noise_clips(4, 160000, seed=0)).cuda()
wav = (0.1 * torch.randn(256, 160000)).cuda()
wav[:4] = x4
for dt in ("f16", "bf16"):
    for res in ("half", "f32"):
        for fold in ("0", "1"):
            os.environ["AVEX_AMD_Q_LOG2E"] = fold
            enc = K.BeatsEncoder(cfg, sd, operand_dtype=dt, residual=res)
            r = enc.forward(x4, want_features=True, want_pooled=True)
            pooled = rel(r["pooled"].cpu().numpy(), g["b4.pooled"])
            frame = rel(r["features"].cpu().numpy()[:, ::16], g["b4.feat_tok16"])
            for _ in range(3):
                enc.forward(wav, want_features=False, want_pooled=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                enc.forward(wav, want_features=False, want_pooled=True)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 10 * 1e3
            print(f"{dt:5s} residual {res:4s} q_log2e {fold}: pooled {pooled:.3e}  frame level {frame:.3e}  step {ms:.2f} ms")
            enc.close()
os.environ.pop("AVEX_AMD_Q_LOG2E", None)
