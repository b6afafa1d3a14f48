#!/usr/bin/env python3
"""Call time of the BEATs forward at small batches against the row count from which the LayerNorm fold (256-tile streaming GEMM) is used
(below it: 128-tile GEMM, split-K for fc2, LayerNorm kernels)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from avex_amd import synth, kernels as K
cfg = synth.BEATS_BASE_CFG
sd = synth.beats_state_dict(cfg, seed=0)
print("batch: " + "  ".join(f"{b:7d}" for b in (1, 2, 4, 8, 16, 32, 64)))
for thr in (sys.argv[1:] or ("1", "auto:1024", "auto:2048", "auto:4096", "auto:8192", "auto:16384", "auto:32768")):
    os.environ["AVEX_AMD_LN_FOLD"] = thr
    enc = K.BeatsEncoder(cfg, sd, operand_dtype="f16", residual="half")
    row = []
    for B in (1, 2, 4, 8, 16, 32, 64):
        wav = (0.1 * torch.randn(B, 160000)).cuda()
        for _ in range(5): enc.forward(wav, want_features=False, want_pooled=True)
        torch.cuda.synchronize(); t0 = time.perf_counter(); n = 30
        for _ in range(n): enc.forward(wav, want_features=False, want_pooled=True)
        torch.cuda.synchronize(); row.append((time.perf_counter() - t0) / n * 1e3)
    print(f"{thr:11s} min_tiles {os.environ.get('AVEX_AMD_GEMM_256_MIN_TILES', '0'):>4s}: " + "  ".join(f"{t:7.3f}" for t in row) + "  ms per call")
    enc.close()
