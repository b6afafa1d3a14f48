#!/usr/bin/env python3
"""Fixed per-tile cost of the GEMM: time vs K at fixed M, N (rounds = tiles / 256 CUs)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from avex_amd import kernels as K
M = 256 * 496
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for N in (768, 2304):
    for Kd in (64, 128, 256, 512, 768, 1536, 3072):
        x = torch.randn(M, Kd, device="cuda").half(); w = (torch.randn(N, Kd, device="cuda") * 0.05).half()
        bias = torch.randn(N, device="cuda")
        for mode in ("half", "gelu"):
            kw = dict(bias=bias, variant=variant, out_f32=False, out_half=True, gelu=(mode == "gelu"))
            for _ in range(2): K.gemm(x, w, **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): K.gemm(x, w, **kw)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            tiles = (M // 256) * (N // 256)
            print(f"N={N} K={Kd:5d} {mode:5s}: {ms*1e3:8.1f} us  rounds={tiles/256:5.2f}  us/tile-round={ms*1e3/(tiles/256):6.2f}  {2.0*M*N*Kd/ms/1e9:7.1f} TF", flush=True)
