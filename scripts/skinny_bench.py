#!/usr/bin/env python3
"""EfficientNet's long thin 1 x 1 convolutions as stand-alone products: the 128-tile kernel (variant 3) against the skinny streaming kernel
(variant 7), HBM GB/s on the algorithmic bytes (A read + output written)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from avex_amd import kernels as K

def run(M, Kd, N, variant, silu):
    a = (torch.randn(M, Kd, device="cuda") * 0.5).half()
    w = (torch.randn(N, Kd, device="cuda") * 0.1).half()
    b = torch.randn(N, device="cuda") * 0.1
    f = lambda: K.gemm(a, w, bias=b, silu=silu, out_f32=False, out_half=True, variant=variant)
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 10
    for _ in range(n): f()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / n * 1e3
    gb = (M * Kd * 2 + M * N * 2) / 1e9
    return ms, gb / ms

for M, Kd, N, silu in ((8208384, 64, 128, True), (8208384, 64, 128, False), (2056192, 64, 256, True), (2056192, 128, 128, False), (2056192, 256, 128, False)):
    r = []
    for v in (3, 7):
        if v == 3 and N % 128: r.append((float("nan"), float("nan"))); continue
        r.append(run(M, Kd, N, v, silu))
    print(f"M={M} K={Kd} N={N} silu={silu}: 128-tile {r[0][0]:.3f} ms ({r[0][1]:.2f} TB/s)   skinny {r[1][0]:.3f} ms ({r[1][1]:.2f} TB/s)")
