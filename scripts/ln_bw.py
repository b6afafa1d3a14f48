#!/usr/bin/env python3
"""LayerNorm kernel (half in -> half out, as in the default residual stream) against the row count: bytes moved per second when the
tensor fits the 256 MB Infinity Cache and when it does not (is the 70 us per launch in the step an HBM figure or a kernel figure?)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from avex_amd import kernels as K
C = 768
g = torch.ones(C, device="cuda"); b = torch.zeros(C, device="cuda")
for M in (8192, 16384, 31744, 63488, 126976, 253952):
    x = torch.randn(M, C, device="cuda").half()
    for _ in range(3): K.layernorm(x, g, b, want_f32=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n): K.layernorm(x, g, b, want_f32=False)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    mb = 2 * M * C * 2 / 1e6
    print(f"rows {M:7d}: {mb:7.1f} MB in+out  {us:7.1f} us  {mb / us :.2f} TB/s")
