#!/usr/bin/env python3
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from avex_amd import kernels as K, synth
x = torch.randn(300, 512, device="cuda"); x[7, 33] = float("nan")
w = torch.ones(512, device="cuda"); b = torch.zeros(512, device="cuda")
for half_in in (False, True):
    o32, oh = K.layernorm(x.half() if half_in else x, w, b)
    print("layernorm half_in", half_in, "f32 nan rows", int(torch.isnan(o32).any(1).sum()), "half nan rows", int(torch.isnan(oh.float()).any(1).sum()))
a = torch.randn(512, 256, device="cuda").half(); a[5, 9] = float("nan")
wt = torch.randn(512, 256, device="cuda").half() * 0.05
bias = torch.zeros(512, device="cuda")
for M in (512,):
    r = K.gemm(a, wt, bias=bias, out_f32=True, out_half=True)
    print("gemm out_f32 nan rows", int(torch.isnan(r["f32"]).any(1).sum()), "half nan rows", int(torch.isnan(r["half"].float()).any(1).sum()))
    r = K.gemm(a, wt, bias=bias, out_f32=False, out_half=True)
    print("gemm half-only nan rows", int(torch.isnan(r["half"].float()).any(1).sum()))
h = torch.tensor([float("nan"), 1.0, 7e4, -7e4, float("inf")], device="cuda")
print("cast_to_half", K.cast_to_half(h).float().tolist() if hasattr(K, "cast_to_half") else "n/a")
