#!/usr/bin/env python3
"""Micro-benchmark of the MFMA GEMM through the C ABI on the BEATs shapes (random operands).
    python scripts/gemm_bench.py [--variant V] [--iters N] [--clips B] [--shapes qkv,fc1,fc2,out]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from avex_amd import kernels as K

ap = argparse.ArgumentParser()
ap.add_argument("--variant", type=int, default=0)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--clips", type=int, default=256)
ap.add_argument("--shapes", default="qkv,out,fc1,fc2")
ap.add_argument("--dtype", default="f16")
a = ap.parse_args()
M = a.clips * 496
SH = {"qkv": (2304, 768, dict(out_f32=False, out_half=True)), "out": (768, 768, dict(out_f32=False, out_half=True, resid=True)),
      "fc1": (3072, 768, dict(out_f32=False, out_half=True, gelu=True)), "fc2": (768, 3072, dict(out_f32=False, out_half=True, resid=True)),
      "sq": (4096, 4096, dict(out_f32=False, out_half=True))}
td = torch.float16 if a.dtype == "f16" else torch.bfloat16
for name in a.shapes.split(","):
    N, Kd, opt = SH[name]
    Mm = 4096 if name == "sq" else M
    x = torch.randn(Mm, Kd, device="cuda").to(td)
    w = (torch.randn(N, Kd, device="cuda") * 0.05).to(td)
    bias = torch.randn(N, device="cuda")
    kw = dict(bias=bias, variant=a.variant, out_f32=False, out_half=True, gelu=opt.get("gelu", False))
    if opt.get("resid"):
        kw["resid_half"] = torch.randn(Mm, N, device="cuda").to(td)
        kw["alpha"] = 2.2
    for _ in range(3):
        K.gemm(x, w, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        K.gemm(x, w, **kw)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    print(f"{name:4s} M={Mm} N={N} K={Kd} variant={a.variant}: {ms*1e3:8.1f} us  {2.0*Mm*N*Kd/ms/1e9:7.1f} TFLOP/s", flush=True)
