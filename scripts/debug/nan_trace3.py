#!/usr/bin/env python3
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from avex_amd import kernels as K
for M in (512, 2048):
    a = torch.randn(M, 256, device="cuda").half(); a[5, 9] = float("nan"); a[7, 100] = float("inf")
    wt = torch.randn(512, 256, device="cuda").half() * 0.05
    bias = torch.zeros(512, device="cuda")
    ref = a.float() @ wt.float().T
    print("M", M, "torch: nan rows", torch.isnan(ref).any(1).nonzero().flatten().tolist(), "inf rows", torch.isinf(ref).any(1).nonzero().flatten().tolist())
    for variant in (0, 1, 3, 5):
        try:
            r = K.gemm(a, wt, bias=bias, out_f32=True, out_half=True, variant=variant)
            f = r["f32"]; h = r["half"].float()
            print("  variant", variant, "f32 nan rows", torch.isnan(f).any(1).nonzero().flatten().tolist(), "non-finite rows", (~torch.isfinite(f)).any(1).nonzero().flatten().tolist(),
                  "| half nan rows", torch.isnan(h).any(1).nonzero().flatten().tolist(), "row5 sample", f[5, :3].tolist(), "row7 sample", f[7, :3].tolist())
        except Exception as e:
            print("  variant", variant, "failed:", str(e)[:100])
