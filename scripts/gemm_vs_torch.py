#!/usr/bin/env python3
"""Reference point only (nothing in the product calls it): the same GEMM shapes through torch.matmul (rocBLAS / hipBLASLt)
and through avexhip_gemm, bias + f16 output, on random data, same process and device."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from avex_amd import kernels as K
M = 256 * 496
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for name, N, Kd in (("qkv", 2304, 768), ("out", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)):
    x = torch.randn(M, Kd, device="cuda").half(); w = (torch.randn(N, Kd, device="cuda") * 0.05).half(); b = torch.randn(N, device="cuda")
    bh = b.half()
    ms_t = t(lambda: torch.addmm(bh, x, w.t()))
    ms_l = t(lambda: torch.nn.functional.linear(x, w, bh))
    ms_a = t(lambda: K.gemm(x, w, bias=b, out_f32=False, out_half=True))
    fl = 2.0 * M * N * Kd / 1e9
    print(f"{name}: N={N} K={Kd}  torch.addmm {ms_t*1e3:7.1f} us {fl/ms_t:7.1f} TF | F.linear {ms_l*1e3:7.1f} us {fl/ms_l:7.1f} TF | avexhip_gemm {ms_a*1e3:7.1f} us {fl/ms_a:7.1f} TF", flush=True)
