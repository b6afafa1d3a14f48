#!/usr/bin/env python3
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from avex_amd import kernels as K
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
T, H = 496, 12
qkv = torch.randn(B * T, 3 * H * 64, device="cuda").half()
tab = torch.randn(H, 2 * T - 1, device="cuda") * 0.3
gw = torch.randn(8, 64, device="cuda") * 0.1; gb = torch.randn(8, device="cuda") * 0.1; ga = torch.ones(H, device="cuda")
for _ in range(2): K.attention(qkv, B, T, H, tab, gw, gb, ga)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters): K.attention(qkv, B, T, H, tab, gw, gb, ga)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
print(f"attention B={B}: {ms*1e3:.1f} us  {4.0*B*T*T*H*64/ms/1e9:.1f} TFLOP/s  per block-round {ms*1e3/(B*H/256):.2f} us")
