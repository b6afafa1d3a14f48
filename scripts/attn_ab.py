#!/usr/bin/env python3
"""A/B of the attention variants in ONE process (rule: perf deltas come from interleaved rounds on one device).
   python scripts/attn_ab.py [B] [rounds] [variants, e.g. 2,3]
Alternates AVEX_AMD_ATT_VARIANT launch group by launch group on random data at the power cap; prints per-variant median / min
microseconds per launch, TFLOP/s and the agreement of the outputs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from avex_amd import kernels as K

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 8
variants = sys.argv[3].split(",") if len(sys.argv) > 3 else ["2", "3"]
T = int(os.environ.get("ATT_T", "496")); H = 12
NOBIAS = os.environ.get("ATT_NOBIAS", "0") == "1"      # the encoders without relative position bias (EAT, AVES): plain softmax(q k^T / 8) v
per = 12      # launches per group (one step's worth)
torch.manual_seed(0)
qkv = torch.randn(B * T, 3 * H * 64, device="cuda").half()
tab = torch.randn(H, 2 * T - 1, device="cuda") * 0.3
gw = torch.randn(8, 64, device="cuda") * 0.1; gb = torch.randn(8, device="cuda") * 0.1; ga = torch.ones(H, device="cuda")
if NOBIAS:
    tab = gw = gb = ga = None
outs = {}
for v in variants:
    os.environ["AVEX_AMD_ATT_VARIANT"] = v
    outs[v] = K.attention(qkv, B, T, H, tab, gw, gb, ga).float()
torch.cuda.synchronize()
ref = outs[variants[0]]
for v in variants[1:]:
    d = (outs[v] - ref).norm() / ref.norm()
    print(f"variant {v} vs {variants[0]}: rel-L2 {d.item():.3e}, max abs {(outs[v]-ref).abs().max().item():.3e}")
# warm the board to its cap
os.environ["AVEX_AMD_ATT_VARIANT"] = variants[0]
for _ in range(200): K.attention(qkv, B, T, H, tab, gw, gb, ga)
torch.cuda.synchronize()
times = {v: [] for v in variants}
for r in range(rounds):
    for v in variants:
        os.environ["AVEX_AMD_ATT_VARIANT"] = v
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(per): K.attention(qkv, B, T, H, tab, gw, gb, ga)
        e1.record(); torch.cuda.synchronize()
        times[v].append(e0.elapsed_time(e1) / per * 1e3)
fl = 4.0 * B * T * T * H * 64
for v in variants:
    t = np.array(times[v])
    print(f"variant {v}: median {np.median(t):.1f} us  min {t.min():.1f} us  ({fl/np.median(t)/1e6:.0f} TFLOP/s)  all: {' '.join(f'{x:.0f}' for x in t)}")
