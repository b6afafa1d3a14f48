#!/usr/bin/env python3
"""Variable-length serving: N distinct clip lengths back to back through one BEATs handle.  Run under `rocprofv3 --hip-trace --stats` with
N = 1 and N = 40: the hipMalloc / hipFree / hipMemcpy counts of the two runs must be EQUAL (a new length costs one small kernel on the
forward's stream -- the bias table -- and nothing else; round 3 paid a blocking hipMalloc + hipMemcpy per new length and a device-wide
hipFree past 16 lengths).  Inputs, outputs and the workspace are allocated up front for the longest clip.

    python scripts/varlen_alloc.py 40
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from avex_amd import kernels as K, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
lengths = [8000 + 7920 * i for i in range(40)]
enc = K.BeatsEncoder(synth.BEATS_BASE_CFG, synth.beats_state_dict(synth.BEATS_BASE_CFG, seed=0))
big = torch.from_numpy(synth.noise_clips(2, lengths[-1], seed=1)).cuda()
enc.forward(big, want_features=False, want_pooled=True)          # longest first: workspace and caching-allocator blocks exist from here on
torch.cuda.synchronize()
print("MARK begin variable-length phase", flush=True)
outs = []
for i in range(40):
    L = lengths[i % n]                                          # n = 1: forty forwards of ONE length; n = 40: forty different lengths
    outs.append(enc.forward(big[:, :L], want_features=False, want_pooled=True)["pooled"][0, :4].clone())
torch.cuda.synchronize()
print("MARK end; tokens of the last clip:", enc.num_tokens(lengths[(40 - 1) % n]), "checksum", float(sum(o.sum() for o in outs)))
