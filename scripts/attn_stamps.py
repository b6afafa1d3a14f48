#!/usr/bin/env python3
"""Per-wave phase stamps of the persistent attention kernel (diagnostic build: AVEX_AMD_DIAG=1 AVEX_AMD_EXTRA_CFLAGS=-DATT_STAMPS=1
python -m avex_amd.build; run with AVEX_AMD_LIB=avex_amd/lib/libavexhip_diag.so).  Prints, per wave of a few workgroups, the
shader cycles each segment of a phase took, the shader clock (s_memtime against the 100 MHz s_memrealtime), and knock-out timings."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from avex_amd import kernels as K, _capi
B, T, H = 256, 496, 12
qkv = torch.randn(B * T, 3 * H * 64, device="cuda").half()
tab = torch.randn(H, 2 * T - 1, device="cuda") * 0.3
gw = torch.randn(8, 64, device="cuda") * 0.1; gb = torch.randn(8, device="cuda") * 0.1; ga = torch.ones(H, device="cuda")
for _ in range(3): K.attention(qkv, B, T, H, tab, gw, gb, ga)
torch.cuda.synchronize()
L = _capi.lib()
buf = np.zeros(64 * 8 * 32 * 8, dtype=np.uint64)
assert L.avexhip_debug_att_stamps(buf.ctypes.data, buf.size) == 0
s = buf.reshape(64, 8, 32, 8).astype(np.int64)
names = ["vmwait", "barrier", "-", "setup", "tiles", "finish"]
ph0, ph1 = 2, 22
d = np.diff(s[:, :, :, :7], axis=3)                    # [blk, wave, ph, 6]
tot = s[:, :, ph1, 0] - s[:, :, ph0, 0]
rt = s[:, :, ph1, 7] - s[:, :, ph0, 7]
print(f"shader clock over phases {ph0}..{ph1}: {np.median(tot / np.maximum(rt, 1)) * 100:.0f} MHz; cycles per item (2 phases): {np.median(tot) / ((ph1 - ph0) / 2):.0f}")
for half in (0, 1):
    sel = d[:, :, ph0 + half:ph1:2, :]                 # even phases = half 0
    med = np.median(sel.reshape(-1, 8 if False else sel.shape[1], sel.shape[2], 6), axis=(0, 2))   # per wave
    print(f"half {half}: median cycles per wave  " + "  ".join(f"{n}" for n in names))
    for w in range(8):
        print(f"   wave {w}: " + "  ".join(f"{int(med[w, i]):6d}" for i in range(6)))

# one key tile's stage stamps (interleaved tile body: phase 6, key tile 3), slot 31
t = s[:, :, 31, :7]
if t[:, :, 0].min() > 0:
    dt = np.diff(t, axis=2)
    print("one key tile (phase 6, tile 3), median cycles per stage:  S(q0) | S(q1)+exp(q0) | check, V wait, next K/bias loads | PV(q0)+exp(q1) | check | PV(q1)+next start | total")
    for w in range(8):
        print(f"   wave {w}: " + "  ".join(f"{int(np.median(dt[:, w, i])):6d}" for i in range(6)) + f"  {int(np.median(t[:, w, 6] - t[:, w, 0])):6d}")
