import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from avex_amd import kernels as K
B, T, E, H = 2, 24, 128, 4
hd = E // H
g = torch.Generator().manual_seed(0)
qkv = torch.randn(B, T, 3 * E, generator=g)
pad = torch.zeros(B, T, dtype=torch.bool); pad[-1, T // 2:] = True
def ref(kp):
    q, k, v = (qkv[..., i * E:(i + 1) * E].reshape(B, T, H, hd).permute(0, 2, 1, 3).double() for i in range(3))
    s = q @ k.transpose(-1, -2) / hd ** 0.5
    if kp is not None:
        s = s.masked_fill(kp[:, None, None, :], float("-inf"))
    return (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(B, T, E)
for kp in (None, pad):
    out = K.mha_f32(qkv.cuda(), H, None if kp is None else kp.cuda()).cpu().double()
    err = (out - ref(kp)).abs()
    print("kp" if kp is not None else "none", "max err", float(err.max()))
    print(" per clip/query max err:\n", np.array2string(err.amax(-1).numpy(), precision=1, max_line_width=250))
    print(" per head:", err.reshape(B, T, H, hd).amax((1, 3)).numpy())
