#!/usr/bin/env python3
"""dwconv_lds_kernel against dwconv_kernel on one shape, through the library's internal entry points (debug)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from avex_amd import _capi
lib = _capi.lib()
f_old = getattr(lib, "_ZN3avx12dwconv_partsEPKviiiiiiPKfS3_PvPfmPliP12ihipStream_t")
f_new = getattr(lib, "_ZN3avx16dwconv_lds_partsEPKviiiiiiPKfS3_PvPfmPliP12ihipStream_t")
for f in (f_old, f_new):
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_long), C.c_int, C.c_void_p]
def run(B, H, W, Cp, k, st, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.randn(B, H, W, Cp, device="cuda", generator=g).half()
    w = (0.3 * torch.randn(k * k, Cp, device="cuda", generator=g)).float()
    b = (0.1 * torch.randn(Cp, device="cuda", generator=g)).float()
    pad = (k - 1) // 2
    Ho, Wo = (H + 2 * pad - k) // st + 1, (W + 2 * pad - k) // st + 1
    outs = []
    for f in (f_old, f_new):
        out = torch.zeros(B, Ho, Wo, Cp, device="cuda", dtype=torch.half)
        part = torch.zeros(64 << 20, device="cuda", dtype=torch.float32)
        rows = C.c_long(0)
        rc = f(x.data_ptr(), B, H, W, Cp, k, st, w.data_ptr(), b.data_ptr(), out.data_ptr(), part.data_ptr(), part.numel() * 4, C.byref(rows), 0, None)
        torch.cuda.synchronize()
        assert rc == 0, _capi.last_error()
        pool = part[: B * rows.value * Cp].view(B, rows.value, Cp).sum(1)
        outs.append((out.float().cpu().numpy(), pool.cpu().numpy()))
    d = np.abs(outs[0][0] - outs[1][0])
    bad = np.argwhere(d > 0)
    print(f"B={B} {H}x{W}x{Cp} k{k} s{st}: out {Ho}x{Wo}; elements that differ {len(bad)} of {d.size}; max |diff| {d.max():.3e}; pool max diff {np.abs(outs[0][1]-outs[1][1]).max():.3e}")
    if len(bad):
        ys = sorted(set(bad[:, 1].tolist())); xs = sorted(set(bad[:, 2].tolist()))
        print("   rows", ys[:20], "cols", xs[:40])
for shape in [(1, 27, 63, 32, 5, 2), (1, 25, 69, 32, 5, 2), (1, 27, 63, 32, 3, 2), (1, 27, 63, 32, 5, 1), (2, 64, 101, 32, 5, 2), (1, 28, 64, 32, 5, 2), (1, 27, 64, 32, 5, 2), (1, 28, 63, 32, 5, 2)]:
    run(*shape)
