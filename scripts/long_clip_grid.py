#!/usr/bin/env python3
"""Few long clips: the attention's unit of work is (clip, head, query block of 512), dealt to the workgroups in consecutive runs.
AVEX_AMD_ATT_GRID = clips x heads reproduces the previous mapping (one workgroup per (clip, head) walking all its query blocks),
so both are timed in one process on the same handle.  Prints the attention stage (serialised HIP events) and the whole forward."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from avex_amd import synth, kernels as K


def timeit(fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


cfg = synth.BEATS_BASE_CFG
enc = K.BeatsEncoder(cfg, synth.beats_state_dict(cfg, seed=0), operand_dtype="f16", residual="half")
print("clips x seconds  tokens  units | per (clip, head): forward ms, attention ms | per (clip, head, query block): forward ms, attention ms")
for B, secs in ((1, 60), (1, 120), (2, 60), (4, 30), (8, 20), (32, 60)):
    wav = (0.1 * torch.randn(B, 16000 * secs)).cuda()
    Tn = enc.num_tokens(16000 * secs)
    rem = Tn % 512
    nqb = Tn // 512 if 0 < rem <= 32 else (Tn + 511) // 512
    row = []
    outs = []
    for grid in (B * 12, 0):
        if grid:
            os.environ["AVEX_AMD_ATT_GRID"] = str(grid)
        else:
            os.environ.pop("AVEX_AMD_ATT_GRID", None)
        dt = timeit(lambda: enc.forward(wav, want_features=False, want_pooled=True), 10)
        enc.set_profiling(True)
        enc.forward(wav, want_features=False, want_pooled=True)
        torch.cuda.synchronize()
        att = sum(ms for name, ms, _ in enc.last_profile() if name == "attention")
        enc.set_profiling(False)
        row.append((1e3 * dt, att))
        outs.append(enc.forward(wav, want_features=False, want_pooled=True)["pooled"].clone())
    same = torch.equal(outs[0], outs[1])
    print(f"{B:3d} x {secs:3d} s  {Tn:6d}  {B * 12 * nqb:5d} | {row[0][0]:8.2f} {row[0][1]:8.3f} | {row[1][0]:8.2f} {row[1][1]:8.3f} | identical output: {same}")
