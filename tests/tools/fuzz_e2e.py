#!/usr/bin/env python3
"""Differential fuzzing of the BEATs path against the CPU oracle: random batch sizes, clip lengths (0.2 s .. 26 s, i.e. both
attention instantiations and the tail kernel), amplitudes, padding masks, chunk sizes, operand types and residual modes.
    python tests/tools/fuzz_e2e.py [cases] [seed]
Prints one line per case and the worst ratio to the tolerance; exits 1 on the first violation."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from avex_amd import synth, kernels as K
from oracle import beats_oracle as O

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
cfg = synth.BEATS_BASE_CFG
sd = synth.beats_state_dict(cfg, seed=0)
rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
worst = 0.0
encs = {}
for c in range(cases):
    dt = ["f16", "bf16"][int(rng.integers(0, 5) == 0)]
    res = ["half", "f32"][int(rng.integers(0, 3) == 0)]
    chunk = int(rng.choice([1, 2, 3, 7, 256]))
    key = (dt, res, chunk)
    if key not in encs:
        encs[key] = K.BeatsEncoder(cfg, sd, operand_dtype=dt, max_chunk_clips=chunk, residual=res)
    enc = encs[key]
    B = int(rng.integers(1, 5))
    kind = int(rng.integers(0, 4))
    n = int([rng.integers(2800, 20000), rng.integers(20000, 170000), rng.integers(164000, 172000), rng.integers(170000, 420000)][kind])
    if kind == 3:
        B = min(B, 2)
    amp = float(10 ** rng.uniform(-3, 0.5))
    x = (amp * rng.standard_normal((B, n))).astype(np.float32)
    mask = None
    if rng.integers(0, 2):
        mask = np.zeros((B, n), bool)
        for b in range(B):
            if rng.integers(0, 2):
                cut = int(rng.integers(n // 4, n))
                mask[b, cut:] = True
                x[b, cut:] = 0.0
    t0 = time.time()
    f, taps = O.beats_forward(x, sd, cfg, padding_mask=mask)
    tok = f.shape[1]
    frame_pad = None
    if mask is not None:
        frames = 1 + (n - 400) // 160
        fm = O.forward_padding_mask(frames, mask)
        frame_pad = O.forward_padding_mask(tok, fm)
    hook = int(rng.integers(0, 13))
    r = enc.forward(torch.from_numpy(x).cuda(), hook_layers=[hook], want_features=True, want_pooled=False,
                    frame_pad=torch.from_numpy(frame_pad).cuda() if frame_pad is not None else None)
    got = r["features"].cpu().numpy()
    tol = {"f16": 1e-3, "bf16": 6e-3}[dt]
    errs = []
    kept = tok
    for b in range(B):
        keep = ~frame_pad[b] if frame_pad is not None else np.ones(tok, bool)
        if keep.sum() == 0:
            continue
        kept = min(kept, int(keep.sum()))
        errs.append(rel(got[b][keep].mean(0), f[b][keep].mean(0)))
    if kept < 64:
        tol *= 3.0          # a handful of tokens: nothing averages out, the pooled error is the frame-level error (tests/test_gpu_e2e.py::test_edge_sizes)
    name = O.layer_names(cfg)[hook]
    hk = rel(r["hooks"][hook].cpu().numpy().mean(1), taps[name].mean(1)) if frame_pad is None else 0.0
    e = max(errs + [hk])
    worst = max(worst, e / tol)
    print(f"case {c:3d}: {dt} res={res} chunk={chunk} B={B} samples={n} tokens={tok} amp={amp:.3g} mask={'y' if mask is not None else 'n'} "
          f"hook={hook}  pooled {max(errs):.2e} hook {hk:.2e}  (tol {tol:g})  oracle {time.time() - t0:.1f} s", flush=True)
    if not np.isfinite(got).all() or e >= tol:
        print("VIOLATION"); sys.exit(1)
print(f"{cases} cases, worst error / tolerance = {worst:.2f}")
