#!/usr/bin/env python3
"""Differential fuzzing of the EfficientNet handle: random stage layouts (expansion ratio 1 / 4 / 6, 3 x 3 and 5 x 5, stride 1 and 2,
8 .. 96 channels: the fused block front at K = 32 and 64, the LDS depthwise kernel on every block it can take, narrow 32-channel
outputs, residual blocks), random image sizes (ragged tiles on both axes), both operand types.  Each case runs the shipped kernels
against the round-3 forms of the same library (AVEX_AMD_MBCONV=0 AVEX_AMD_DW_LDS=0) and against the NumPy oracle, and repeats the
shipped forward (bit for bit).
    python tests/tools/fuzz_effnet.py [cases] [seed]
Prints one line per case and the worst ratio to the tolerance; exits 1 on the first violation."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from avex_amd import synth
from avex_amd.effnet_encoder import EfficientNetB0Encoder
from oracle import effnet_oracle as EO

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
rel = lambda a, b: float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
WIDTHS = [8, 16, 24, 32, 40, 48, 64, 80, 96]
worst = 0.0
for c in range(cases):
    n_stages = int(rng.integers(2, 6))
    stages, cin, n_s2 = [], 32, 0                                # (the wrapper fixes the stem at 32 channels and the head at 1280, like B0 / B1)
    for si in range(n_stages):
        er = int(rng.choice([1, 4, 6])) if si else int(rng.choice([1, 6]))
        k = int(rng.choice([3, 5]))
        st = int(rng.choice([1, 2])) if n_s2 < 3 else 1
        n_s2 += st == 2
        cout = int(rng.choice(WIDTHS))
        stages.append((er, k, st, cin, cout, int(rng.integers(1, 3))))
        cin = cout
    head = 1280
    H, W = int(rng.integers(33, 90)), int(rng.integers(40, 160))
    B = int(rng.integers(1, 4))
    dt = ["f16", "bf16"][int(rng.integers(0, 4) == 0)]
    sd = synth.effnet_b0_state_dict(seed=c, stages=stages, head=head)
    mel = np.abs(rng.standard_normal((B, H, W))).astype(np.float32) * np.float32(0.5)
    x = torch.from_numpy(mel).cuda()
    enc = EfficientNetB0Encoder(sd, operand_dtype=dt, stages=stages)
    names = enc.tap_names()
    os.environ["AVEX_AMD_DW_LDS"] = "2"
    a = enc.forward(x, hook_layers=names, want_features=True, want_pooled=True)
    a2 = enc.forward(x, hook_layers=names, want_features=True, want_pooled=True)
    os.environ["AVEX_AMD_MBCONV"] = "0"; os.environ["AVEX_AMD_DW_LDS"] = "0"
    b = enc.forward(x, hook_layers=names, want_features=True, want_pooled=True)
    del os.environ["AVEX_AMD_MBCONV"]; del os.environ["AVEX_AMD_DW_LDS"]
    d = enc.forward(x, hook_layers=names, want_features=True, want_pooled=True)      # the shipped per-layer choice
    ref, taps = EO.effnet_features(mel, sd, stages)
    tol_ab = 1e-3 if dt == "f16" else 8e-3          # operand roundings that flip because the squeeze sums are added in another order (f16: 5e-4 per flipped element)
    tol_or = 1.5e-3 if dt == "f16" else 1.2e-2
    e_ab = max([rel(a["hooks"][n].cpu().numpy(), b["hooks"][n].cpu().numpy()) for n in names[1:]] + [rel(a["pooled"].cpu().numpy(), b["pooled"].cpu().numpy())])
    e_d = rel(d["pooled"].cpu().numpy(), b["pooled"].cpu().numpy())
    e_or = max(rel(a["features"].cpu().numpy(), ref), max(rel(a["hooks"][n].cpu().numpy(), taps[n]) for n in names[1:]))
    same = torch.equal(a["features"], a2["features"]) and all(torch.equal(a["hooks"][n], a2["hooks"][n]) for n in names)
    ok = e_ab < tol_ab and e_d < tol_ab and e_or < tol_or and same and bool(torch.isfinite(a["features"]).all())
    worst = max(worst, e_ab / tol_ab, e_d / tol_ab, e_or / tol_or)
    print(f"case {c:3d}: {dt} B={B} {H}x{W} stages={stages}  fused/unfused {e_ab:.2e} default/unfused {e_d:.2e} (tol {tol_ab})  oracle {e_or:.2e} (tol {tol_or})  repeat {'ok' if same else 'DIFFERS'}", flush=True)
    if not ok:
        print("VIOLATION"); sys.exit(1)
    del enc
print(f"{cases} cases, worst error / tolerance = {worst:.2f}")
