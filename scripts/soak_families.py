#!/usr/bin/env python3
"""Back-to-back forwards of the EAT, AVES and EfficientNet handles for a while: every pooled result against the first, bit for
bit (no family keeps a float atomic: the EfficientNet squeeze sums are per-workgroup partials added in order)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from avex_amd import synth, kernels as K
from avex_amd.eat_encoder import EatEncoder
from avex_amd.aves_encoder import AvesEncoder
from avex_amd.effnet_encoder import EfficientNetB0Encoder

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0


def soak(name, step, exact):
    first = step().clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 0; bad = 0; worst = 0.0
    while time.perf_counter() - t0 < secs:
        out = step()
        if exact:
            bad += int(not torch.equal(out, first))
        else:
            worst = max(worst, float((out - first).norm() / first.norm()))
        n += 1
    torch.cuda.synchronize()
    print(f"{name}: {n} steps in {secs:.0f} s, " + (f"{bad} differed from the first" if exact else f"largest relative difference to the first {worst:.2e}"))


eat = EatEncoder(synth.EAT_BASE_CFG, synth.eat_state_dict(), operand_dtype="f16")
w5 = torch.from_numpy(synth.noise_clips(128, 80000, seed=0)).cuda()
soak("EAT-base 128 x 5 s", lambda: eat.forward(w5, want_features=False, pooling="mean")["pooled"], True)
del eat
aves = AvesEncoder(synth.AVES_BASE_CFG, synth.aves_state_dict(), operand_dtype="f16")
w10 = torch.from_numpy(synth.noise_clips(64, 160000, seed=1)).cuda()
soak("AVES 64 x 10 s", lambda: aves.forward(w10, want_features=False, want_pooled=True)["pooled"], True)
del aves
eff = EfficientNetB0Encoder(synth.effnet_b0_state_dict())
plan = K.MelspecPlan(n_fft=800, hop_length=160, n_mels=128, normalize=True)
soak("EfficientNet-B0 128 x 10 s", lambda: eff.forward(plan(w10.repeat(2, 1)), want_features=False, want_pooled=True)["pooled"], True)
