#!/usr/bin/env python3
"""A/B of the whole 256-clip BEATs step inside one process: two handles created under two environments (the knobs are read when a
handle is created), timed alternately so that box, clocks and temperature are shared.

    python scripts/ab_step.py --a AVEX_AMD_LN_FOLD=0 --b AVEX_AMD_LN_FOLD=1 [--pairs 3] [--steps 20] [--batch 256] [--parity]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from avex_amd import kernels as K
from avex_amd import synth


def make(envs, cfg, sd, dtype):
    old = {}
    for kv in filter(None, envs.split(",")):
        k, v = kv.split("=", 1)
        old[k] = os.environ.get(k)
        os.environ[k] = v
    try:
        return K.BeatsEncoder(cfg, sd, operand_dtype=dtype)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--a", default="")
    ap.add_argument("--b", default="")
    ap.add_argument("--pairs", type=int, default=3)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--parity", action="store_true", help="pooled rel-L2 of both against the reference golden (rows 0-3)")
    ap.add_argument("--live", action="store_true", help="also set each side's environment around its timed forwards (knobs the launchers read per launch, e.g. AVEX_AMD_ATT_VARIANT)")
    args = ap.parse_args()
    cfg = synth.BEATS_BASE_CFG
    sd = synth.beats_state_dict(cfg, seed=0)
    wav = torch.from_numpy(synth.noise_clips(args.batch, 160000, seed=0)).cuda()
    encs = {"A": make(args.a, cfg, sd, args.dtype), "B": make(args.b, cfg, sd, args.dtype)}
    envs = {"A": args.a, "B": args.b}

    class live:      # the side's environment for the duration of a block (--live)
        def __init__(self, name): self.kv = [kv.split("=", 1) for kv in filter(None, envs[name].split(","))] if args.live else []
        def __enter__(self):
            self.old = {k: os.environ.get(k) for k, _ in self.kv}
            for k, v in self.kv: os.environ[k] = v
        def __exit__(self, *a):
            for k, v in self.old.items():
                if v is None: os.environ.pop(k, None)
                else: os.environ[k] = v
    print(f"A: {args.a or '(defaults)'}   B: {args.b or '(defaults)'}", flush=True)
    for name, e in encs.items():
        with live(name):
            for _ in range(3):
                e.forward(wav, want_features=False, want_pooled=True)
    torch.cuda.synchronize()
    res = {"A": [], "B": []}
    for p in range(args.pairs):
        for name in ("A", "B") if p % 2 == 0 else ("B", "A"):
            e = encs[name]
            torch.cuda.synchronize()
            with live(name):
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    e.forward(wav, want_features=False, want_pooled=True)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            res[name].append(args.batch * args.steps / dt)
            print(f"pair {p} {name}: {res[name][-1]:9.1f} clips/s  ({1e3 * dt / args.steps:.3f} ms/step)", flush=True)
    ma, mb = float(np.median(res["A"])), float(np.median(res["B"]))
    print(f"median A {ma:.1f}  B {mb:.1f}  B/A {mb / ma:.4f}")
    if args.parity:
        g = np.load(os.path.join(ROOT, "tests", "golden", "base_api.npz"))["b4.pooled"]
        x = wav.clone()
        x[:4] = torch.from_numpy(synth.noise_clips(4, 160000, seed=0)).cuda()
        for name, e in encs.items():
            with live(name):
                pz = e.forward(x, want_features=False, want_pooled=True)["pooled"][:4].cpu().numpy()
            print(f"parity {name}: pooled rel-L2 vs the reference golden {np.linalg.norm(pz - g) / np.linalg.norm(g):.3e}; overflow events {e.overflow_events()}")
    for e in encs.values():
        prof_on = getattr(e, "set_profiling", None)
    for name, e in encs.items():
        e.set_profiling(True)
        with live(name):
            e.forward(wav, want_features=False, want_pooled=True)
        pr = e.last_profile()
        e.set_profiling(False)
        print(f"stages {name}: " + "  ".join(f"{n} {ms:.3f}" for n, ms, _ in pr))


if __name__ == "__main__":
    main()
