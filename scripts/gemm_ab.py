#!/usr/bin/env python3
"""A/B of two builds of the library INSIDE ONE PROCESS (same box, same clocks, same temperature), alternating launch by launch groups:
the four layer GEMMs in the forms the 256-clip step launches them (see scripts/gemm_forms.py), or the whole BEATs step.

    python scripts/gemm_ab.py --a avex_amd/lib/libavexhip_prev.so --b avex_amd/lib/libavexhip.so [--shapes qkv,out,fc1,fc2] [--rounds 6] [--iters 20] [--step]

Both libraries are loaded RTLD_LOCAL | RTLD_DEEPBIND so that each resolves its own symbols; `avex_amd._capi` is pointed at one or
the other before every group of calls.  Reports the per-round times and the median of the B / A ratios.
"""
import argparse, ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from avex_amd import _capi, kernels as K, synth

ap = argparse.ArgumentParser()
ap.add_argument("--a", required=True)
ap.add_argument("--b", required=True)
ap.add_argument("--shapes", default="qkv,out,fc1,fc2")
ap.add_argument("--rounds", type=int, default=6)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--clips", type=int, default=256)
ap.add_argument("--step", action="store_true", help="also A/B the whole 256-clip BEATs step (one handle per library)")
ap.add_argument("--env-a", default="", help="KEY=VAL,... set while A's calls are made (knobs read per launch)")
ap.add_argument("--env-b", default="")
a = ap.parse_args()


def load(path):
    h = C.CDLL(os.path.abspath(path), mode=os.RTLD_LOCAL | os.RTLD_DEEPBIND)
    for name, (res, args) in _capi.SYMBOLS.items():
        fn = getattr(h, name)
        fn.restype, fn.argtypes = res, args
    return h


LIBS = {"A": load(a.a), "B": load(a.b)}
ENVS = {"A": dict(kv.split("=", 1) for kv in a.env_a.split(",") if kv), "B": dict(kv.split("=", 1) for kv in a.env_b.split(",") if kv)}


class use:
    def __init__(self, which):
        self.which = which

    def __enter__(self):
        _capi._lib = LIBS[self.which]
        self.old = {k: os.environ.get(k) for k in ENVS[self.which]}
        os.environ.update(ENVS[self.which])

    def __exit__(self, *exc):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


M = a.clips * 496
dev = "cuda"
torch.manual_seed(0)


def make(name):
    N, Kd = {"qkv": (2304, 768), "out": (768, 768), "fc1": (3072, 768), "fc2": (768, 3072)}[name]
    x = torch.randn(M, Kd, device=dev).half()
    w = (torch.randn(N, Kd, device=dev) * 0.05).half()
    kw = dict(bias=torch.randn(N, device=dev), out_f32=False, out_half=True)
    rows = torch.stack([torch.rand(M + 1, device=dev) + 0.5, torch.randn(M + 1, device=dev) * 0.1], 1).contiguous()
    if name in ("qkv", "fc1"):
        kw.update(ln_rows=rows, ln_s=torch.randn(N, device=dev))
        if name == "fc1":
            kw["gelu"] = True
    else:
        kw.update(alpha=2.2, lnr_y=torch.randn(M, N, device=dev).half(), lnr_rows=rows[:M].contiguous(), lnr_gamma=torch.rand(N, device=dev) + 0.5,
                  lnr_beta=torch.randn(N, device=dev), stats_out=True)
    return x, w, kw


def timed(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for name in [s for s in a.shapes.split(",") if s]:
    x, w, kw = make(name)
    outs = {}
    for which in "AB":
        with use(which):
            for _ in range(5):
                r = K.gemm(x, w, **kw)
            outs[which] = {k: v.clone() for k, v in r.items()}
    torch.cuda.synchronize()
    same = all(torch.equal(outs["A"][k], outs["B"][k]) for k in outs["A"])
    t = {"A": [], "B": []}
    for rnd in range(a.rounds):
        for which in ("AB" if rnd % 2 == 0 else "BA"):
            with use(which):
                t[which].append(timed(lambda: K.gemm(x, w, **kw), a.iters))
    ra = np.array(t["B"]) / np.array(t["A"])
    print(f"{name:4s} A {np.median(t['A']):7.1f} us  B {np.median(t['B']):7.1f} us   B/A median {np.median(ra):.4f} (min {ra.min():.4f} max {ra.max():.4f})   outputs bit-identical: {same}", flush=True)
    del x, w, kw, outs

if a.step:
    cfg = synth.BEATS_BASE_CFG
    sd = synth.beats_state_dict(cfg, seed=0)
    wav = torch.from_numpy(synth.noise_clips(a.clips, 160000, seed=0)).cuda()
    encs = {}
    for which in "AB":
        with use(which):
            encs[which] = K.BeatsEncoder(cfg, sd, operand_dtype="f16")
            for _ in range(3):
                p = encs[which].forward(wav, want_features=False, want_pooled=True)["pooled"]
            encs[which + "out"] = p.clone()
    torch.cuda.synchronize()
    t = {"A": [], "B": []}
    for rnd in range(a.rounds):
        for which in ("AB" if rnd % 2 == 0 else "BA"):
            with use(which):
                t[which].append(timed(lambda: encs[which].forward(wav, want_features=False, want_pooled=True), 10) / 1e3)
    ra = np.array(t["B"]) / np.array(t["A"])
    d = (encs["Aout"] - encs["Bout"]).abs().max().item()
    print(f"step A {np.median(t['A']):7.3f} ms ({a.clips / np.median(t['A']) * 1e3:.0f} clips/s)  B {np.median(t['B']):7.3f} ms ({a.clips / np.median(t['B']) * 1e3:.0f} clips/s)   "
          f"B/A median {np.median(ra):.4f} (min {ra.min():.4f} max {ra.max():.4f})   max |pooled A - B| {d:.3e}", flush=True)
    for which in "AB":
        with use(which):
            encs[which].close()
