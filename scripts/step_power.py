#!/usr/bin/env python3
"""Board power and shader clock while the whole 256-clip step loops (sysfs hwmon sampled every 0.1 s), plus each kernel family
looped on its own: which parts of the step sit at the power cap (energy-bound) and which do not (cycle-bound)."""
import os, sys, subprocess, threading, time, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from avex_amd import synth, kernels as K

def _hwmon():
    """hwmon directory of torch's device 0 (sysfs; no child process from a GPU-initialised process)."""
    import glob
    pr = torch.cuda.get_device_properties(0)
    c = glob.glob(f"/sys/bus/pci/devices/{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0/hwmon/hwmon*")
    return c[0] if c else None

HW = None

def sampler(stop, out):
    global HW
    HW = HW or _hwmon()
    while not stop.is_set():
        try:
            pw = float(open(os.path.join(HW, "power1_input")).read()) * 1e-6
            sc = float(open(os.path.join(HW, "freq1_input")).read()) * 1e-6
            out.append((pw, int(sc)))
        except Exception:  # noqa: BLE001
            pass
        time.sleep(0.1)

def loop(name, fn, seconds=4.0):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    stop, s = threading.Event(), []
    th = threading.Thread(target=sampler, args=(stop, s)); th.start()
    t0 = time.time(); n = 0
    while time.time() - t0 < seconds:
        fn(); n += 1
        if n % 4 == 0: torch.cuda.synchronize()
    torch.cuda.synchronize(); el = time.time() - t0
    stop.set(); th.join()
    pw = [p for p, _ in s if p]; ck = [c for _, c in s if c]
    med = lambda v: sorted(v)[len(v) // 2] if v else None
    print(f"{name:28s} {1e3*el/n:8.3f} ms/iter  power median {med(pw)} W (max {max(pw) if pw else None})  sclk median {med(ck)} MHz", flush=True)

cfg = synth.BEATS_BASE_CFG
sd = synth.beats_state_dict(cfg, seed=0)
B, T, H, E = 256, 496, 12, 768
wav = (0.1 * torch.randn(B, 160000)).cuda()
enc = K.BeatsEncoder(cfg, sd, operand_dtype="f16", max_chunk_clips=256, residual="half")
loop("whole step (batch 256)", lambda: enc.forward(wav, want_features=False, want_pooled=True))
M = B * T
x = torch.randn(M, E, device="cuda").half(); w1 = (torch.randn(3072, E, device="cuda") * 0.05).half(); b1 = torch.randn(3072, device="cuda")
loop("gemm fc1 (gelu)", lambda: K.gemm(x, w1, bias=b1, gelu=True, out_f32=False, out_half=True))
qkv = torch.randn(M, 3 * E, device="cuda").half(); tab = torch.randn(H, 2 * T - 1, device="cuda") * 0.3
gw = torch.randn(8, 64, device="cuda") * 0.1; gb = torch.randn(8, device="cuda") * 0.1; ga = torch.ones(H, device="cuda")
loop("attention", lambda: K.attention(qkv, B, T, H, tab, gw, gb, ga))
g = torch.ones(E, device="cuda"); bb = torch.zeros(E, device="cuda")
loop("layernorm (half in, f32+half out)", lambda: K.layernorm(x, g, bb))
xh = x
loop("layernorm (half in, half out)", lambda: K.layernorm(xh, g, bb, want_f32=False))
w2 = (torch.randn(E, 3072, device="cuda") * 0.02).half(); hmid = torch.randn(M, 3072, device="cuda").half(); b2 = torch.randn(E, device="cuda")
loop("gemm fc2 (+half residual)", lambda: K.gemm(hmid, w2, bias=b2, resid_half=x, alpha=2.2, out_f32=False, out_half=True))
wq = (torch.randn(2304, E, device="cuda") * 0.05).half(); bq = torch.randn(2304, device="cuda")
loop("gemm qkv", lambda: K.gemm(x, wq, bias=bq, out_f32=False, out_half=True))
plan = K.FbankPlan()
loop("fbank (256 clips, f32 out)", lambda: plan(wav))
