#!/usr/bin/env python3
"""Is the GEMM power/clock bound?  Same launches on random and on all-zero operands (DVFS gives the clock back on zeros),
with board power sampled from rocm-smi while a 3-second loop runs.   python scripts/gemm_power.py [variant]"""
import os, sys, subprocess, threading, time, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from avex_amd import kernels as K
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 0
M = 256 * 496

def power_sampler(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
            pw = re.findall(r"Power \(W\):\s*([0-9.]+)", r)
            sclk = re.findall(r"sclk clock level:.*\((\d+)Mhz\)", r)
            out.append((float(pw[0]) if pw else None, int(sclk[0]) if sclk else None))
        except Exception as e:  # noqa: BLE001
            out.append((None, str(e)[:40]))
        time.sleep(0.2)

for (N, Kd, gelu) in ((2304, 768, False), (3072, 768, True), (768, 3072, False)):
    for fill in ("random", "zeros"):
        if fill == "random":
            x = torch.randn(M, Kd, device="cuda").half(); w = (torch.randn(N, Kd, device="cuda") * 0.05).half()
        else:
            x = torch.zeros(M, Kd, device="cuda").half(); w = torch.zeros(N, Kd, device="cuda").half()
        bias = torch.randn(N, device="cuda")
        kw = dict(bias=bias, variant=variant, out_f32=False, out_half=True, gelu=gelu)
        for _ in range(20): K.gemm(x, w, **kw)
        torch.cuda.synchronize()
        stop, samples = threading.Event(), []
        th = threading.Thread(target=power_sampler, args=(stop, samples)); th.start()
        t0 = time.time(); n = 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        while time.time() - t0 < 3.0:
            for _ in range(50): K.gemm(x, w, **kw)
            n += 50
            torch.cuda.synchronize()
        e1.record(); torch.cuda.synchronize()
        stop.set(); th.join()
        ms = e0.elapsed_time(e1) / n
        pws = [p for p, _ in samples if p]
        clk = [c for _, c in samples if isinstance(c, int)]
        print(f"N={N} K={Kd} gelu={int(gelu)} {fill:6s}: {ms*1e3:8.1f} us {2.0*M*N*Kd/ms/1e9:7.1f} TF  power W {pws[-4:] if pws else samples[-1:]}  sclk {clk[-4:]}", flush=True)
