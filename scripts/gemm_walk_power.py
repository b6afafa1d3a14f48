#!/usr/bin/env python3
"""One layer GEMM in the step's form (scripts/gemm_forms.py: folded LayerNorm), looped for 3 s with board power and shader clock sampled
from rocm-smi: time per launch, watts, MHz, joules per launch.  The tile walk comes from AVEX_AMD_GEMM_TILE_ORDER, the kernel build from
AVEX_AMD_LIB (round 6: the column-group walk with / without the non-temporal A stream, scripts/walks_r06.sh).
    python scripts/gemm_walk_power.py qkv|fc1|out|fc2 [label]"""
import os, sys, subprocess, threading, time, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from avex_amd import kernels as K
name = sys.argv[1]
label = sys.argv[2] if len(sys.argv) > 2 else ""
M = 256 * 496
N, Kd = {"qkv": (2304, 768), "out": (768, 768), "fc1": (3072, 768), "fc2": (768, 3072)}[name]
torch.manual_seed(0)
x = torch.randn(M, Kd, device="cuda").half(); w = (torch.randn(N, Kd, device="cuda") * 0.05).half(); bias = torch.randn(N, device="cuda")
rows = torch.stack([torch.rand(M + 1, device="cuda") + 0.5, torch.randn(M + 1, device="cuda") * 0.1], 1).contiguous()
kw = dict(bias=bias, out_f32=False, out_half=True)
if name in ("qkv", "fc1"):
    kw.update(ln_rows=rows, ln_s=torch.randn(N, device="cuda"))
    if name == "fc1":
        kw["gelu"] = True
else:
    kw.update(alpha=2.2, lnr_y=torch.randn(M, N, device="cuda").half(), lnr_rows=rows[:M].contiguous(), lnr_gamma=torch.rand(N, device="cuda") + 0.5,
              lnr_beta=torch.randn(N, device="cuda"), stats_out=True)


def sampler(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
            pw = re.findall(r"Power \(W\):\s*([0-9.]+)", r); sclk = re.findall(r"sclk clock level:.*\((\d+)Mhz\)", r)
            out.append((float(pw[0]) if pw else None, int(sclk[0]) if sclk else None))
        except Exception:  # noqa: BLE001
            pass
        time.sleep(0.15)


for _ in range(30):
    K.gemm(x, w, **kw)
torch.cuda.synchronize()
stop, samples = threading.Event(), []
th = threading.Thread(target=sampler, args=(stop, samples)); th.start()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.time(); n = 0
e0.record()
while time.time() - t0 < 3.0:
    for _ in range(50):
        K.gemm(x, w, **kw)
    n += 50
    torch.cuda.synchronize()
e1.record(); torch.cuda.synchronize()
stop.set(); th.join()
us = e0.elapsed_time(e1) / n * 1e3
pw = sorted(p for p, _ in samples[2:] if p); ck = sorted(c for _, c in samples[2:] if c)
pm = pw[len(pw) // 2] if pw else float("nan"); cm = ck[len(ck) // 2] if ck else 0
print(f"{name:4s} {label:44s} {us:7.1f} us  {2.0 * M * N * Kd / us / 1e6:7.1f} TF/s  {pm:6.0f} W  {cm:5d} MHz  {pm * us * 1e-6:6.3f} J per launch", flush=True)
