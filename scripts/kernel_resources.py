#!/usr/bin/env python3
"""Registers / scratch / LDS / occupancy of every kernel of one csrc file, from hipcc's own remarks (no GPU needed):
    python scripts/kernel_resources.py gemm.hip [filter] [-D...]
One line per kernel: VGPRs, AGPRs, SGPRs, scratch bytes per lane (must be 0 on the layer loop), LDS bytes, waves per SIMD."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
flt = next((a for a in sys.argv[2:] if not a.startswith("-")), "")
extra = [a for a in sys.argv[2:] if a.startswith("-")]
path = src if os.path.exists(src) else os.path.join(ROOT, "avex_amd", "csrc", src)
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-x", "hip", "-fno-gpu-rdc", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"),
       "-Rpass-analysis=kernel-resource-usage", "-c", path, "-o", "/dev/null"] + extra
if os.path.basename(path) in ("fbank.hip", "wavconv.hip", "melspec.hip", "lstm.hip"):
    cmd.append("-fno-slp-vectorize")
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in err.splitlines():
    m = re.search(r"remark: .*?Function Name: (\S+)", line) or re.search(r"remark: .*? Name: (\S+)", line)
    if m:
        cur = m.group(1); rows[cur] = {}
        continue
    m = re.search(r"remark:\s+(VGPRs|AGPRs|SGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).split(" ")[0]] = int(m.group(2))
if not rows:
    print(err[-3000:]); sys.exit(1)
def shorten(n):      # _ZN12_GLOBAL__N_115gemm256p_kernelIDF16_Li1ELi1ELi1EEEvN3avx8GemmArgsE -> gemm256p_kernel<f16,1,1,1>  (c++filt does not know DF16_)
    m = re.match(r"_ZN?(?:12_GLOBAL__N_1)?(\d+)", n)
    if not m:
        return n
    k = int(m.group(1)); st = m.end(); name = n[st:st + k]; rest = n[st + k:]
    if not rest.startswith("I"):
        return name
    args, i = [], 1
    while i < len(rest) and rest[i] != "E":
        if rest.startswith("DF16_", i): args.append("f16"); i += 5
        elif rest.startswith("DF16b", i): args.append("bf16"); i += 5
        elif rest[i] == "L":
            j = rest.index("E", i); v = rest[i + 2:j]; args.append(("-" + v[1:]) if v.startswith("n") else v); i = j + 1
        else:
            args.append(rest[i]); i += 1
    return f"{name}<{','.join(args)}>"
for name, d in ((shorten(n), d) for n, d in rows.items()):
    short = name
    if flt in short:
        print(f"{short:60s} v{d.get('VGPRs', 0):3d} a{d.get('AGPRs', 0):3d} s{d.get('SGPRs', 0):3d} scratch {d.get('ScratchSize', 0):4d} lds {d.get('LDS', 0):6d} occ {d.get('Occupancy', 0)}")
