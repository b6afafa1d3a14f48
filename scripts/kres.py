#!/usr/bin/env python3
"""Register / scratch / LDS usage of every kernel of one translation unit (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel.

    python scripts/kres.py avex_amd/csrc/gemm.hip [extra hipcc flags]
"""
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avex_amd import build as B  # noqa: E402


def main() -> int:
    src = sys.argv[1]
    extra = sys.argv[2:]
    flags = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-x", "hip", "-Wno-unused-result", "-fno-gpu-rdc", "-ffp-contract=off",
             "-Rpass-analysis=kernel-resource-usage"] + B.EXTRA_FLAGS.get(os.path.basename(src), []) + extra
    r = subprocess.run([B.hipcc()] + flags + ["-c", src, "-o", "/dev/null"], capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stderr[-4000:])
        return 1
    cur = None
    rows = {}
    for ln in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", ln)
        if m:
            cur = m.group(1)
            rows[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z][A-Za-z ]*?)(?: \[[^\]]*\])?: (\d+)", ln)
        if m and cur:
            rows[cur][m.group(1).strip()] = int(m.group(2))
    for name, d in rows.items():
        m = re.match(r"_ZN?(?:12_GLOBAL__N_1)?(\d+)", name)          # _ZN12_GLOBAL__N_115gemm256p_kernelIDF16_Li1EEEv... -> gemm256p_kernel<DF16_,1>
        dem = name
        if m:
            n = int(m.group(1))
            rest = name[m.end():]
            targs = re.match(r"I(.*?)E+v", rest[n:])
            dem = rest[:n] + ("<" + re.sub(r"L[ib](\d+)E", r",\1", targs.group(1)).replace("DF16_", "f16").replace("DF16b", "bf16") + ">" if targs else "")
        print(f"{dem[:70]:70s} VGPR {d.get('VGPRs', -1):3d} AGPR {d.get('AGPRs', -1):3d} SGPR {d.get('TotalSGPRs', -1):3d} scratch {d.get('ScratchSize', -1):4d} spill {d.get('VGPRs Spill', -1)} "
              f"occ {d.get('Occupancy', -1)} LDS {d.get('LDS Size', -1)}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
