"""The stand-alone microbenchmarks under scripts/micro/ are evidence DESIGN.md and profiles/ cite: they must keep compiling against the
kernels' shared headers (round 6: two of them had stopped).  Compile-only for gfx950 -- hipcc cross-compiles without a GPU."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(900)
def test_every_microbenchmark_compiles_for_gfx950():
    if not (shutil.which("make") and os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("needs make and hipcc")
    d = os.path.join(ROOT, "scripts", "micro")
    r = subprocess.run(["make", "-C", d, "-j4", "check"], capture_output=True, text=True, timeout=850)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    n = len([f for f in os.listdir(d) if f.endswith(".hip")])
    assert f"{n} microbenchmarks compile for gfx950" in r.stdout
