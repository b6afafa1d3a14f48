"""The C ABI used from a program that is neither Python nor torch: tests/c_abi/abi_smoke.cpp is compiled with hipcc against
include/avexhip.h + libavexhip.so and run on the GPU (device buffers from hipMalloc, GEMM + GELU, LayerNorm, mean pooling,
the error path)."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_consumer(built_lib, tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = tmp_path / "abi_smoke"
    lib = os.path.join(ROOT, "avex_amd", "lib")
    cmd = [hipcc, "-O1", "--offload-arch=gfx950", os.path.join(ROOT, "tests", "c_abi", "abi_smoke.cpp"), "-I", os.path.join(ROOT, "include"),
           "-L", lib, "-lavexhip", f"-Wl,-rpath,{lib}", "-o", str(exe)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "C ABI SMOKE OK" in r.stdout, r.stdout + r.stderr
