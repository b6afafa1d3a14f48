"""Reject gfx950 instruction forms that compute wrong values when a kernel shares a CU with matrix work.

Measured on MI355X (ROCm 7.2; ``scripts/debug/conc_probe5.hip`` / ``conc_probe6.hip``, results in
``profiles/r02a_pk_opsel_erratum.txt``): a packed-fp32 arithmetic instruction ``v_pk_add_f32`` / ``v_pk_mul_f32`` /
``v_pk_fma_f32`` whose SECOND source takes its HIGH register for the LOW result while the first source does not
(``op_sel:[0,1]`` / ``op_sel:[0,1,x]``) reads that register as 0.0 in lanes 32-63 in about 4 % of its executions while ANOTHER
wave on the CU is issuing MFMA instructions -- no dependency hazard is involved (operands settled for 8 wait states) and the
instruction is exact when the CU runs no matrix work.  ``op_sel:[1,0]``, ``[1,1]``, every ``op_sel_hi`` form, ``v_pk_mov_b32``
and ``v_mov_b64`` are unaffected.  hipcc's SLP vectoriser emits the bad form for complex (float2) butterflies: that was round 1's
"fbank kernel is wrong beside a GEMM" bug.

``check_library()`` disassembles every gfx950 code object bundled in the shared library and fails on the form, so a compiler
or source change cannot bring it back unnoticed (called by ``avex_amd.build`` after linking and by ``tests/test_isa_lint.py``).
"""
from __future__ import annotations

import os
import re
import struct
import subprocess
import tempfile
from typing import List, Tuple

LLVM_BIN = os.environ.get("AVEX_AMD_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
# low-result operand select: src0 = 0, src1 = 1 (third entry, when present, is src2 and harmless)
_BAD = re.compile(r"\bv_pk_(?:add|mul|fma)_f32\b.*\bop_sel:\[0,1(?:,[01])?\]")


def _fatbin(lib_path: str) -> bytes:
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "fat.bin")
        subprocess.run([os.path.join(LLVM_BIN, "llvm-objcopy"), f"--dump-section=.hip_fatbin={out}", lib_path], check=True, capture_output=True)
        with open(out, "rb") as f:
            return f.read()


def device_code_objects(lib_path: str, arch: str = "gfx950") -> List[bytes]:
    """Every code object for ``arch`` in the library's .hip_fatbin (one clang offload bundle per translation unit)."""
    blob = _fatbin(lib_path)
    objs: List[bytes] = []
    pos = blob.find(_MAGIC)
    while pos >= 0:
        (n,) = struct.unpack_from("<Q", blob, pos + len(_MAGIC))
        cur = pos + len(_MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, cur)
            triple = blob[cur + 24: cur + 24 + tlen].decode("ascii", "replace")
            cur += 24 + tlen
            if arch in triple and size:
                objs.append(blob[pos + off: pos + off + size])
        pos = blob.find(_MAGIC, pos + len(_MAGIC))
    return objs


def disassemble(code_object: bytes) -> str:
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "dev.co")
        with open(p, "wb") as f:
            f.write(code_object)
        r = subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "-d", "--no-show-raw-insn", p], check=True, capture_output=True, text=True)
        return r.stdout


def find_bad_instructions(lib_path: str) -> List[Tuple[str, str]]:
    """[(kernel symbol, instruction text)] for every occurrence of the forbidden form."""
    hits: List[Tuple[str, str]] = []
    for co in device_code_objects(lib_path):
        sym = "?"
        for line in disassemble(co).splitlines():
            m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
            if m:
                sym = m.group(1)
            elif _BAD.search(line):
                hits.append((sym, line.strip()))
    return hits


def check_library(lib_path: str) -> int:
    """Raise RuntimeError if the library contains the forbidden form; returns the number of code objects checked."""
    objs = device_code_objects(lib_path)
    if not objs:
        raise RuntimeError(f"{lib_path}: no gfx950 code object found in .hip_fatbin")
    hits = find_bad_instructions(lib_path)
    if hits:
        lines = "\n".join(f"  {s}: {i}" for s, i in hits[:20])
        raise RuntimeError(f"{lib_path}: {len(hits)} packed-fp32 instruction(s) with op_sel:[0,1] (wrong beside MFMA work on gfx950, "
                           f"see avex_amd/isa_lint.py):\n{lines}")
    return len(objs)


if __name__ == "__main__":
    import sys
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libavexhip.so")
    print(f"{path}: {check_library(path)} gfx950 code objects, no forbidden packed-fp32 operand selects")
