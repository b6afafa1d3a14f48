"""Reject gfx950 instruction forms that compute wrong values when a kernel shares a CU with matrix work.

Measured on MI355X (ROCm 7.2; ``scripts/debug/conc_probe5.hip`` / ``conc_probe6.hip``, results in
``profiles/r02a_pk_opsel_erratum.txt``): a packed-fp32 arithmetic instruction ``v_pk_add_f32`` / ``v_pk_mul_f32`` /
``v_pk_fma_f32`` whose SECOND source takes its HIGH register for the LOW result while the first source does not
(``op_sel:[0,1]`` / ``op_sel:[0,1,x]``) reads that register as 0.0 in lanes 32-63 in about 4 % of its executions while ANOTHER
wave on the CU is issuing MFMA instructions -- no dependency hazard is involved (operands settled for 8 wait states) and the
instruction is exact when the CU runs no matrix work.  ``op_sel:[1,0]``, ``[1,1]``, every ``op_sel_hi`` form, ``v_pk_mov_b32``
and ``v_mov_b64`` are unaffected.  hipcc's SLP vectoriser emits the bad form for complex (float2) butterflies: that was round 1's
"fbank kernel is wrong beside a GEMM" bug.

Second rule (round 4, ``profiles/r04d_store_hazard.txt``): a buffer / global store of MORE than 64 bits whose data registers a VALU
instruction overwrites within the next two instructions.  hipcc inserts the two wait states the hardware needs -- except when the
store's scalar-offset operand is a register (LLVM's hazard recogniser: "this hazard only exists if the instruction is not using a
register in the soffset field").  On gfx950 it exists there too: ``buffer_store_dwordx4 v[48:51], v120, s[28:31], s65 offen nt``
directly followed by ``v_pk_mul_f32 v[48:49], ...`` stored the new value of v49 for lanes 12-15 of every 16.  The kernels keep that
operand 0 (``gemm.hip`` ``buf_st16``); the lint catches whatever form brings the pattern back.

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


_WIDE_STORE = re.compile(r"^\s*(?:buffer|global|flat|scratch)_store_dwordx[34]\s+(.*)$")
_VREG = re.compile(r"\bv(?:\[(\d+):(\d+)\]|(\d+))")


def _vregs(operand: str) -> set:
    out = set()
    for m in _VREG.finditer(operand):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def _store_data_regs(line: str) -> set:
    """Data VGPRs of a wide store: the first operand of buffer_store (vdata, vaddr, srsrc, soffset), the second of global / flat / scratch (vaddr, vdata, saddr)."""
    m = _WIDE_STORE.match(line)
    if not m:
        return set()
    ops = [o.strip() for o in m.group(1).split(",")]
    is_buffer = line.lstrip().startswith("buffer_")
    return _vregs(ops[0] if is_buffer else (ops[1] if len(ops) > 1 else ""))


def _valu_dest_regs(line: str) -> set:
    t = line.strip()
    if not t.startswith("v_") or t.startswith(("v_cmp", "v_cmpx", "v_nop", "v_readlane", "v_readfirstlane")):
        return set()
    first = t.split(None, 1)[1].split(",")[0] if " " in t else ""
    return _vregs(first)


def find_store_hazards(text: str) -> List[Tuple[str, str]]:
    """Wide stores whose data registers are overwritten by a VALU instruction less than two wait states later."""
    hits: List[Tuple[str, str]] = []
    sym = "?"
    pending: List[Tuple[set, int, str]] = []          # (data registers, wait states seen since, store text)
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m:
            sym, pending = m.group(1), []
            continue
        body = line.split("//")[0].strip()
        if not body or body.endswith(":"):
            continue
        if body.startswith(("s_endpgm", "s_branch", "s_setpc_b64", "s_swappc_b64")):      # the listing continues with another block
            pending = []
            continue
        dest = _valu_dest_regs(body)
        for regs, _, st in pending:
            if dest & regs:
                hits.append((sym, f"{st}  ->  {body}"))
        nop = re.match(r"s_nop\s+(\d+)", body)
        step = int(nop.group(1)) + 1 if nop else 1
        pending = [(r, n + step, st) for r, n, st in pending if n + step < 2]
        regs = _store_data_regs(body)
        if regs:
            pending.append((regs, 0, body))
    return hits


def find_bad_instructions(lib_path: str) -> List[Tuple[str, str]]:
    """[(kernel symbol, instruction text)] for every occurrence of a forbidden form."""
    hits: List[Tuple[str, str]] = []
    for co in device_code_objects(lib_path):
        sym = "?"
        text = disassemble(co)
        for line in text.splitlines():
            m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
            if m:
                sym = m.group(1)
            elif _BAD.search(line):
                hits.append((sym, line.strip()))
        hits += find_store_hazards(text)
    return hits


def check_library(lib_path: str) -> int:
    """Raise RuntimeError if the library contains the forbidden form; returns the number of code objects checked."""
    objs = device_code_objects(lib_path)
    if not objs:
        raise RuntimeError(f"{lib_path}: no gfx950 code object found in .hip_fatbin")
    hits = find_bad_instructions(lib_path)
    if hits:
        lines = "\n".join(f"  {s}: {i}" for s, i in hits[:20])
        raise RuntimeError(f"{lib_path}: {len(hits)} forbidden instruction form(s) -- packed fp32 with op_sel:[0,1] (wrong beside MFMA work on "
                           f"gfx950) or a wide store whose data registers are overwritten too early (see avex_amd/isa_lint.py):\n{lines}")
    return len(objs)


if __name__ == "__main__":
    import sys
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libavexhip.so")
    print(f"{path}: {check_library(path)} gfx950 code objects, no forbidden packed-fp32 operand selects")
