"""The C-ABI library loads and exports every symbol include/avexhip.h declares (no GPU needed)."""
import os
import re

from avex_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "avexhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(avexhip_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(built_lib):
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(built_lib, name), f"libavexhip.so does not export {name}"
    # the ctypes binding table covers the header exactly
    assert sorted(_capi.SYMBOLS) == declared


def test_host_only_entry_points(built_lib):
    assert built_lib.avexhip_abi_version() == 2
    assert built_lib.avexhip_device_count() >= 0
    assert built_lib.avexhip_rel_bucket(0, 320, 800) == 0
    assert built_lib.avexhip_rel_bucket(1, 320, 800) == 161
    assert built_lib.avexhip_rel_bucket(-495, 320, 800) == 143
    assert isinstance(_capi.last_error(), str)


def test_struct_layouts_match_header(tmp_path):
    """sizeof/offsetof of every ABI struct as gcc sees include/avexhip.h == the ctypes mirror."""
    import ctypes as C
    import shutil
    import subprocess
    import pytest
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    structs = {"avexhip_fbank_config": _capi.FbankConfig, "avexhip_gemm_args": _capi.GemmArgs,
               "avexhip_beats_config": _capi.BeatsConfig, "avexhip_tensor": _capi.Tensor}
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{ROOT}/include/avexhip.h"', "int main(void){"]
    for cname, cls in structs.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines.append("return 0;}")
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-o", str(exe), str(src)], check=True)
    out = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, cls in structs.items():
        assert int(out[cname]) == C.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(out[f"{cname}.{fname}"]) == getattr(cls, fname).offset, f"{cname}.{fname}"


def test_no_cpu_fallback_without_gpu(built_lib):
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from avex_amd import kernels
    with pytest.raises(_capi.AvexHipError):
        kernels.FbankPlan()
    with pytest.raises(_capi.AvexHipError):
        kernels.gemm(torch.zeros(8, 64, dtype=torch.float16), torch.zeros(128, 64, dtype=torch.float16))
