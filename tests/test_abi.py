"""The C-ABI library loads and exports every symbol include/avexhip.h declares (no GPU needed)."""
import os
import re

from avex_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "avexhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"#ifdef AVEX_DIAG.*?#endif", "", text, flags=re.S)      # diagnostic-build-only exports
    return sorted(set(re.findall(r"\b(avexhip_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(built_lib):
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(built_lib, name), f"libavexhip.so does not export {name}"
    # the ctypes binding table covers the header exactly
    assert sorted(_capi.SYMBOLS) == declared


def test_product_library_has_no_debug_exports(built_lib):
    """Clock stamps, canaries and debug entry points live behind -DAVEX_DIAG (a separate library); the product one has none."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _capi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    names = [ln.split()[-1] for ln in out.splitlines() if ln.strip()]
    assert not [n for n in names if "debug" in n.lower()], [n for n in names if "debug" in n.lower()]
    assert not hasattr(built_lib, "avexhip_debug_gemm_stamps")


def test_host_only_entry_points(built_lib):
    assert built_lib.avexhip_abi_version() == _capi.header_abi_version()
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


def test_rel_bucket_product_function_matches_reference_golden(built_lib, golden_dir):
    """The library's own bucket function (what the Toeplitz bias tables are built from) against the reference's
    _relative_positions_bucket (backbone.py:438-473) for every offset a 496-token clip has -- bit-exact -- and against the
    oracle restatement beyond it (the > 512-token range has no golden)."""
    import numpy as np
    from oracle import beats_oracle as O
    g = np.load(f"{golden_dir}/base_api.npz")["bucket_rel_-495..495"]
    got = np.array([built_lib.avexhip_rel_bucket(d, 320, 800) for d in range(-495, 496)], dtype=np.int64)
    assert np.array_equal(got, g.astype(np.int64))
    rel = np.arange(-4000, 4001)
    want = O.relative_position_bucket(rel, 320, 800)
    got = np.array([built_lib.avexhip_rel_bucket(int(d), 320, 800) for d in rel], dtype=np.int64)
    assert np.array_equal(got, np.asarray(want, dtype=np.int64))
    for nb, md in ((32, 64), (32, 128), (64, 256)):
        want = O.relative_position_bucket(np.arange(-300, 301), nb, md)
        got = np.array([built_lib.avexhip_rel_bucket(int(d), nb, md) for d in range(-300, 301)], dtype=np.int64)
        assert np.array_equal(got, np.asarray(want, dtype=np.int64)), (nb, md)


def test_graft_entry_build():
    """The driver's build entry: compiles/links if stale, loads the library, checks the ABI number against the header."""
    import importlib
    g = importlib.import_module("__graft_entry__")
    g.build()


def test_oracle_is_test_infrastructure_only():
    """Nothing the product ships or measures with may import `oracle/`: the package, the native sources, and scripts/ (measurement
    helpers) stay clear of it; bench.py touches it in the cpu_baseline leg only, __graft_entry__ in build() (load check) and smoke()."""
    import ast
    import pathlib
    root = pathlib.Path(__file__).resolve().parent.parent

    def oracle_imports(path):
        tree = ast.parse(path.read_text())
        hits = []
        for node in ast.walk(tree):
            if isinstance(node, ast.ImportFrom) and node.module and node.module.split(".")[0] == "oracle":
                hits.append(node.lineno)
            elif isinstance(node, ast.Import) and any(a.name.split(".")[0] == "oracle" for a in node.names):
                hits.append(node.lineno)
        return hits

    for d in ("avex_amd", "scripts"):
        for f in sorted((root / d).rglob("*.py")):
            assert oracle_imports(f) == [], f"{f.relative_to(root)} imports the test oracle"
    src = (root / "bench.py").read_text()
    tree = ast.parse(src)
    owners = {}
    for fn in [n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef)]:
        for node in ast.walk(fn):
            if isinstance(node, ast.ImportFrom) and node.module and node.module.split(".")[0] == "oracle":
                owners[fn.name] = node.lineno
    # the cpu_baseline leg is two functions: the parent that times, and the worker body its child processes run
    assert set(owners) <= {"cpu_baseline", "cpu_baseline_worker"} and owners, owners
