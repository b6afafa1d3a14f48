"""The shipped libavexhip.so is what its sources produce: on the GPU box itself, compile every source of avex_amd/csrc for gfx950 into a side
library (avex_amd/build.py with AVEX_AMD_LIB_SUFFIX), run the same BEATs forward through both libraries in fresh processes and compare the
embeddings bit for bit.  (The library travels prebuilt and build() rebuilds only when stale, so nothing else on the box compiles the product.)"""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SUFFIX = "srccheck"

FORWARD = r"""
import sys, numpy as np, torch
sys.path.insert(0, {root!r})
from avex_amd import synth, kernels as K, _capi
cfg = dict(synth.BEATS_BASE_CFG, encoder_layers=3)
enc = K.BeatsEncoder(cfg, synth.beats_state_dict(cfg, seed=5), operand_dtype="f16", residual="half")
wav = torch.from_numpy(synth.noise_clips(12, 160000, seed=9)).cuda()          # 5 952 token rows: the streaming GEMM, the folded LayerNorms, attention variant 3
out = enc.forward(wav, want_features=True, want_pooled=True)
torch.cuda.synchronize()
np.savez({out!r}, pooled=out["pooled"].float().cpu().numpy(), features=out["features"].float().cpu().numpy(), lib=np.array(_capi.LIB_PATH))
"""


def _exports(path):
    r = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    return sorted(line.split()[-1] for line in r.stdout.splitlines() if " T " in line and "avexhip_" in line)


@pytest.mark.timeout(1800)
def test_library_built_from_source_here_reproduces_the_shipped_one(built_lib, tmp_path):
    env = dict(os.environ, AVEX_AMD_LIB_SUFFIX=SUFFIX)
    env.pop("AVEX_AMD_LIB", None)
    side = os.path.join(ROOT, "avex_amd", "lib", f"libavexhip_{SUFFIX}.so")
    objdir = os.path.join(ROOT, "avex_amd", "_build_" + SUFFIX)
    try:
        r = subprocess.run([sys.executable, "-c", f"import sys; sys.path.insert(0, {ROOT!r}); from avex_amd import build; print(build.build(force=True))"],
                           capture_output=True, text=True, timeout=1500, env=env)
        assert r.returncode == 0 and os.path.exists(side), (r.stdout + r.stderr)[-3000:]
        assert "compiling" in r.stdout, r.stdout[-500:]                       # it really compiled (17 sources), it did not find objects
        shipped = os.path.join(ROOT, "avex_amd", "lib", "libavexhip.so")
        assert _exports(side) == _exports(shipped)
        outs = {}
        for name, lib in (("shipped", shipped), ("side", side)):
            o = str(tmp_path / f"{name}.npz")
            e = dict(os.environ, AVEX_AMD_LIB=lib)
            e.pop("AVEX_AMD_LIB_SUFFIX", None)
            r = subprocess.run([sys.executable, "-c", FORWARD.format(root=ROOT, out=o)], capture_output=True, text=True, timeout=600, env=e)
            assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
            outs[name] = np.load(o)
            assert str(outs[name]["lib"]) == lib
        for k in ("pooled", "features"):
            a, b = outs["shipped"][k], outs["side"][k]
            assert np.isfinite(a).all() and a.shape == b.shape
            assert np.array_equal(a, b), f"{k}: the library built here differs from the shipped one (max |diff| {np.abs(a - b).max():.3g})"
    finally:
        import shutil
        if os.path.exists(side):
            os.remove(side)
        shutil.rmtree(objdir, ignore_errors=True)
