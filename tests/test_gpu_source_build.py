"""The shipped libavexhip.so is what its sources produce: on the GPU box itself, compile every source of avex_amd/csrc for gfx950 into a side
library (avex_amd/build.py with AVEX_AMD_LIB_SUFFIX), run one small forward of EVERY handle -- BEATs, EAT, EfficientNet-B0 (with its mel
frontend), AVES -- and the probe-head kernels (fp32 dense, MultiheadAttention core, LSTM layer pair, layer mix) through both libraries in
fresh processes and compare the outputs bit for bit.  (The library travels prebuilt and build() rebuilds only when stale, so nothing else on
the box compiles the product.)"""
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
res = dict(pooled=out["pooled"].float().cpu().numpy(), features=out["features"].float().cpu().numpy())
# EAT (csrc/encoders.cpp: token_embed_ln, the 513-token attention with its tail kernel), both residual streams
from avex_amd.eat_encoder import EatEncoder
ecfg = dict(synth.EAT_BASE_CFG, depth=2)
eat = EatEncoder(ecfg, synth.eat_state_dict(ecfg), operand_dtype="f16")
ew = torch.from_numpy(synth.noise_clips(3, 48000, seed=10)).cuda()
r = eat.forward(ew, hook_layers=[1], pooling="cls")
res.update(eat_features=r["features"].cpu().numpy(), eat_cls=r["pooled"].cpu().numpy(), eat_tap=r["hooks"][1].cpu().numpy(),
           eat_mean=eat.forward(ew, want_features=False, pooling="mean")["pooled"].cpu().numpy())
# EfficientNet-B0 (csrc/effnet.hip, effnet_handle.cpp) behind the mel frontend (csrc/melspec.hip)
from avex_amd.effnet_encoder import EfficientNetB0Encoder
eff = EfficientNetB0Encoder(synth.effnet_b0_state_dict())
plan = K.MelspecPlan(n_fft=800, hop_length=160, n_mels=128, normalize=True)
mel = plan(torch.from_numpy(synth.noise_clips(3, 64000, seed=11)).cuda())
r = eff.forward(mel, want_features=True, want_pooled=True)
res.update(mel=mel.float().cpu().numpy(), eff_features=r["features"].float().cpu().numpy(), eff_pooled=r["pooled"].cpu().numpy())
# AVES (csrc/wavconv.hip, strided-row GEMMs, posconv)
from avex_amd.aves_encoder import AvesEncoder
acfg = dict(synth.AVES_BASE_CFG, encoder_num_layers=2)
aves = AvesEncoder(acfg, synth.aves_state_dict(acfg))
r = aves.forward(torch.from_numpy(synth.noise_clips(3, 32000, seed=12)).cuda(), hook_layers=[1], want_features=True, want_pooled=True)
res.update(aves_features=r["features"].cpu().numpy(), aves_pooled=r["pooled"].cpu().numpy(), aves_tap=r["hooks"][1].cpu().numpy())
# probe-head kernels (csrc/probe.hip, lstm.hip, attention_hd.hip)
g = torch.Generator().manual_seed(3)
x = torch.randn(3, 50, 768, generator=g).cuda(); w = (torch.randn(96, 768, generator=g) * 0.03).cuda(); b = torch.randn(96, generator=g).cuda()
res["dense"] = K.dense_f32(x, w, b, act="gelu").cpu().numpy()
res["mha"] = K.mha_f32(torch.randn(3, 50, 3 * 96, generator=g).cuda(), 4).cpu().numpy()
res["att_hd"] = K.attention_hd(torch.randn(3 * 50, 3 * 8 * 96, generator=g).cuda().half(), 3, 50, 8, 96).float().cpu().numpy()
H = 64
xg, xr = torch.randn(3, 50, 4 * H, generator=g).cuda(), torch.randn(3, 50, 4 * H, generator=g).cuda()
wh, wr = (torch.randn(H, 4 * H, generator=g) * 0.1).cuda(), (torch.randn(H, 4 * H, generator=g) * 0.1).cuda()
lo = torch.empty(3, 50, 2 * H, device="cuda")
K.lstm_layer_pair(xg, wh, xr, wr, lo)
res["lstm"] = lo.cpu().numpy()
res["mix"] = K.layer_mix([x[0], x[1], x[2]], torch.randn(3, generator=g).cuda()).cpu().numpy()
torch.cuda.synchronize()
np.savez({out!r}, lib=np.array(_capi.LIB_PATH), **res)
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
        keys = [k for k in outs["shipped"].files if k != "lib"]
        assert len(keys) >= 17 and sorted(keys) == sorted(k for k in outs["side"].files if k != "lib")
        for k in keys:
            a, b = outs["shipped"][k], outs["side"][k]
            assert np.isfinite(a).all() and a.shape == b.shape
            assert np.array_equal(a, b), f"{k}: the library built here differs from the shipped one (max |diff| {np.abs(a - b).max():.3g})"
    finally:
        import shutil
        if os.path.exists(side):
            os.remove(side)
        shutil.rmtree(objdir, ignore_errors=True)
