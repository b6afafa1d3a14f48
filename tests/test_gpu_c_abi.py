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


def test_c_abi_encoder_handle_consumer(built_lib, tmp_path):
    """avexhip_beats_create / _workspace_bytes / _forward / _destroy from a C++ program (tests/c_abi/beats_consumer.cpp): a 2-layer
    BEATs-width checkpoint and two clips go in through a flat file; features, pooled embedding and hook taps 0 and L must be
    bit-identical to the Python class on the same library and within tolerance of the CPU oracle."""
    import ctypes as C
    import struct
    import numpy as np
    import torch
    from avex_amd import synth, kernels as K
    from oracle import beats_oracle as O
    cfg = dict(synth.BEATS_BASE_CFG, encoder_layers=2)
    sd = synth.beats_state_dict(cfg, seed=2)
    x = synth.noise_clips(2, 32000, seed=77)
    ccfg = K.make_beats_config(cfg, "f16", 0, "half")
    blob = bytearray(struct.pack("<i", len(sd)))
    for k, v in sd.items():
        a = np.ascontiguousarray(v, np.float32)
        blob += struct.pack("<i", len(k)) + k.encode() + struct.pack("<q", a.size) + a.tobytes()
    blob += struct.pack("<iq", 2, 32000) + x.tobytes() + bytes(ccfg)
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    fin.write_bytes(bytes(blob))
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = tmp_path / "beats_consumer"
    lib = os.path.join(ROOT, "avex_amd", "lib")
    r = subprocess.run([hipcc, "-O1", "--offload-arch=gfx950", os.path.join(ROOT, "tests", "c_abi", "beats_consumer.cpp"), "-I", os.path.join(ROOT, "include"),
                        "-L", lib, "-lavexhip", f"-Wl,-rpath,{lib}", "-o", str(exe)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([str(exe), str(fin), str(fout)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "BEATS CONSUMER OK" in r.stdout, r.stdout + r.stderr
    raw = fout.read_bytes()
    B, Tt, E = struct.unpack_from("<3i", raw, 0)
    assert (B, Tt, E) == (2, 96, 768)
    arr = np.frombuffer(raw, np.float32, offset=12)
    n = B * Tt * E
    feat, pool, h0, hl = arr[:n].reshape(B, Tt, E), arr[n:n + B * E].reshape(B, E), arr[n + B * E:2 * n + B * E].reshape(B, Tt, E), arr[2 * n + B * E:].reshape(B, Tt, E)
    enc = K.BeatsEncoder(cfg, sd, operand_dtype="f16")
    py = enc.forward(torch.from_numpy(x).cuda(), hook_layers=[0, 2], want_features=True, want_pooled=True)
    assert np.array_equal(py["features"].cpu().numpy(), feat) and np.array_equal(py["pooled"].cpu().numpy(), pool)
    assert np.array_equal(py["hooks"][0].cpu().numpy(), h0) and np.array_equal(py["hooks"][2].cpu().numpy(), hl)
    f_ref, taps = O.beats_forward(x, sd, cfg)
    rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
    assert rel(pool, O.pooled(f_ref)) < 1e-3
    assert rel(h0.mean(1), taps["backbone.post_extract_proj"].mean(1)) < 1e-3
    assert rel(hl.mean(1), taps["backbone.encoder.layers.1.fc2"].mean(1)) < 1e-3


def test_c_abi_eat_handle_consumer(built_lib, tmp_path):
    """avexhip_eat_create / _workspace_bytes / _forward / _overflow_count / _destroy from a C++ program (tests/c_abi/eat_consumer.cpp):
    a 2-block EAT-width checkpoint and two 3 s clips go in through a flat file; features, class-token pooling and the last block's
    attn.proj tap must be bit-identical to the Python wrapper on the same library and within tolerance of the CPU oracle
    (SURVEY.md section 8 a15 / config C3's model; eat_hf.py:241-289)."""
    import ctypes as C
    import struct
    import numpy as np
    import torch
    from avex_amd import _capi, synth
    from avex_amd.eat_encoder import EatEncoder
    from oracle import eat_oracle as EO
    cfg = dict(synth.EAT_BASE_CFG, depth=2)
    sd = synth.eat_state_dict(cfg)
    x = synth.noise_clips(2, 48000, seed=78)
    c = _capi.EatConfig()
    c.embed_dim, c.num_heads, c.depth, c.ffn_dim, c.patch_size, c.target_length, c.n_mels = 768, 12, 2, 3072, 16, 1024, 128
    c.norm_eps, c.norm_mean, c.norm_std = 1e-6, -4.268, 4.569
    c.operand_dtype, c.max_chunk_clips, c.residual_dtype = _capi.F16, 0, 1
    blob = bytearray(struct.pack("<i", len(sd)))
    for k, v in sd.items():
        a = np.ascontiguousarray(v, np.float32)
        blob += struct.pack("<i", len(k)) + k.encode() + struct.pack("<q", a.size) + a.tobytes()       # keys with the reference's "backbone.model." prefix
    blob += struct.pack("<iq", 2, 48000) + x.tobytes() + bytes(c)
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    fin.write_bytes(bytes(blob))
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = tmp_path / "eat_consumer"
    lib = os.path.join(ROOT, "avex_amd", "lib")
    r = subprocess.run([hipcc, "-O1", "--offload-arch=gfx950", os.path.join(ROOT, "tests", "c_abi", "eat_consumer.cpp"), "-I", os.path.join(ROOT, "include"),
                        "-L", lib, "-lavexhip", f"-Wl,-rpath,{lib}", "-o", str(exe)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([str(exe), str(fin), str(fout)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "EAT CONSUMER OK" in r.stdout, r.stdout + r.stderr
    raw = fout.read_bytes()
    B, Tt, E = struct.unpack_from("<3i", raw, 0)
    assert (B, Tt, E) == (2, 513, 768)
    arr = np.frombuffer(raw, np.float32, offset=12)
    n = B * Tt * E
    feat, pool, hook = arr[:n].reshape(B, Tt, E), arr[n:n + B * E].reshape(B, E), arr[n + B * E:].reshape(B, Tt, E)
    enc = EatEncoder(cfg, sd, operand_dtype="f16", residual="half")      # the consumer's residual_dtype = 1
    py = enc.forward(torch.from_numpy(x).cuda(), hook_layers=[1], want_features=True, pooling="cls")
    assert np.array_equal(py["features"].cpu().numpy(), feat) and np.array_equal(py["pooled"].cpu().numpy(), pool)
    assert np.array_equal(py["hooks"][1].cpu().numpy(), hook)
    assert np.array_equal(pool, feat[:, 0])                                           # class-token pooling = token 0 of the features
    ref, taps = EO.eat_forward(x, sd, cfg)
    rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
    assert rel(feat.mean(1), ref.mean(1)) < 1e-3
    assert rel(hook.mean(1), taps["backbone.model.blocks.1.attn.proj"].mean(1)) < 2e-3
