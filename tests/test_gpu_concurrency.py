"""Kernels must be correct when they share the device with other kernels (two streams, co-resident workgroups).

Round 1 found the fbank kernel computing wrong values when it overlapped a GEMM on another stream (api.cpp kept every frontend
launch in front of the stream fork because of it).  Round 2 traced it to one instruction form: packed-fp32 arithmetic with
op_sel:[0,1] reads a wrong operand in lanes 32-63 while another wave on the CU issues MFMAs (profiles/r02a_pk_opsel_erratum.txt);
hipcc's SLP vectoriser had produced it for the complex butterflies.  The library is now built without it and avex_amd/isa_lint.py
(tests/test_isa_lint.py) keeps it out.  These tests run each victim kernel on stream B while a small-tile MFMA GEMM (128 x 128
tiles, 64 KiB of LDS: its workgroups share CUs with the victim's) loops on stream A, and bit-compare with the serial result."""
import numpy as np
import pytest
import torch

from avex_amd import synth
from avex_amd import kernels as K

pytestmark = pytest.mark.gpu


def _aggressors():
    M = 32 * 496
    g = torch.Generator(device="cpu").manual_seed(0)
    x = torch.randn(M, 768, generator=g).cuda().half()
    w = (torch.randn(2304, 768, generator=g) * 0.05).cuda().half()
    return {
        "gemm_small_tile": lambda: K.gemm(x, w, out_f32=False, out_half=True, variant=3),
        "gemm_default": lambda: K.gemm(x, w, out_f32=False, out_half=True),
    }


@pytest.mark.parametrize("aggressor", ["gemm_small_tile", "gemm_default"])
def test_fbank_beside_gemm_is_bit_identical(built_lib, aggressor):
    wav = torch.from_numpy(synth.noise_clips(32, 160000, seed=0)).cuda()
    plan = K.FbankPlan(norm_mean=15.41663, norm_div=13.11164)
    ref = plan(wav).clone()
    torch.cuda.synchronize()
    fn = _aggressors()[aggressor]
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    wrong = 0
    for it in range(40):                      # 40 rounds x 5 launches = 200 overlapped fbank launches
        torch.cuda.synchronize()
        with torch.cuda.stream(sa):
            for _ in range(30):
                fn()
        with torch.cuda.stream(sb):
            outs = [plan(wav) for _ in range(5)]
        torch.cuda.synchronize()
        wrong += sum(int((o != ref).sum()) for o in outs)
    assert wrong == 0, f"{wrong} wrong fbank elements beside {aggressor}"


def test_encoder_two_stream_lanes_match_one_stream(built_lib, monkeypatch):
    """The handle's optional two-lane chunk overlap (AVEX_AMD_STREAMS=2) must be bit-identical to the single-stream result."""
    sd = synth.beats_state_dict(synth.BEATS_BASE_CFG, seed=0)
    wav = torch.from_numpy(synth.noise_clips(24, 160000, seed=3)).cuda()
    monkeypatch.setenv("AVEX_AMD_STREAMS", "1")
    e1 = K.BeatsEncoder(synth.BEATS_BASE_CFG, sd, operand_dtype="f16", max_chunk_clips=6)
    p1 = e1.forward(wav, want_pooled=True)["pooled"].clone()
    e1.close()
    monkeypatch.setenv("AVEX_AMD_STREAMS", "2")
    e2 = K.BeatsEncoder(synth.BEATS_BASE_CFG, sd, operand_dtype="f16", max_chunk_clips=6)
    for _ in range(5):
        p2 = e2.forward(wav, want_pooled=True)["pooled"]
        torch.cuda.synchronize()
        assert torch.equal(p1, p2)
    e2.close()


def _victims():
    wav = torch.from_numpy(synth.noise_clips(16, 160000, seed=1)).cuda()
    eat = K.FbankPlan(input_scale=1.0, window=K.hann_window(400), norm_mean=-4.268, norm_div=2 * 4.569)
    mel = K.MelspecPlan(n_fft=800, hop_length=160, win_length=800, n_mels=128)
    g = torch.Generator(device="cpu").manual_seed(2)
    cw = (torch.randn(512, 10, generator=g) * 0.3).cuda()
    gw, gb = torch.rand(512, generator=g).cuda() + 0.5, torch.randn(512, generator=g).cuda() * 0.1
    x32 = torch.randn(16 * 496, 768, generator=g).cuda()
    lw, lb = torch.rand(768, generator=g).cuda(), torch.rand(768, generator=g).cuda()
    fp = 31999 // 8 * 8 + 8
    return {
        "eat_fbank_padded": lambda: eat.padded(wav, 1024, remove_clip_mean=True),
        "melspec_800": lambda: mel(wav),
        "wavconv0": lambda: K.wavconv0(wav, cw, gw, gb, fp)[: 16 * fp],
        "layernorm": lambda: K.layernorm(x32, lw, lb)[0],
    }


@pytest.mark.parametrize("victim", ["eat_fbank_padded", "melspec_800", "wavconv0", "layernorm"])
def test_other_frontends_beside_gemm_are_bit_identical(built_lib, victim):
    fn = _victims()[victim]
    ref = fn().clone()
    torch.cuda.synchronize()
    ag = _aggressors()["gemm_small_tile"]
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    wrong = 0
    for it in range(10):
        torch.cuda.synchronize()
        with torch.cuda.stream(sa):
            for _ in range(30):
                ag()
        with torch.cuda.stream(sb):
            outs = [fn() for _ in range(4)]
        torch.cuda.synchronize()
        wrong += sum(int((o != ref).sum()) for o in outs)
    assert wrong == 0, f"{wrong} wrong {victim} elements beside the small-tile GEMM"
