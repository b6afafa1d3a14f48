"""A forward recorded as a hipGraph (`avexhip_beats_graph_*`, `BeatsEncoder.capture`) replays the eager forward bit for bit (`-m gpu`)."""
import numpy as np
import pytest
import torch

from avex_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def enc(built_lib):
    from avex_amd import kernels as K
    e = K.BeatsEncoder(synth.BEATS_BASE_CFG, synth.beats_state_dict(synth.BEATS_BASE_CFG, seed=0), operand_dtype="f16")
    yield e
    e.close()


def test_graph_replay_equals_eager_forward(enc):
    B, T = 2, 32000
    g = enc.capture(B, T, hook_layers=[0, 5, 12], want_features=True, want_pooled=True)
    assert g.nodes >= 80          # 12 layers x 7 launches + frontend: the whole forward is in the graph, not a stub
    for seed in (3, 4, 5):        # new data through the SAME graph: the input is read at replay time, not at capture time
        x = torch.from_numpy(synth.noise_clips(B, T, seed=seed)).cuda()
        ref = enc.forward(x, hook_layers=[0, 5, 12], want_features=True, want_pooled=True)
        g.wav.copy_(x)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(g.features, ref["features"]) and torch.equal(g.pooled, ref["pooled"])
        for i in (0, 5, 12):
            assert torch.equal(g.hooks[i], ref["hooks"][i])
    g.close()


def test_graph_with_padding_mask_and_pooled_hooks(enc):
    B, T = 3, 48000
    g = enc.capture(B, T, hook_layers=[12], hook_pooled=True, want_features=False, want_pooled=True, with_frame_pad=True)
    x = torch.from_numpy(synth.noise_clips(B, T, seed=8)).cuda()
    pad = torch.zeros((B, g.tokens), dtype=torch.uint8)
    pad[1, g.tokens // 2:] = 1
    for p in (pad, torch.zeros_like(pad)):       # the mask is a static input too
        ref = enc.forward(x, hook_layers=[12], hook_pooled=True, want_features=False, want_pooled=True, frame_pad=p)
        g.wav.copy_(x)
        g.frame_pad.copy_(p)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(g.pooled, ref["pooled"]) and torch.equal(g.hooks[12], ref["hooks"][12])
    g.close()


def test_graph_many_replays_back_to_back(enc):
    """Replays queue on the stream like kernels: 50 in a row without a host synchronisation, last result = eager result."""
    B, T = 1, 160000
    g = enc.capture(B, T, want_features=False, want_pooled=True)
    x = torch.from_numpy(synth.noise_clips(B, T, seed=0)).cuda()
    g.wav.copy_(x)
    for _ in range(50):
        g.replay()
    torch.cuda.synchronize()
    ref = enc.forward(x, want_features=False, want_pooled=True)["pooled"]
    assert torch.equal(g.pooled, ref)
    gold = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "base_api.npz"))["b1.pooled"]
    err = float(np.linalg.norm(g.pooled.cpu().numpy() - gold) / np.linalg.norm(gold))
    assert err < 1e-3
    g.close()


def test_graph_capture_refuses_profiling_mode(enc):
    from avex_amd._capi import AvexHipError
    enc.set_profiling(True)
    try:
        with pytest.raises(AvexHipError):
            enc.capture(1, 16000)
    finally:
        enc.set_profiling(False)
    enc.forward(torch.zeros(1, 16000).cuda())       # the handle is usable afterwards
