"""Variable-length inference: a new clip length costs a forward one small kernel on its own stream -- the relative position bias table
[H, 2T' - 1] (compute_bias, backbone.py:475-492) is built from the resident bucket table into an arena that lives as long as the handle,
or, when the arena is full, into the caller's workspace.  No table is ever freed under a recorded graph (`-m gpu`)."""
import numpy as np
import pytest
import torch

from avex_amd import synth

pytestmark = pytest.mark.gpu

# 40 distinct token counts: 16 kHz clips from 0.5 s to 20 s in uneven steps (48 .. 992 tokens)
LENGTHS = [8000 + 7920 * i for i in range(40)]


def _enc(monkeypatch, arena_mb=None):
    from avex_amd import kernels as K
    if arena_mb is None:
        monkeypatch.delenv("AVEX_AMD_BIAS_ARENA_MB", raising=False)
    else:
        monkeypatch.setenv("AVEX_AMD_BIAS_ARENA_MB", str(arena_mb))
    return K.BeatsEncoder(synth.BEATS_BASE_CFG, synth.beats_state_dict(synth.BEATS_BASE_CFG, seed=0), operand_dtype="f16")


def test_forty_lengths_back_to_back_match_workspace_built_tables(built_lib, monkeypatch):
    """The same 40 lengths through a handle with the default arena, one whose arena holds about half of the tables (the rest fall back to
    the workspace) and one without an arena (every forward builds its table in the workspace): bit-identical embeddings, and a length
    seen again gives what it gave the first time."""
    encs = [_enc(monkeypatch), _enc(monkeypatch, 1), _enc(monkeypatch, 0)]
    tokens = set()
    first = {}
    for n in LENGTHS + LENGTHS[:3]:
        x = torch.from_numpy(synth.noise_clips(2, n, seed=n % 97)).cuda()
        outs = [e.forward(x, want_features=False, want_pooled=True) for e in encs]
        tokens.add(outs[0]["tokens"])
        for o in outs[1:]:
            assert torch.equal(o["pooled"], outs[0]["pooled"]), n
        if n in first:
            assert torch.equal(outs[0]["pooled"], first[n]), n
        first[n] = outs[0]["pooled"].clone()
    assert len(tokens) == 40
    for e in encs:
        e.close()


def test_graph_survives_many_other_lengths(built_lib, monkeypatch):
    """A recorded forward keeps replaying the eager result after 40 eager forwards of other lengths on the same handle (round 3 evicted
    and freed the least recently used table after 16 lengths -- possibly the one a graph pointed to)."""
    enc = _enc(monkeypatch, 1)
    B, T = 2, 40000
    g = enc.capture(B, T, want_features=False, want_pooled=True)
    x = torch.from_numpy(synth.noise_clips(B, T, seed=5)).cuda()
    ref = enc.forward(x, want_features=False, want_pooled=True)["pooled"].clone()
    for n in LENGTHS:
        enc.forward(torch.from_numpy(synth.noise_clips(1, n, seed=1)).cuda(), want_features=False, want_pooled=True)
    g.wav.copy_(x)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(g.pooled, ref)
    g.close()
    enc.close()


def test_table_built_on_another_stream_is_ordered(built_lib, monkeypatch):
    """First use of a length on a side stream, second use immediately on the current stream: the second forward waits for the table."""
    enc = _enc(monkeypatch)
    x = torch.from_numpy(synth.noise_clips(2, 52000, seed=3)).cuda()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        a = enc.forward(x, want_features=False, want_pooled=True)["pooled"]
    # a second workspace so that the two forwards do not share scratch while they overlap
    enc2_ws = enc._ws
    enc._ws = None
    b = enc.forward(x, want_features=False, want_pooled=True)["pooled"]
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    del enc2_ws
    enc.close()
