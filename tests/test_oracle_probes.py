"""The probe-head oracle against the reference's own probe classes (goldens from tests/golden/make_probe_goldens.py)."""
import numpy as np

from _util import rel_l2
from oracle import probe_oracle as P


def _sd(g, prefix):
    return {k[len(prefix) + 4:]: g[k] for k in g.files if k.startswith(prefix + ".sd.")}


def test_linear_probe_pinned(golden_dir):
    g = np.load(f"{golden_dir}/probes.npz")
    embs = list(g["embs"])
    assert rel_l2(P.linear_probe(embs, _sd(g, "lin")), g["lin.logits"]) < 2e-6
    assert rel_l2(P.linear_probe(embs[0], _sd(g, "lin1")), g["lin1.logits"]) < 2e-6
    # the mix itself is bit-exact fp32 when the softmax weights agree; check through a weightless (all-ones) case
    assert np.array_equal(P.layer_mix(embs), (((0 + embs[0]) + embs[1]) + embs[2]) + embs[3])


def test_mlp_probe_pinned(golden_dir):
    g = np.load(f"{golden_dir}/probes.npz")
    for act in ("relu", "gelu", "tanh"):
        assert rel_l2(P.mlp_probe(list(g["embs"]), _sd(g, f"mlp_{act}"), act), g[f"mlp_{act}.logits"]) < 2e-6


def test_attention_probe_pinned(golden_dir):
    g = np.load(f"{golden_dir}/probes.npz")
    assert rel_l2(P.attention_probe(list(g["seqs"]), _sd(g, "att"), num_heads=4), g["att.logits"]) < 5e-6


def test_transformer_probe_pinned(golden_dir):
    g = np.load(f"{golden_dir}/probes_seq.npz")
    seqs = list(g["seqs"])
    assert rel_l2(P.transformer_probe(seqs, _sd(g, "tr"), num_heads=4), g["tr.logits"]) < 5e-6
    # with a key padding mask the reference's encoder returns zeros at the padded positions and the mean includes them
    assert rel_l2(P.transformer_probe(seqs, _sd(g, "tr"), num_heads=4, key_pad=g["pad"]), g["tr.logits_pad"]) < 5e-6
    assert int(g["tr1.num_heads"]) == 8                                   # 12 heads do not divide 128: lowered to 8 (transformer_probe.py:58-63)
    assert rel_l2(P.transformer_probe(seqs[0], _sd(g, "tr1"), num_heads=8), g["tr1.logits"]) < 5e-6


def test_lstm_probe_pinned(golden_dir):
    g = np.load(f"{golden_dir}/probes_seq.npz")
    seqs = list(g["seqs"])
    assert rel_l2(P.lstm_probe(seqs, _sd(g, "lstm")), g["lstm.logits"]) < 5e-6
    assert rel_l2(P.lstm_probe(seqs, _sd(g, "bilstm")), g["bilstm.logits"]) < 5e-6
