"""Host side of the device probe heads: constructor surface, state_dict key names identical to the reference's probe
classes (from the committed goldens), loud failures without a GPU / for training."""
import numpy as np
import pytest
import torch

from avex_amd import probes as P


def _keys(g, prefix):
    return {k[len(prefix) + 4:] for k in g.files if k.startswith(prefix + ".sd.")}


def test_state_dict_keys_match_reference(golden_dir):
    g = np.load(f"{golden_dir}/probes.npz")
    lin = P.LinearProbe(None, [], 37, device="cpu", feature_mode=True, input_dim=[(768,)] * 4)
    assert set(lin.state_dict()) == _keys(g, "lin")
    lin1 = P.LinearProbe(None, [], 37, device="cpu", feature_mode=True, input_dim=768)
    assert set(lin1.state_dict()) == _keys(g, "lin1") and not hasattr(lin1, "layer_weights")
    mlp = P.MLPProbe(None, [], 37, device="cpu", feature_mode=True, input_dim=[(768,)] * 4, hidden_dims=[256, 64], dropout_rate=0.1)
    assert set(mlp.state_dict()) == _keys(g, "mlp_relu")
    assert {k: tuple(v.shape) for k, v in mlp.state_dict().items()} == {k[12:]: g[k].shape for k in g.files if k.startswith("mlp_relu.sd.")}
    mlp0 = P.MLPProbe(None, [], 3, device="cpu", feature_mode=True, input_dim=16, hidden_dims=[8], dropout_rate=0.0)
    assert set(mlp0.state_dict()) == {"mlp.0.weight", "mlp.0.bias", "mlp.2.weight", "mlp.2.bias"}      # no Dropout slot
    att = P.AttentionProbe(None, [], 37, device="cpu", feature_mode=True, input_dim=[(24, 128)] * 3, aggregation="none", num_heads=4,
                           num_layers=2, max_sequence_length=64, use_positional_encoding=True)
    assert set(att.state_dict()) == _keys(g, "att")
    assert torch.allclose(att.pos_encoding, torch.from_numpy(g["att.sd.pos_encoding"]), atol=1e-6)
    att.load_state_dict({k: torch.from_numpy(g["att.sd." + k]) for k in _keys(g, "att")})
    assert "Number of layers: 3" in att.get_learned_weights_table()


def test_sequence_probe_keys_match_reference(golden_dir):
    g = np.load(f"{golden_dir}/probes_seq.npz")
    tr = P.TransformerProbe(None, [], 19, device="cpu", feature_mode=True, input_dim=[(40, 128)] * 3, aggregation="none", num_heads=4,
                            attention_dim=192, num_layers=2, max_sequence_length=64, use_positional_encoding=True)
    assert {k: tuple(v.shape) for k, v in tr.state_dict().items()} == {k[6:]: g[k].shape for k in g.files if k.startswith("tr.sd.")}
    ls = P.LSTMProbe(None, [], 19, device="cpu", feature_mode=True, input_dim=[(40, 128)] * 3, aggregation="none", lstm_hidden_size=64,
                     num_layers=2, bidirectional=True, max_sequence_length=64, use_positional_encoding=True)
    assert {k: tuple(v.shape) for k, v in ls.state_dict().items()} == {k[10:]: g[k].shape for k in g.files if k.startswith("bilstm.sd.")}
    # the shipped evaluation configs: lstm_hidden_size below max_sequence_length / 4 = 300 units (lstm_probe.py:60)
    big = P.LSTMProbe(None, [], 3, device="cpu", feature_mode=True, input_dim=(40, 128), aggregation="none", lstm_hidden_size=128, max_sequence_length=1200)
    assert big.hidden == 300 and big.state_dict()["lstm.weight_hh_l0"].shape == (1200, 300)
    with pytest.raises(ValueError):          # more units than threads of a workgroup
        P.LSTMProbe(None, [], 3, device="cpu", feature_mode=True, input_dim=(40, 128), aggregation="none", lstm_hidden_size=2048)


def test_refusals():
    lin = P.LinearProbe(None, [], 5, device="cpu", feature_mode=True, input_dim=32)
    with pytest.raises(NotImplementedError):
        lin.train()
    assert lin.eval() is lin
    with pytest.raises(RuntimeError):                                         # no CPU fallback: the kernels need device tensors
        lin(torch.zeros(2, 32))
    with pytest.raises(ValueError):
        P.LinearProbe(None, [], 5, device="cpu", feature_mode=True)
    with pytest.raises(NotImplementedError):
        P.LinearProbe(None, [], 5, device="cpu", feature_mode=True, input_dim=32, freeze_backbone=False)
    with pytest.raises(ValueError):
        P.MLPProbe(None, [], 5, device="cpu", feature_mode=True, input_dim=32, activation="swish")


def test_embedding_projectors_match_reference(golden_dir):
    """Taps of unequal width / length: which taps get an nn.Linear, to what width, under which state_dict keys -- as the reference's
    own probe classes decided for the same shapes (tests/golden/probes_proj.npz, made by make_probe_goldens.py)."""
    g = np.load(f"{golden_dir}/probes_proj.npz")
    lin = P.LinearProbe(None, [], 37, device="cpu", feature_mode=True, input_dim=[(768,), (768,), (512,)])
    want = {k[7:]: g[k].shape for k in g.files if k.startswith("lin.sd.")}
    assert {k: tuple(v.shape) for k, v in lin.state_dict().items()} == want
    assert lin.embedding_projectors[0] is None and lin.embedding_projectors[2].in_features == 512 and lin.inferred_dim == 768
    att = P.AttentionProbe(None, [], 37, device="cpu", feature_mode=True, input_dim=[(24, 128), (24, 128), (31, 96), (40, 128)],
                           aggregation="none", num_heads=4, attention_dim=128, num_layers=1, dropout_rate=0.0)
    want = {k[7:]: g[k].shape for k in g.files if k.startswith("att.sd.")}
    assert {k: tuple(v.shape) for k, v in att.state_dict().items()} == want
    # no majority sequence length -> the target is the longest (40), so the two 24-long taps get 128 -> 128 projectors too
    assert [m is None for m in att.embedding_projectors] == [False, False, False, True]
    att.load_state_dict({k[7:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("att.sd.")})


def test_probe_factory_mirrors_reference():
    """build_probe_from_config (models/probes/utils/factory.py:56-186): mode selection, errors, probe-specific fields, signature filtering;
    a plain mapping stands in for the reference's ProbeConfig (the live reference object is exercised by tests/test_binding_contract.py)."""
    cfg = dict(probe_type="MLP", target_layers=["last_layer"], aggregation="mean", input_processing="pooled", freeze_backbone=True,
               hidden_dims=[32, 8], dropout_rate=0.0, activation="gelu", num_heads=4, lstm_hidden_size=None, target_length=None)
    pr = P.build_probe_from_config(cfg, num_classes=7, device="cpu", input_dim=64)
    assert isinstance(pr, P.MLPProbe) and pr.feature_mode and set(pr.state_dict()) == {"mlp.0.weight", "mlp.0.bias", "mlp.2.weight", "mlp.2.bias",
                                                                                      "mlp.4.weight", "mlp.4.bias"}
    seq = dict(probe_type="lstm", target_layers=["last_layer"], aggregation="none", input_processing="sequence", lstm_hidden_size=64,
               num_layers=1, bidirectional=True)
    ls = P.build_probe_from_config(seq, num_classes=3, device="cpu", input_dim=(40, 128))
    assert isinstance(ls, P.LSTMProbe) and ls.bidirectional and ls.hidden == 64
    assert set(P.list_probe_classes()) == {"linear", "mlp", "attention", "lstm", "transformer"} and P.get_probe_class("Transformer") is P.TransformerProbe
    with pytest.raises(ValueError):
        P.build_probe_from_config(cfg, num_classes=7, device="cpu")                                   # neither base_model nor input_dim
    with pytest.raises(ValueError):
        P.build_probe_from_config(cfg, num_classes=7, device="cpu", base_model=object(), input_dim=64) # both
    with pytest.raises(ValueError):
        P.build_probe_from_config(dict(cfg, probe_type="svm"), num_classes=7, device="cpu", input_dim=64)
    with pytest.raises(ValueError):
        P.build_probe_from_config(dict(cfg, input_processing="sequence"), num_classes=7, device="cpu", input_dim=64)
