"""Device probe heads against the reference's probe classes (committed goldens) and the oracle."""
import numpy as np
import pytest
import torch

from _util import rel_l2
from avex_amd import synth
from oracle import probe_oracle as PO

pytestmark = pytest.mark.gpu


def _sd(g, prefix, dev="cuda"):
    return {k[len(prefix) + 4:]: torch.from_numpy(g[k]).to(dev) for k in g.files if k.startswith(prefix + ".sd.")}


def test_layer_mix_bit_exact(built_lib):
    from avex_amd import kernels as K
    g = torch.Generator().manual_seed(0)
    taps = [torch.randn(7, 333, generator=g).cuda() for _ in range(5)]
    out = K.layer_mix(taps)                                       # no weights: ones, sequential fp32 adds
    ref = torch.zeros_like(taps[0])
    for t in taps:
        ref = ref + 1.0 * t
    assert torch.equal(out, ref)
    lw = torch.randn(5, generator=g).cuda()
    w = torch.softmax(lw, 0)
    ref = torch.zeros_like(taps[0])
    for t, wl in zip(taps, w):
        ref = ref + wl * t
    assert rel_l2(K.layer_mix(taps, lw).cpu().numpy(), ref.cpu().numpy()) < 3e-7
    with pytest.raises(ValueError):
        K.layer_mix([taps[0], taps[1][:, :10]])


@pytest.mark.parametrize("M,N,K_", [(5, 37, 768), (1, 1, 1), (130, 129, 19), (300, 2304, 768), (257, 64, 4099)])
@pytest.mark.parametrize("act", [None, "relu", "gelu", "tanh"])
def test_dense_f32(built_lib, M, N, K_, act):
    from avex_amd import kernels as K
    g = torch.Generator().manual_seed(M * 7 + N)
    x = torch.randn(M, K_, generator=g)
    w = torch.randn(N, K_, generator=g) * K_ ** -0.5
    b = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g)
    y = x.double() @ w.double().T + b.double()
    y = {None: y, "relu": torch.relu(y), "gelu": torch.nn.functional.gelu(y), "tanh": torch.tanh(y)}[act] + r.double()
    out = K.dense_f32(x.cuda(), w.cuda(), b.cuda(), act=act, resid=r.cuda())
    assert rel_l2(out.cpu().numpy(), y.numpy()) < 2e-6
    out = K.dense_f32(x.cuda(), w.cuda())                          # no bias, no residual
    assert rel_l2(out.cpu().numpy(), (x.double() @ w.double().T).numpy()) < 2e-6


@pytest.mark.parametrize("B,T,E,H", [(2, 24, 128, 4), (3, 496, 768, 8), (1, 17, 96, 2), (2, 100, 64, 16), (2, 200, 512, 8), (1, 130, 256, 2), (1, 129, 96, 1)])
def test_mha_f32(built_lib, B, T, E, H):
    from avex_amd import kernels as K
    g = torch.Generator().manual_seed(T)
    x = torch.randn(B, T, E, generator=g)
    mha = torch.nn.MultiheadAttention(E, H, batch_first=True).eval()
    pad = torch.zeros(B, T, dtype=torch.bool)
    pad[-1, T // 2:] = True
    for kp in (None, pad):
        with torch.no_grad():
            ref, _ = mha.double()(x.double(), x.double(), x.double(), key_padding_mask=kp)
        mha.float()
        qkv = K.dense_f32(x.cuda(), mha.in_proj_weight.detach().float().cuda(), mha.in_proj_bias.detach().float().cuda())
        att = K.mha_f32(qkv, H, None if kp is None else kp.cuda())
        out = K.dense_f32(att, mha.out_proj.weight.detach().float().cuda(), mha.out_proj.bias.detach().float().cuda())
        assert rel_l2(out.cpu().numpy(), ref.numpy()) < 3e-6


HALF_BAR = 3e-3      # the sequence probes' attention layers on f16 operands (avexhip_stack_*) against the fp32 reference


def _fp32_and_half(monkeypatch, call):
    """call() on the fp32 kernels (AVEX_AMD_PROBE_PRECISION=fp32: the reference's own precision) and on the default path."""
    monkeypatch.setenv("AVEX_AMD_PROBE_PRECISION", "fp32")
    exact = call().cpu().numpy()
    monkeypatch.delenv("AVEX_AMD_PROBE_PRECISION")
    return exact, call().cpu().numpy()


def test_probes_match_reference_goldens(built_lib, golden_dir, monkeypatch):
    from avex_amd import probes as P
    g = np.load(f"{golden_dir}/probes.npz")
    embs = [torch.from_numpy(e).cuda() for e in g["embs"]]
    lin = P.LinearProbe(None, [], 37, feature_mode=True, input_dim=[(768,)] * 4)
    lin.load_state_dict(_sd(g, "lin"))
    assert rel_l2(lin({f"l{i}": e for i, e in enumerate(embs)}).cpu().numpy(), g["lin.logits"]) < 3e-6
    lin1 = P.LinearProbe(None, [], 37, feature_mode=True, input_dim=768)
    lin1.load_state_dict(_sd(g, "lin1"))
    assert rel_l2(lin1(embs[0]).cpu().numpy(), g["lin1.logits"]) < 3e-6
    for act in ("relu", "gelu", "tanh"):
        mlp = P.MLPProbe(None, [], 37, feature_mode=True, input_dim=[(768,)] * 4, hidden_dims=[256, 64], dropout_rate=0.1, activation=act)
        mlp.load_state_dict(_sd(g, f"mlp_{act}"))
        assert rel_l2(mlp(embs).cpu().numpy(), g[f"mlp_{act}.logits"]) < 3e-6
    seqs = [torch.from_numpy(e).cuda() for e in g["seqs"]]
    att = P.AttentionProbe(None, [], 37, feature_mode=True, input_dim=[(24, 128)] * 3, aggregation="none", num_heads=4, num_layers=2,
                           dropout_rate=0.0, max_sequence_length=64, use_positional_encoding=True)
    att.load_state_dict(_sd(g, "att"))
    exact, half = _fp32_and_half(monkeypatch, lambda: att(seqs))
    assert rel_l2(exact, g["att.logits"]) < 1e-5
    assert att._stack is not None and rel_l2(half, g["att.logits"]) < HALF_BAR      # 4 heads of 32: the half-precision layer kernels


@pytest.fixture(scope="module")
def beats_model(built_lib, tmp_path_factory):
    import avex_amd
    from safetensors.numpy import save_file
    path = tmp_path_factory.mktemp("ckpt") / "synthetic_beats.safetensors"
    save_file({k: np.ascontiguousarray(v) for k, v in synth.beats_state_dict(synth.BEATS_BASE_CFG, seed=0).items()}, str(path))
    return avex_amd.load_model("esp_aves2_sl_beats_all", device="cuda", checkpoint_path=str(path), return_features_only=True).eval()


def test_probe_with_base_model(built_lib, beats_model, monkeypatch):
    """The online-probing shape: the BEATs model mirror as base_model, hooks on three layers, logits without leaving the device;
    checked against the probe oracle applied to the same device embeddings."""
    from avex_amd import probes as P
    layers = beats_model.register_hooks_for_layers(["backbone.encoder.layers.3.fc2", "backbone.encoder.layers.7.fc2", "last_layer"])
    try:
        wav = torch.from_numpy(synth.noise_clips(3, 32000, seed=3)).cuda()
        gen = torch.Generator().manual_seed(5)
        # aggregation "mean": one (B, 768 * 3) tensor -> Linear(2304, C)
        lin = P.LinearProbe(beats_model, layers, 11, aggregation="mean", target_length=32000)
        assert lin.inferred_dim == 768 * len(layers)
        sd = {"classifier.weight": torch.randn(11, lin.inferred_dim, generator=gen) * 0.03, "classifier.bias": torch.randn(11, generator=gen)}
        lin.load_state_dict({k: v.cuda() for k, v in sd.items()}, strict=False)   # base_model.* keys stay, as in the reference
        logits = lin({"raw_wav": wav, "padding_mask": None})
        emb = beats_model.extract_embeddings(wav, aggregation="mean")
        assert logits.is_cuda and logits.shape == (3, 11)
        assert rel_l2(logits.cpu().numpy(), PO.linear_probe(emb.cpu().numpy(), {k: v.numpy() for k, v in sd.items()})) < 3e-6
        # aggregation "none": a list of (B, T', 768) taps -> learned mix -> attention probe
        att = P.AttentionProbe(beats_model, layers, 11, aggregation="none", num_heads=8, num_layers=1, dropout_rate=0.0, target_length=32000)
        sd = {k: (torch.randn(v.shape, generator=gen) * (0.5 if v.dim() == 1 else v.shape[-1] ** -0.5)) for k, v in att.state_dict().items()
              if not k.startswith("base_model.")}
        att.load_state_dict({k: v.cuda() for k, v in sd.items()}, strict=False)
        exact, half = _fp32_and_half(monkeypatch, lambda: att(wav))
        taps = beats_model.extract_embeddings(wav, aggregation="none")
        assert isinstance(taps, list) and len(taps) == len(layers)
        ref = PO.attention_probe([t.cpu().numpy() for t in taps], {k: v.numpy() for k, v in sd.items()}, num_heads=8)
        assert rel_l2(exact, ref) < 1e-5
        assert att._stack is not None and rel_l2(half, ref) < HALF_BAR      # 8 heads of 96
    finally:
        beats_model.deregister_all_hooks()


def test_attention_probe_padding_mask(built_lib, golden_dir, monkeypatch):
    """key_padding_mask path of the attention probe (attention_probe.py:124-128): a mask of the sequence length masks keys, a mask of any
    other length is dropped; against the oracle on the golden weights."""
    from avex_amd import probes as P
    g = np.load(f"{golden_dir}/probes.npz")
    seqs = [torch.from_numpy(e).cuda() for e in g["seqs"]]
    att = P.AttentionProbe(None, [], 37, feature_mode=True, input_dim=[(24, 128)] * 3, aggregation="none", num_heads=4, num_layers=2,
                           dropout_rate=0.0, max_sequence_length=64, use_positional_encoding=True)
    att.load_state_dict(_sd(g, "att"))
    sd = {k[7:]: g[k] for k in g.files if k.startswith("att.sd.")}
    pad = np.zeros((3, 24), bool); pad[1, 15:] = True; pad[2, :3] = True
    ref = PO.attention_probe(list(g["seqs"]), sd, num_heads=4, key_pad=pad)
    out, half = _fp32_and_half(monkeypatch, lambda: att(seqs, padding_mask=torch.from_numpy(pad).cuda()))
    assert rel_l2(out, ref) < 1e-5 and rel_l2(half, ref) < HALF_BAR
    assert rel_l2(out, g["att.logits"]) > 1e-2 and rel_l2(half, g["att.logits"]) > 1e-2      # the mask changed something
    dropped, dropped_half = _fp32_and_half(monkeypatch, lambda: att(seqs, padding_mask=torch.zeros(3, 160000, dtype=torch.bool).cuda()))
    assert rel_l2(dropped, g["att.logits"]) < 1e-5 and rel_l2(dropped_half, g["att.logits"]) < HALF_BAR   # sample-level mask: ignored, as in the reference


def test_projectors_and_interpolation_match_reference(built_lib, golden_dir, monkeypatch):
    """Embedding projectors (base_probes.py:254-289,333-367) and the linear resampling of longer taps to the shortest sequence
    (:398-411) on the device against the reference's own probe classes (probes_proj.npz)."""
    from avex_amd import probes as P
    from avex_amd import kernels as K
    g = np.load(f"{golden_dir}/probes_proj.npz")
    lin = P.LinearProbe(None, [], 37, device="cuda", feature_mode=True, input_dim=[(768,), (768,), (512,)])
    lin.load_state_dict({k[7:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("lin.sd.")})
    out = lin({f"l{i}": torch.from_numpy(g[f"lin.emb{i}"]).cuda() for i in range(3)})
    assert rel_l2(out.cpu().numpy(), g["lin.logits"]) < 5e-6
    att = P.AttentionProbe(None, [], 37, device="cuda", feature_mode=True, input_dim=[(24, 128), (24, 128), (31, 96), (40, 128)],
                           aggregation="none", num_heads=4, attention_dim=128, num_layers=1, dropout_rate=0.0)
    att.load_state_dict({k[7:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("att.sd.")})
    seqs = [torch.from_numpy(g[f"att.seq{i}"]).cuda() for i in range(4)]
    comb = att._combine(seqs)
    assert comb.shape == (3, 24, 128) and rel_l2(comb.cpu().numpy(), g["att.combined"]) < 5e-6
    out, half = _fp32_and_half(monkeypatch, lambda: att({f"l{i}": s for i, s in enumerate(seqs)}))
    assert rel_l2(out, g["att.logits"]) < 2e-5 and rel_l2(half, g["att.logits"]) < HALF_BAR
    # the resampling kernel alone against torch's own interpolate, up- and down-sampling, odd ratios
    for tin, tout in ((40, 24), (31, 24), (24, 40), (7, 7), (100, 1), (3, 17)):
        x = torch.from_numpy(synth.normal(f"itp{tin}", (2, tin, 96), 1.0))
        ref = torch.nn.functional.interpolate(x.transpose(1, 2), size=tout, mode="linear", align_corners=False).transpose(1, 2)
        got = K.seq_interp_linear(x.cuda(), tout).cpu()
        # same formula, same float arithmetic; torch.s vectorised CPU kernel may contract w0 * x0 + w1 * x1 into an FMA
        assert got.shape == ref.shape and float((got - ref).abs().max()) < 2e-5 and rel_l2(got.numpy(), ref.numpy()) < 2e-6, (tin, tout)


def test_sequence_probes_match_reference_goldens(built_lib, golden_dir):
    """The two remaining sequence probes against the reference's own TransformerProbe / LSTMProbe (tests/golden/make_seq_probe_goldens.py):
    transformer with learned positions, with and without a key padding mask (the reference's encoder returns zeros at padded positions and
    averages over all of them), the head-count adjustment (12 heads on 128 channels -> 8), one- and two-directional two-layer LSTMs."""
    from avex_amd import probes as P
    g = np.load(f"{golden_dir}/probes_seq.npz")
    seqs = [torch.from_numpy(e).cuda() for e in g["seqs"]]
    tr = P.TransformerProbe(None, [], 19, feature_mode=True, input_dim=[(40, 128)] * 3, aggregation="none", num_heads=4, attention_dim=192,
                            num_layers=2, dropout_rate=0.0, max_sequence_length=64, use_positional_encoding=True)
    tr.load_state_dict(_sd(g, "tr"))
    assert rel_l2(tr(seqs).cpu().numpy(), g["tr.logits"]) < 1e-5
    pad = torch.from_numpy(g["pad"]).cuda()
    assert rel_l2(tr(seqs, padding_mask=pad).cpu().numpy(), g["tr.logits_pad"]) < 1e-5
    tr1 = P.TransformerProbe(None, [], 19, feature_mode=True, input_dim=(40, 128), aggregation="none", num_heads=12, attention_dim=96,
                             num_layers=1, dropout_rate=0.1)
    assert tr1.num_heads == int(g["tr1.num_heads"]) == 8
    tr1.load_state_dict(_sd(g, "tr1"))
    assert rel_l2(tr1(seqs[0]).cpu().numpy(), g["tr1.logits"]) < 1e-5
    ls = P.LSTMProbe(None, [], 19, feature_mode=True, input_dim=[(40, 128)] * 3, aggregation="none", lstm_hidden_size=64, num_layers=2,
                     bidirectional=False, dropout_rate=0.0)
    ls.load_state_dict(_sd(g, "lstm"))
    assert rel_l2(ls(seqs).cpu().numpy(), g["lstm.logits"]) < 1e-5
    bi = P.LSTMProbe(None, [], 19, feature_mode=True, input_dim=[(40, 128)] * 3, aggregation="none", lstm_hidden_size=64, num_layers=2,
                     bidirectional=True, dropout_rate=0.0, max_sequence_length=64, use_positional_encoding=True)
    bi.load_state_dict(_sd(g, "bilstm"))
    assert rel_l2(bi(seqs).cpu().numpy(), g["bilstm.logits"]) < 1e-5


@pytest.mark.parametrize("B,T,D,H", [(5, 496, 64, 128), (6, 64, 96, 256), (3, 48, 64, 320), (2, 24, 64, 576), (5, 40, 64, 300), (3, 30, 32, 100)])
def test_lstm_layer_long_sequence(built_lib, B, T, D, H):
    """The recurrence kernel at the probe's real length (496 steps) and in its three thread layouts (four / two / one thread per hidden unit:
    H <= 256, <= 512, <= 1024), batches that are no multiple of the clips per workgroup, both directions, against torch.nn.LSTM in fp64."""
    from avex_amd import kernels as K
    torch.manual_seed(3)
    lstm = torch.nn.LSTM(D, H, batch_first=True, bidirectional=True).double().eval()
    x = torch.randn(B, T, D)
    with torch.no_grad():
        ref, _ = lstm(x.double())
    out = torch.empty((B, T, 2 * H), dtype=torch.float32, device="cuda")
    parts = []
    for di, sfx in enumerate(("", "_reverse")):
        w_ih, w_hh = getattr(lstm, f"weight_ih_l0{sfx}").float().cuda(), getattr(lstm, f"weight_hh_l0{sfx}").float().cuda()
        b = (getattr(lstm, f"bias_ih_l0{sfx}") + getattr(lstm, f"bias_hh_l0{sfx}")).float().cuda()
        xg = K.dense_f32(x.cuda(), w_ih, b)
        K.lstm_layer(xg, w_hh.t().contiguous(), out, col=di * H, reverse=bool(di))
        parts.append((xg, w_hh.t().contiguous()))
    assert rel_l2(out.cpu().numpy(), ref.numpy()) < 2e-5
    both = torch.full_like(out, float("nan"))      # both directions in one launch: the same arithmetic
    K.lstm_layer_pair(parts[0][0], parts[0][1], parts[1][0], parts[1][1], both)
    assert torch.equal(both, out)


def test_lstm_probe_shipped_config_shape(built_lib):
    """The LSTM probe as the reference's evaluation configs build it (lstm_hidden_size 128, max_sequence_length 1200 -> 300 units, two
    layers) against the oracle (pinned to the reference's class at other sizes)."""
    from avex_amd import probes as P
    cfg = dict(probe_type="lstm", target_layers=["last_layer"], aggregation="none", input_processing="sequence", lstm_hidden_size=128, num_layers=2,
               bidirectional=False, max_sequence_length=1200, use_positional_encoding=False, dropout_rate=0.3)
    pr = P.build_probe_from_config(cfg, num_classes=11, device="cuda", input_dim=(60, 256))
    assert pr.hidden == 300
    gen = torch.Generator().manual_seed(8)
    sd = {k: (torch.randn(v.shape, generator=gen) * (0.3 if v.dim() == 1 else v.shape[-1] ** -0.5)) for k, v in pr.state_dict().items()}
    pr.load_state_dict(sd)
    x = torch.randn(4, 60, 256, generator=gen)
    ref = PO.lstm_probe(x.numpy(), {k: v.numpy() for k, v in sd.items()})
    assert rel_l2(pr(x.cuda()).cpu().numpy(), ref) < 2e-5


def test_transformer_probe_shipped_config_shape(built_lib, monkeypatch):
    """The transformer probe as the reference's evaluation configs build it behind BEATs (768 channels, 8 heads -> head_dim 96, feed-forward
    128, one layer, no positions): heads of 96 run on attention_hd.hip inside the layer-stack handle; both precisions against the oracle,
    with a key padding mask."""
    from avex_amd import probes as P
    cfg = dict(probe_type="transformer", target_layers=["last_layer"], aggregation="none", input_processing="sequence", num_heads=8, attention_dim=128,
               num_layers=1, max_sequence_length=1200, use_positional_encoding=False, dropout_rate=0.3)
    pr = P.build_probe_from_config(cfg, num_classes=11, device="cuda", input_dim=(100, 768))
    assert pr.num_heads == 8 and pr._stack_ok(768, 8, 128)
    gen = torch.Generator().manual_seed(18)
    sd = {k: (torch.randn(v.shape, generator=gen) * (0.3 if v.dim() == 1 else v.shape[-1] ** -0.5)) for k, v in pr.state_dict().items()}
    pr.load_state_dict(sd)
    x = torch.randn(3, 100, 768, generator=gen)
    pad = torch.zeros(3, 100, dtype=torch.bool); pad[0, 70:] = True
    npsd = {k: v.numpy() for k, v in sd.items()}
    exact, half = _fp32_and_half(monkeypatch, lambda: pr(x.cuda()))
    ref = PO.transformer_probe(x.numpy(), npsd, num_heads=8)
    assert rel_l2(exact, ref) < 1e-5 and pr._stack is not None and rel_l2(half, ref) < HALF_BAR
    exact, half = _fp32_and_half(monkeypatch, lambda: pr(x.cuda(), padding_mask=pad.cuda()))
    ref_pad = PO.transformer_probe(x.numpy(), npsd, num_heads=8, key_pad=pad.numpy())
    assert rel_l2(exact, ref_pad) < 1e-5 and rel_l2(half, ref_pad) < HALF_BAR and rel_l2(ref_pad, ref) > 1e-3


def test_transformer_probe_half_precision_stack(built_lib, monkeypatch):
    """At the probe's default shape (768 channels, 12 heads of 64, feed-forward 768) the encoder part runs on the half-precision layer kernels
    (avexhip_stack_*): against the same probe on the fp32 kernels and against the oracle, with and without a key padding mask; a
    reloaded state dict rebuilds the handle."""
    from avex_amd import probes as P
    torch.manual_seed(9)
    B, T, D, C = 5, 80, 768, 23
    x = torch.randn(B, T, D).cuda()
    pr = P.TransformerProbe(None, [], C, feature_mode=True, input_dim=(T, D), aggregation="none", num_layers=2, dropout_rate=0.0)
    gen = torch.Generator().manual_seed(4)
    sd = {k: (torch.randn(v.shape, generator=gen) * (0.3 if v.dim() == 1 else v.shape[-1] ** -0.5)) for k, v in pr.state_dict().items()}
    pr.load_state_dict(sd)
    pad = torch.zeros(B, T, dtype=torch.bool); pad[2, 50:] = True
    ref = PO.transformer_probe(x.cpu().numpy(), {k: v.numpy() for k, v in sd.items()}, num_heads=12)
    ref_pad = PO.transformer_probe(x.cpu().numpy(), {k: v.numpy() for k, v in sd.items()}, num_heads=12, key_pad=pad.numpy())
    fast, fast_pad = pr(x).cpu().numpy(), pr(x, padding_mask=pad.cuda()).cpu().numpy()
    assert pr._stack is not None                                    # the half-precision handle ran
    monkeypatch.setenv("AVEX_AMD_PROBE_PRECISION", "fp32")
    slow = pr(x).cpu().numpy()
    monkeypatch.delenv("AVEX_AMD_PROBE_PRECISION")
    assert rel_l2(slow, ref) < 1e-5
    assert rel_l2(fast, ref) < 2e-3 and rel_l2(fast_pad, ref_pad) < 2e-3
    assert rel_l2(fast_pad, fast) > 1e-3                            # the mask changed something
    sd2 = {k: v * 1.5 if k.endswith("linear2.weight") else v for k, v in sd.items()}
    pr.load_state_dict(sd2)
    ref2 = PO.transformer_probe(x.cpu().numpy(), {k: v.numpy() for k, v in sd2.items()}, num_heads=12)
    assert rel_l2(pr(x).cpu().numpy(), ref2) < 2e-3                 # new weights -> new handle
