"""Golden vectors for the probe heads, produced by the reference's own probe classes in THIS container.

    python tests/golden/make_probe_goldens.py      ->  tests/golden/probes.npz

The reference's LinearProbe / MLPProbe / AttentionProbe (avex/models/probes/*.py) are built in feature_mode (no backbone),
given seeded weights, and run on seeded embeddings; inputs, weights and logits are stored.  Data only: nothing of the
reference's source travels.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import numpy as np
import torch

from _ref_import import import_reference

import_reference()
from avex.models.probes.attention_probe import AttentionProbe  # noqa: E402
from avex.models.probes.linear_probe import LinearProbe  # noqa: E402
from avex.models.probes.mlp_probe import MLPProbe  # noqa: E402


def seed_params(mod, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for _, p in sorted(mod.named_parameters()):
            p.copy_(torch.randn(p.shape, generator=g) * (0.5 if p.dim() == 1 else p.shape[-1] ** -0.5))


def main():
    out = {}
    g = torch.Generator().manual_seed(1234)
    B, D, L, C = 5, 768, 4, 37
    embs = [torch.randn(B, D, generator=g) * (1 + 0.3 * i) for i in range(L)]
    out["embs"] = torch.stack(embs).numpy()

    lin = LinearProbe(None, [], C, device="cpu", feature_mode=True, input_dim=[(D,)] * L)
    seed_params(lin, 1)
    lin.eval()
    out["lin.logits"] = lin({f"l{i}": e for i, e in enumerate(embs)}).detach().numpy()
    for k, v in lin.state_dict().items():
        out[f"lin.sd.{k}"] = v.numpy()
    # uniform case: one tensor, no layer weights
    lin1 = LinearProbe(None, [], C, device="cpu", feature_mode=True, input_dim=D)
    seed_params(lin1, 2)
    out["lin1.logits"] = lin1(embs[0]).detach().numpy()
    for k, v in lin1.state_dict().items():
        out[f"lin1.sd.{k}"] = v.numpy()

    for act in ("relu", "gelu", "tanh"):
        mlp = MLPProbe(None, [], C, device="cpu", feature_mode=True, input_dim=[(D,)] * L, hidden_dims=[256, 64], dropout_rate=0.1, activation=act)
        seed_params(mlp, 3)
        mlp.eval()
        out[f"mlp_{act}.logits"] = mlp({f"l{i}": e for i, e in enumerate(embs)}).detach().numpy()
        for k, v in mlp.state_dict().items():
            out[f"mlp_{act}.sd.{k}"] = v.numpy()

    # sequence taps [B, T, D] for the attention probe (aggregation "none")
    T, D3 = 24, 128
    seqs = [torch.randn(3, T, D3, generator=g) for _ in range(3)]
    out["seqs"] = torch.stack(seqs).numpy()
    att = AttentionProbe(None, [], C, device="cpu", feature_mode=True, input_dim=[(T, D3)] * 3, aggregation="none", num_heads=4, attention_dim=D3,
                         num_layers=2, dropout_rate=0.0, max_sequence_length=64, use_positional_encoding=True)
    seed_params(att, 4)
    att.eval()
    out["att.logits"] = att({f"l{i}": e for i, e in enumerate(seqs)}).detach().numpy()
    for k, v in att.state_dict().items():
        out[f"att.sd.{k}"] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "probes.npz"), **out)
    print({k: v.shape for k, v in out.items()})

    # ---- taps of unequal width / length: embedding projectors (base_probes.py:254-289,333-367) and the interpolation to the
    # shortest sequence (:398-411) -> probes_proj.npz
    out = {}
    g = torch.Generator().manual_seed(4321)
    dims = [768, 768, 512]
    embs = [torch.randn(B, d, generator=g) for d in dims]
    for i, e in enumerate(embs):
        out[f"lin.emb{i}"] = e.numpy()
    lin = LinearProbe(None, [], C, device="cpu", feature_mode=True, input_dim=[(d,) for d in dims])
    seed_params(lin, 5)
    lin.eval()
    out["lin.logits"] = lin({f"l{i}": e for i, e in enumerate(embs)}).detach().numpy()
    for k, v in lin.state_dict().items():
        out[f"lin.sd.{k}"] = v.numpy()
    shapes = [(24, 128), (24, 128), (31, 96), (40, 128)]
    seqs = [torch.randn(3, t, d, generator=g) for t, d in shapes]
    for i, e in enumerate(seqs):
        out[f"att.seq{i}"] = e.numpy()
    att = AttentionProbe(None, [], C, device="cpu", feature_mode=True, input_dim=shapes, aggregation="none", num_heads=4, attention_dim=128,
                         num_layers=1, dropout_rate=0.0, use_positional_encoding=False)
    seed_params(att, 6)
    att.eval()
    out["att.logits"] = att({f"l{i}": e for i, e in enumerate(seqs)}).detach().numpy()
    out["att.combined"] = att._combine_or_reshape_embeddings(seqs).detach().numpy()
    for k, v in att.state_dict().items():
        out[f"att.sd.{k}"] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "probes_proj.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
