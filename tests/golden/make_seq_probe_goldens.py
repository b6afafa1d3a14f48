"""Golden vectors for the two remaining sequence probes (development container only).

    python tests/golden/make_seq_probe_goldens.py      ->  tests/golden/probes_seq.npz

The reference's TransformerProbe and LSTMProbe (avex/models/probes/{transformer,lstm}_probe.py) are built in feature_mode (no backbone),
given seeded weights, and run on seeded sequence embeddings, with and without a key padding mask; inputs, weights and logits are
stored.  Data only.
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
from avex.models.probes.lstm_probe import LSTMProbe  # noqa: E402
from avex.models.probes.transformer_probe import TransformerProbe  # noqa: E402


def seed_params(mod, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for _, p in sorted(mod.named_parameters()):
            p.copy_(torch.randn(p.shape, generator=g) * (0.5 if p.dim() == 1 else p.shape[-1] ** -0.5))


def main():
    out = {}
    g = torch.Generator().manual_seed(77)
    C, B, T, D = 19, 3, 40, 128
    seqs = [torch.randn(B, T, D, generator=g) for _ in range(3)]
    out["seqs"] = torch.stack(seqs).numpy()
    pad = torch.zeros(B, T, dtype=torch.bool)
    pad[1, 25:] = True
    out["pad"] = pad.numpy()

    tr = TransformerProbe(None, [], C, device="cpu", feature_mode=True, input_dim=[(T, D)] * 3, aggregation="none", num_heads=4, attention_dim=192,
                          num_layers=2, dropout_rate=0.0, max_sequence_length=64, use_positional_encoding=True)
    seed_params(tr, 11)
    tr.eval()
    with torch.no_grad():
        out["tr.logits"] = tr({f"l{i}": e for i, e in enumerate(seqs)}).numpy()
        out["tr.logits_pad"] = tr({f"l{i}": e for i, e in enumerate(seqs)}, padding_mask=pad).numpy()
    for k, v in tr.state_dict().items():
        out[f"tr.sd.{k}"] = v.numpy()
    # one tap, no positional encoding, the default width of the feed-forward
    tr1 = TransformerProbe(None, [], C, device="cpu", feature_mode=True, input_dim=(T, D), aggregation="none", num_heads=12, attention_dim=96,
                           num_layers=1, dropout_rate=0.1)
    seed_params(tr1, 12)
    tr1.eval()
    with torch.no_grad():
        out["tr1.logits"] = tr1(seqs[0]).numpy()
    out["tr1.num_heads"] = np.int64(tr1.num_heads)             # 12 does not divide 128: the probe lowers it (transformer_probe.py:58-63)
    for k, v in tr1.state_dict().items():
        out[f"tr1.sd.{k}"] = v.numpy()

    for name, kw in (("lstm", dict(lstm_hidden_size=64, num_layers=2, bidirectional=False)),
                     ("bilstm", dict(lstm_hidden_size=64, num_layers=2, bidirectional=True, max_sequence_length=64, use_positional_encoding=True))):
        ls = LSTMProbe(None, [], C, device="cpu", feature_mode=True, input_dim=[(T, D)] * 3, aggregation="none", dropout_rate=0.0, **kw)
        seed_params(ls, 13)
        ls.eval()
        with torch.no_grad():
            out[f"{name}.logits"] = ls({f"l{i}": e for i, e in enumerate(seqs)}).numpy()
        for k, v in ls.state_dict().items():
            out[f"{name}.sd.{k}"] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "probes_seq.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
