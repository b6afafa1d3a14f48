#!/usr/bin/env python3
"""The sequence probes at the size they meet behind BEATs: 256 clips x 496 tokens x 768 channels (one tap, feature mode), ms per batch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from avex_amd import probes as P

B, T, D, C = 256, 496, 768, 50
x = torch.randn(B, T, D, device="cuda")


def timeit(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


probes = {
    "attention (8 heads, 2 layers)": P.AttentionProbe(None, [], C, feature_mode=True, input_dim=(T, D), aggregation="none", num_heads=8, num_layers=2),
    "transformer (12 heads, 4 layers, ff 768)": P.TransformerProbe(None, [], C, feature_mode=True, input_dim=(T, D), aggregation="none"),
    "lstm (2 layers, 256 units)": P.LSTMProbe(None, [], C, feature_mode=True, input_dim=(T, D), aggregation="none"),
    "lstm (2 layers, 256 units, bidirectional)": P.LSTMProbe(None, [], C, feature_mode=True, input_dim=(T, D), aggregation="none", bidirectional=True),
}
# the shapes of the reference's shipped evaluation configs (configs/evaluation_configs/*: 8 heads, attention_dim 128, 1 layer; 128 units -> 300)
probes["transformer, shipped config (8 heads of 96, ff 128, 1 layer)"] = P.TransformerProbe(None, [], C, feature_mode=True, input_dim=(T, D), aggregation="none",
                                                                                             num_heads=8, attention_dim=128, num_layers=1, max_sequence_length=1200)
probes["lstm, shipped offline config (1 layer, 300 units, bidirectional)"] = P.LSTMProbe(None, [], C, feature_mode=True, input_dim=(T, D), aggregation="none", lstm_hidden_size=64,
                                                                                            num_layers=1, bidirectional=True, max_sequence_length=1200)
probes["lstm, shipped config (2 layers, 300 units)"] = P.LSTMProbe(None, [], C, feature_mode=True, input_dim=(T, D), aggregation="none", lstm_hidden_size=128,
                                                                    num_layers=2, max_sequence_length=1200)
for name, pr in probes.items():
    for k, v in pr.state_dict().items():
        if v.dtype == torch.float32 and "norm" not in k:
            v.normal_(0, 0.03)
    print(f"{name:62s} {timeit(lambda: pr(x)):8.2f} ms per {B} clips")
