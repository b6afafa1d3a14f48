"""CPU cross-check of the EfficientNet oracle (row a17, parity unpinned: torchvision is absent).

`oracle/effnet_oracle.py` is NumPy (einsum convolutions, explicit BatchNorm algebra).  This builds the same published
architecture a second way -- PyTorch's own `Conv2d` / `BatchNorm2d` / `SiLU` modules arranged as torchvision's `efficientnet_b0` /
`efficientnet_b1` arrange them (Conv2dNormActivation, MBConv with SqueezeExcitation, parameter names `features.N.M.block.K...`) --
loads the synthetic checkpoint into it with `strict=True` (so every key name and shape the wrapper expects exists) and compares
the two.  It says the oracle's arithmetic is PyTorch's arithmetic for this topology; it cannot say the topology is torchvision's
(only torchvision can), which is why the row stays "unpinned".
"""
import numpy as np
import pytest
import torch
from torch import nn

from avex_amd import synth
from oracle import effnet_oracle as EO


class _CNA(nn.Sequential):          # torchvision.ops.Conv2dNormActivation: conv (no bias) -> BatchNorm2d -> activation
    def __init__(self, cin, cout, k, stride, groups=1, act=True):
        layers = [nn.Conv2d(cin, cout, k, stride, (k - 1) // 2, groups=groups, bias=False), nn.BatchNorm2d(cout, eps=1e-5)]
        if act:
            layers.append(nn.SiLU())
        super().__init__(*layers)


class _SE(nn.Module):               # torchvision.ops.SqueezeExcitation(activation=SiLU, scale_activation=Sigmoid)
    def __init__(self, c, squeeze):
        super().__init__()
        self.fc1 = nn.Conv2d(c, squeeze, 1)
        self.fc2 = nn.Conv2d(squeeze, c, 1)

    def forward(self, x):
        s = x.mean((2, 3), keepdim=True)
        return x * torch.sigmoid(self.fc2(nn.functional.silu(self.fc1(s))))


class _MBConv(nn.Module):
    def __init__(self, er, k, stride, cin, cout):
        super().__init__()
        self.res = stride == 1 and cin == cout
        mid = cin * er
        layers = []
        if er != 1:
            layers.append(_CNA(cin, mid, 1, 1))
        layers.append(_CNA(mid, mid, k, stride, groups=mid))
        layers.append(_SE(mid, max(1, cin // 4)))
        layers.append(_CNA(mid, cout, 1, 1, act=False))
        self.block = nn.Sequential(*layers)

    def forward(self, x):
        y = self.block(x)
        return x + y if self.res else y     # StochasticDepth is the identity in eval mode


def _features(stages, head=1280):
    mods = [_CNA(3, stages[0][3], 3, 2)]
    for er, k, s, cin, cout, n in stages:
        mods.append(nn.Sequential(*[_MBConv(er, k, s if j == 0 else 1, cin if j == 0 else cout, cout) for j in range(n)]))
    mods.append(_CNA(stages[-1][4], head, 1, 1))
    return nn.Sequential(*mods)


@pytest.mark.parametrize("variant", ["b0", "b1"])
def test_effnet_oracle_matches_a_pytorch_restatement(variant):
    stages = synth.EFFNET_STAGES[variant]
    sd = synth.effnet_b0_state_dict(seed=0, stages=stages)
    net = nn.Module()
    net.features = _features(stages)
    own = net.state_dict()
    mapped = {k[len("model."):]: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items() if k.startswith("model.features.")}
    for k in own:
        if k.endswith("num_batches_tracked"):
            mapped.setdefault(k, own[k])
    net.load_state_dict(mapped, strict=True)
    net.eval()
    mel = synth.normal("effnet.mel", (2, 64, 96), 1.0)
    ours, taps = EO.effnet_features(mel, sd, stages)
    got = {}
    hooks = []
    for name in taps:
        mod = net.get_submodule(name[len("model."):])
        hooks.append(mod.register_forward_hook(lambda m, i, o, name=name: got.__setitem__(name, o.detach().numpy())))
    torch.set_num_threads(4)
    with torch.no_grad():
        theirs = net.features(torch.from_numpy(mel)[:, None].repeat(1, 3, 1, 1)).numpy()
    for h in hooks:
        h.remove()
    assert ours.shape == theirs.shape == (2, 1280, 2, 3)
    rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))      # noqa: E731
    assert rel(ours, theirs) < 2e-5
    assert len(got) == len(taps) == (17 if variant == "b0" else 23)
    for name in taps:
        assert taps[name].shape == got[name].shape and rel(taps[name], got[name]) < 2e-5, name
