"""CPU oracle: NumPy restatement of the EfficientNet-B0 ``features`` stack the reference runs on mel spectrograms.

TEST INFRASTRUCTURE ONLY (same rules as beats_oracle.py).

PARITY UNPINNED.  The reference wrapper (avex/models/efficientnet.py:21-215) builds ``torchvision.models.efficientnet_b0`` and
returns ``self.model.features(x)`` for ``x = mel.unsqueeze(1).repeat(1, 3, 1, 1)`` (:133-135, :208).  torchvision (0.26.0 in the
reference's uv.lock) is third-party, is not under /root/reference and is not installed here; nothing in the development
container can produce a golden.  This file restates torchvision's published EfficientNet-B0 (V1) as documented:

  features.0    Conv2d(3, 32, 3, stride 2, padding 1, bias False) -> BatchNorm2d -> SiLU
  features.1-7  MBConv stages (expand ratio, kernel, stride, in, out, layers) =
                (1,3,1,32,16,1) (6,3,2,16,24,2) (6,5,2,24,40,2) (6,3,2,40,80,3) (6,5,1,80,112,3) (6,5,2,112,192,4) (6,3,1,192,320,1)
                MBConv = [1x1 expand + BN + SiLU (ratio != 1)] -> depthwise k x k (padding (k-1)/2) + BN + SiLU ->
                SqueezeExcitation(squeeze = in // 4: avgpool -> fc1 -> SiLU -> fc2 -> sigmoid -> scale) -> 1x1 project + BN;
                residual when stride 1 and in == out (stochastic depth is the identity in eval)
  features.8    Conv2d(320, 1280, 1, bias False) -> BatchNorm2d -> SiLU
  BatchNorm in eval mode (running statistics, eps 1e-5).

Taps (efficientnet.py:82-114): ``model.features.0.0``, every ``*.block.3.0`` and ``model.features.8.0`` -- the raw convolution
outputs BEFORE their BatchNorm.
"""
from __future__ import annotations

from typing import Dict, Mapping, Tuple

import numpy as np


def silu(x: np.ndarray) -> np.ndarray:
    return (x / (1.0 + np.exp(-x))).astype(np.float32)


def conv2d(x: np.ndarray, w: np.ndarray, stride: int, pad: int, groups: int = 1) -> np.ndarray:
    """``x [B, Cin, H, W]``, ``w [Cout, Cin/groups, k, k]`` -> ``[B, Cout, Ho, Wo]`` fp32."""
    B, Cin, H, W = x.shape
    Cout, cg, k, _ = w.shape
    xp = np.pad(x, ((0, 0), (0, 0), (pad, pad), (pad, pad)))
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    iy = np.arange(Ho)[:, None] * stride + np.arange(k)[None, :]
    ix = np.arange(Wo)[:, None] * stride + np.arange(k)[None, :]
    win = xp[:, :, iy[:, None, :, None], ix[None, :, None, :]]          # [B, Cin, Ho, Wo, k, k]
    if groups == 1:
        return np.einsum("bchwyx,ocyx->bohw", win, w, optimize=True).astype(np.float32)
    assert groups == Cin == Cout and cg == 1
    return np.einsum("bchwyx,cyx->bchw", win, w[:, 0], optimize=True).astype(np.float32)


def bn(x: np.ndarray, sd: Mapping[str, np.ndarray], name: str, eps: float = 1e-5) -> np.ndarray:
    sc = sd[name + ".weight"] / np.sqrt(sd[name + ".running_var"] + np.float32(eps))
    sh = sd[name + ".bias"] - sd[name + ".running_mean"] * sc
    return (x * sc[None, :, None, None] + sh[None, :, None, None]).astype(np.float32)


def effnet_features(mel: np.ndarray, sd: Mapping[str, np.ndarray], stages, pre: str = "model.") -> Tuple[np.ndarray, Dict[str, np.ndarray]]:
    """``mel [B, H, W]`` -> (features ``[B, 1280, H', W']``, taps name -> raw conv output ``[B, C, h, w]``)."""
    taps: Dict[str, np.ndarray] = {}
    x = np.repeat(np.asarray(mel, np.float32)[:, None], 3, axis=1)                      # efficientnet.py:133-135
    y = conv2d(x, sd[pre + "features.0.0.weight"], 2, 1)
    taps[pre + "features.0.0"] = y
    x = silu(bn(y, sd, pre + "features.0.1"))
    for si, (er, k, s, cin, cout, n) in enumerate(stages, start=1):
        for j in range(n):
            p = pre + f"features.{si}.{j}.block."
            st = s if j == 0 else 1
            inp = x
            d = 0
            if er != 1:
                x = silu(bn(conv2d(x, sd[p + "0.0.weight"], 1, 0), sd, p + "0.1")); d = 1
            x = silu(bn(conv2d(x, sd[p + f"{d}.0.weight"], st, (k - 1) // 2, groups=x.shape[1]), sd, p + f"{d}.1"))
            m = x.mean(axis=(2, 3), dtype=np.float32)                                   # squeeze
            h = silu(m @ sd[p + f"{d + 1}.fc1.weight"][:, :, 0, 0].T + sd[p + f"{d + 1}.fc1.bias"])
            e = h @ sd[p + f"{d + 1}.fc2.weight"][:, :, 0, 0].T + sd[p + f"{d + 1}.fc2.bias"]
            x = x * (1.0 / (1.0 + np.exp(-e)))[:, :, None, None].astype(np.float32)     # excitation
            y = conv2d(x, sd[p + f"{d + 2}.0.weight"], 1, 0)
            if d == 1:
                taps[p + "3.0"] = y
            x = bn(y, sd, p + f"{d + 2}.1")
            if st == 1 and inp.shape[1] == x.shape[1]:
                x = x + inp
    last = len(stages) + 1
    y = conv2d(x, sd[pre + f"features.{last}.0.weight"], 1, 0)
    taps[pre + f"features.{last}.0"] = y
    return silu(bn(y, sd, pre + f"features.{last}.1")), taps
