"""EfficientNet-B0 ``features`` stack on the MI355X kernels.

The reference (avex/models/efficientnet.py:55-66, 116-137, 208) feeds a mel spectrogram, repeated to three channels, to
``torchvision.models.efficientnet_b0().features``.  Here:

* activations are NHWC in the operand type, channels padded to a multiple of 128 (padding channels are exactly zero in every
  layer: zero weights, zero bias, SiLU(0) = 0), so every 1x1 convolution (expand, project, head: all of the FLOPs that matter)
  is an ``avexhip_gemm`` over ``[B*H*W, C]`` rows with eval-mode BatchNorm folded into weight and bias, SiLU (``gelu = 2``) or
  the residual add in the epilogue;
* the three identical input channels collapse the stem into a 1-channel 3x3 convolution with channel-summed weights;
* depthwise convolutions, the squeeze-excitation pool (accumulated by the depthwise kernel on its way out), its two tiny
  fully connected layers and the channel rescale are the bandwidth-bound HIP kernels of ``csrc/effnet.hip``.

PARITY UNPINNED: torchvision is third-party, absent from the reference tree and both machines; the checker is
``oracle/effnet_oracle.py`` on synthetic weights.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Mapping, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _capi
from . import kernels as K
from .synth import EFFNET_B0_STAGES

__all__ = ["EfficientNetB0Encoder"]


def _pad128(c: int) -> int:
    return ((c + 127) // 128) * 128


class EfficientNetB0Encoder:
    """``[B, n_mels, frames]`` fp32 mel images on the GPU -> ``features [B, 1280, H', W']`` (and the reference's hook taps)."""

    def __init__(self, state: Mapping[str, np.ndarray], operand_dtype: str = "f16", prefix: str = "model.", stages: Sequence = EFFNET_B0_STAGES,
                 bn_eps: float = 1e-5) -> None:
        _capi.require_gpu()
        self.dtype = operand_dtype
        self.stages = [tuple(int(v) for v in s) for s in stages]
        dev = torch.device("cuda", torch.cuda.current_device())
        self.dev = dev
        get = lambda n: np.asarray(state[prefix + n], np.float32)

        def bnfold(name):
            sc = get(name + ".weight") / np.sqrt(get(name + ".running_var") + np.float32(bn_eps))
            return sc.astype(np.float32), (get(name + ".bias") - get(name + ".running_mean") * sc).astype(np.float32)

        def t32(a):
            return torch.from_numpy(np.ascontiguousarray(a.astype(np.float32))).to(dev)

        def pw(conv, bn, kp=None):                # 1x1 conv + BN -> ([Np, Kp] half weight, [Np] fp32 bias, scale, shift)
            w = get(conv + ".weight")[:, :, 0, 0]
            sc, sh = bnfold(bn)
            N, Kd = w.shape
            wp = np.zeros((_pad128(N), kp or _pad128(Kd)), np.float32)      # Kp = channel padding of the activation it reads
            wp[:N, :Kd] = w * sc[:, None]
            bp = np.zeros((_pad128(N),), np.float32); bp[:N] = sh
            return K.to_half(t32(wp), operand_dtype), t32(bp), sc, sh

        # stem: channel-summed (the three input channels are copies), BN-folded, [9, Cp]
        w0 = get("features.0.0.weight").sum(axis=1)                                      # [32, 3, 3]
        sc, sh = bnfold("features.0.1")
        c0 = w0.shape[0]
        # GEMM outputs need a channel count that is a multiple of 128 (N), GEMM inputs only a multiple of 64 (K): the stem and
        # the depthwise layer behind it (the highest-resolution tensors of the network) carry 64 channels, not 128
        cp = ((c0 + 63) // 64) * 64
        ws = np.zeros((9, cp), np.float32); ws[:, :c0] = (w0 * sc[:, None, None]).reshape(c0, 9).T
        bs = np.zeros((cp,), np.float32); bs[:c0] = sh
        self.stem = (t32(ws), t32(bs), c0, sc, sh)
        self.blocks = []
        for si, (er, k, s, cin, cout, n) in enumerate(self.stages, start=1):
            for j in range(n):
                ci = cin if j == 0 else cout
                ce = ci * er
                p = f"features.{si}.{j}.block."
                d = 1 if er != 1 else 0
                blk = dict(name=prefix + p, k=k, stride=s if j == 0 else 1, cin=ci, cexp=ce, cout=cout, tap=(d == 1))
                if d:
                    blk["expand"] = pw(p + "0.0", p + "0.1", kp=cp)[:2]
                    cp = _pad128(ce)
                wd = get(p + f"{d}.0.weight")[:, 0]                                      # [ce, k, k]
                scd, shd = bnfold(p + f"{d}.1")
                wdp = np.zeros((k * k, cp), np.float32); wdp[:, :ce] = (wd * scd[:, None, None]).reshape(ce, k * k).T
                bdp = np.zeros((cp,), np.float32); bdp[:ce] = shd
                blk["dw"] = (t32(wdp), t32(bdp))
                blk["se"] = (t32(get(p + f"{d + 1}.fc1.weight")[:, :, 0, 0]), t32(get(p + f"{d + 1}.fc1.bias")),
                             t32(get(p + f"{d + 1}.fc2.weight")[:, :, 0, 0]), t32(get(p + f"{d + 1}.fc2.bias")))
                wp_, bp_, scp, shp = pw(p + f"{d + 2}.0", p + f"{d + 2}.1", kp=cp)
                cp = _pad128(cout)
                blk["project"] = (wp_, bp_)
                blk["project_bn"] = (t32(scp), t32(shp))
                self.blocks.append(blk)
        last = len(self.stages) + 1
        wh, bh, sch, shh = pw(f"features.{last}.0", f"features.{last}.1", kp=cp)
        self.head = (wh, bh, t32(sch), t32(shh), int(sch.shape[0]))
        self.head_name = prefix + f"features.{last}.0"
        self.stem_name = prefix + "features.0.0"

    def tap_names(self) -> List[str]:
        return [self.stem_name] + [b["name"] + "3.0" for b in self.blocks if b["tap"]] + [self.head_name]

    @torch.no_grad()
    def forward(self, mel: torch.Tensor, hook_layers: Iterable[str] = (), want_features: bool = True, want_pooled: bool = False) -> Dict[str, object]:
        """``hook_layers``: names out of ``tap_names()``; their raw (pre-BatchNorm) convolution outputs come back as fp32 NCHW."""
        if mel.dim() != 3 or mel.dtype != torch.float32 or not mel.is_cuda:
            raise ValueError("mel must be a [B, n_mels, frames] float32 CUDA tensor")
        hooks = set(hook_layers)
        out: Dict[str, object] = {"hooks": {}}
        B = mel.shape[0]
        ws, bs, c0, sc0, sh0 = self.stem
        if self.stem_name in hooks:
            x, raw = K.effnet_stem(mel, ws, bs, self.dtype, want_raw=True)
            # raw holds the BN output before SiLU; the tap is the convolution before its BatchNorm
            r = (raw[..., :c0] - torch.from_numpy(sh0).to(raw.device)) / torch.from_numpy(sc0).to(raw.device)
            out["hooks"][self.stem_name] = r.permute(0, 3, 1, 2).contiguous()
        else:
            x = K.effnet_stem(mel, ws, bs, self.dtype)
        for blk in self.blocks:
            Bn, H, W, Cp = x.shape
            inp = x
            if "expand" in blk:
                we, be = blk["expand"]
                x = K.gemm(x.view(Bn * H * W, Cp), we, bias=be, silu=True, out_f32=False, out_half=True)["half"].view(Bn, H, W, -1)
            wd, bd = blk["dw"]
            x, pool = K.effnet_dwconv(x, wd, bd, blk["k"], blk["stride"])
            K.effnet_se(x, pool, blk["cexp"], *blk["se"])
            Bn, H2, W2, Ce = x.shape
            wp, bp = blk["project"]
            res = inp.view(Bn * H * W, Cp) if (blk["stride"] == 1 and blk["cin"] == blk["cout"]) else None
            tap = blk["tap"] and (blk["name"] + "3.0") in hooks
            r = K.gemm(x.view(Bn * H2 * W2, Ce), wp, bias=bp, resid_half=res, alpha=1.0, out_f32=False, out_half=True, out_raw=tap)
            if tap:        # out_raw = conv * bn_scale + bn_shift (before the residual); undo the folded BatchNorm for the tap
                scp, shp = blk["project_bn"]
                c = blk["cout"]
                rr = (r["raw"][:, :c] - shp) / scp
                out["hooks"][blk["name"] + "3.0"] = rr.view(Bn, H2, W2, c).permute(0, 3, 1, 2).contiguous()
            x = r["half"].view(Bn, H2, W2, -1)
        Bn, H, W, Cp = x.shape
        wh, bh, sch, shh, ch = self.head
        tap = self.head_name in hooks
        r = K.gemm(x.view(Bn * H * W, Cp), wh, bias=bh, silu=True, out_f32=True, out_half=False, out_raw=tap)
        if tap:
            out["hooks"][self.head_name] = ((r["raw"][:, :ch] - shh) / sch).view(Bn, H, W, ch).permute(0, 3, 1, 2).contiguous()
        f = r["f32"][:, :ch].view(Bn, H, W, ch)
        if want_features:
            out["features"] = f.permute(0, 3, 1, 2).contiguous()                       # (B, C, H, W) like the reference
        if want_pooled:
            out["pooled"] = K.mean_pool(f.reshape(Bn, H * W, ch).contiguous())
        return out
