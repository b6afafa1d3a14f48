"""CPU oracle: NumPy restatement of the reference BEATs embedding path.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module; the product path
(``avex_amd``) never does and fails loudly when its HIP library is missing.

Each function cites the reference file:line it restates (paths relative to
``/root/reference``).  The oracle is *pinned*: ``tests/golden/make_goldens.py`` ran
the real reference (imported in the development container) on the same synthetic
weights/inputs and committed its outputs under ``tests/golden/``;
``tests/test_oracle_golden.py`` checks this file against those vectors and against
the known-answer values of SURVEY.md Appendix B.

All arithmetic is fp32 (like the reference's CPU path) unless noted.
"""
from __future__ import annotations

import math
from typing import Dict, List, Mapping, Optional, Sequence, Tuple

import numpy as np
from scipy.special import erf as _erf_scalar_loop

_POOL = None


def _erf(x: np.ndarray) -> np.ndarray:
    """``scipy.special.erf`` over row blocks on a thread pool (the ufunc loop releases the GIL and is single-threaded; it was a
    quarter of the oracle's time).  Elementwise, so the values are those of one call."""
    global _POOL
    x = np.ascontiguousarray(x)
    if x.size < (1 << 18):
        return _erf_scalar_loop(x)
    import os
    from concurrent.futures import ThreadPoolExecutor
    n = min(16, os.cpu_count() or 1)
    if _POOL is None:
        _POOL = ThreadPoolExecutor(max_workers=n)
    flat = x.reshape(-1)
    out = np.empty_like(flat)
    step = -(-flat.size // n)
    list(_POOL.map(lambda i: _erf_scalar_loop(flat[i:i + step], out=out[i:i + step]), range(0, flat.size, step)))
    return out.reshape(x.shape)

F32_EPS = np.float32(1.1920929e-07)  # torch.finfo(float32).eps, beats.py:36


# --------------------------------------------------------------------------------------
# Frontend: kaldi-compatible batched fbank  (avex/models/beats/beats.py:39-163)
# --------------------------------------------------------------------------------------
def hann_window(win_length: int = 400) -> np.ndarray:
    """kaldi ``window_type="hanning"`` (EAT frontend, eat/audio_processor.py:110-119)."""
    # torch.hann_window(periodic=False) is all-fp32: arange * (2 pi / (N-1)) -> cos -> * -0.5 + 0.5
    n = np.arange(win_length, dtype=np.float32)
    c = np.cos(n * np.float32(math.pi * 2 / (win_length - 1))).astype(np.float32)
    return (c * np.float32(-0.5) + np.float32(0.5)).astype(np.float32)


def povey_window(win_length: int = 400) -> np.ndarray:
    """``hann(win, periodic=False) ** 0.85`` (beats.py:75)."""
    return np.power(hann_window(win_length), np.float32(0.85)).astype(np.float32)


def mel_filterbank(n_fft: int = 512, n_mels: int = 128, sample_rate: float = 16000.0,
                   low_freq: float = 20.0, high_freq: float = 8000.0) -> np.ndarray:
    """Triangular kaldi mel bank ``[n_fft//2+1, n_mels]`` (beats.py:82-118)."""
    num_fft_bins = n_fft // 2
    fft_bin_width = sample_rate / n_fft
    mel_low = 1127.0 * math.log(1.0 + low_freq / 700.0)
    mel_high = 1127.0 * math.log(1.0 + high_freq / 700.0)
    mel_delta = (mel_high - mel_low) / (n_mels + 1)
    # torch arithmetic here is fp32 (python floats promoted to the int64->float32 tensor)
    bin_idx = np.arange(n_mels, dtype=np.float32)[:, None]
    left = (np.float32(mel_low) + bin_idx * np.float32(mel_delta)).astype(np.float32)
    center = (np.float32(mel_low) + (bin_idx + np.float32(1.0)) * np.float32(mel_delta)).astype(np.float32)
    right = (np.float32(mel_low) + (bin_idx + np.float32(2.0)) * np.float32(mel_delta)).astype(np.float32)
    freqs = (np.float32(fft_bin_width) * np.arange(num_fft_bins, dtype=np.float32)).astype(np.float32)
    mel_freqs = (np.float32(1127.0) * np.log(np.float32(1.0) + freqs / np.float32(700.0))).astype(np.float32)[None, :]
    up = (mel_freqs - left) / (center - left)
    down = (right - mel_freqs) / (right - center)
    fb = np.maximum(np.float32(0.0), np.minimum(up, down)).astype(np.float32)
    fb = np.pad(fb, ((0, 0), (0, 1)))  # Nyquist column = 0 (beats.py:117)
    return np.ascontiguousarray(fb.T)


def fbank(wav: np.ndarray, *, n_mels: int = 128, win_length: int = 400, hop_length: int = 160,
          preemph: float = 0.97, window: Optional[np.ndarray] = None,
          mel_fb: Optional[np.ndarray] = None) -> np.ndarray:
    """``_BatchedFbank.forward`` (beats.py:120-163).  ``wav`` already scaled (x 2**15 for BEATs)."""
    wav = np.asarray(wav, np.float32)
    n_fft = 1
    while n_fft < win_length:
        n_fft *= 2
    if window is None:
        window = povey_window(win_length)
    if mel_fb is None:
        mel_fb = mel_filterbank(n_fft, n_mels)
    B, T = wav.shape
    n_frames = 1 + (T - win_length) // hop_length if T >= win_length else 0
    idx = (np.arange(n_frames)[:, None] * hop_length + np.arange(win_length)[None, :])
    frames = wav[:, idx]                                            # unfold (beats.py:136)
    frames = frames - frames.mean(axis=-1, keepdims=True, dtype=np.float32)   # (beats.py:140)
    shifted = np.concatenate([frames[..., :1], frames[..., :-1]], axis=-1)    # replicate pad (beats.py:143)
    frames = frames - np.float32(preemph) * shifted                 # (beats.py:144)
    frames = frames * window                                        # (beats.py:147)
    frames = np.pad(frames, ((0, 0), (0, 0), (0, n_fft - win_length)))  # (beats.py:151)
    spec = np.fft.rfft(frames.astype(np.float32), axis=-1)          # (beats.py:154) fp32 pocketfft in numpy>=2
    spec = spec.astype(np.complex64)
    power = (np.abs(spec).astype(np.float32)) ** np.float32(2.0)    # (beats.py:155)
    mel = power.astype(np.float32) @ mel_fb                         # (beats.py:159)
    return np.log(np.maximum(mel, F32_EPS)).astype(np.float32)      # (beats.py:163)


def beats_preprocess(wav: np.ndarray, cfg: Mapping[str, object]) -> np.ndarray:
    """``BEATs.preprocess`` (beats.py:304-323): x*2**15 -> fbank -> (x-mean)/(2*std)."""
    sr = float(cfg.get("sample_frequency", 16000.0))
    win = int(sr * float(cfg.get("frame_length", 25.0)) / 1000.0)
    hop = int(sr * float(cfg.get("frame_shift", 10.0)) / 1000.0)
    fb = fbank(np.asarray(wav, np.float32) * np.float32(2 ** 15), n_mels=int(cfg.get("num_mel_bins", 128)),
               win_length=win, hop_length=hop)
    mean = np.float32(cfg.get("fbank_mean", 15.41663)); std = np.float32(cfg.get("fbank_std", 6.55582))
    return ((fb - mean) / (np.float32(2.0) * std)).astype(np.float32)


def stft_power(wav: np.ndarray, n_fft: int, hop: int, win_length: int, window: np.ndarray, center: bool = True) -> np.ndarray:
    """``torch.stft(..., return_complex=True).abs().pow(2)`` (audio_utils.py:139-148): reflect padding of n_fft // 2 when
    ``center``, window centre-padded to n_fft, one-sided.  ``[B, T]`` -> ``[B, n_fft // 2 + 1, frames]`` fp32."""
    x = np.asarray(wav, np.float32)
    if center:
        x = np.pad(x, ((0, 0), (n_fft // 2, n_fft // 2)), mode="reflect")
    w = np.zeros(n_fft, np.float32)
    lo = (n_fft - win_length) // 2
    w[lo:lo + win_length] = window
    frames = 1 + (x.shape[1] - n_fft) // hop
    idx = np.arange(frames)[:, None] * hop + np.arange(n_fft)[None, :]
    spec = np.fft.rfft((x[:, idx] * w).astype(np.float32), axis=-1).astype(np.complex64)
    return (np.abs(spec).astype(np.float32) ** np.float32(2.0)).transpose(0, 2, 1)


def htk_mel_fb(n_freqs: int, n_mels: int, sample_rate: int) -> np.ndarray:
    """``torchaudio.transforms.MelScale(n_mels, sample_rate, n_stft).fb`` (audio_utils.py:97-101; torchaudio defaults f_min 0,
    f_max sr // 2, norm None, mel_scale "htk"), restated from torchaudio's documented ``melscale_fbanks``: triangles on points
    equally spaced in mel(f) = 2595 log10(1 + f / 700), evaluated at the linear bin frequencies.  UNPINNED (torchaudio absent)."""
    all_freqs = np.linspace(0.0, sample_rate // 2, n_freqs)
    m = np.linspace(0.0, 2595.0 * np.log10(1.0 + (sample_rate // 2) / 700.0), n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]
    return np.maximum(0.0, np.minimum(-slopes[:, :-2] / f_diff[:-1], slopes[:, 2:] / f_diff[1:])).astype(np.float32)


def audio_processor(wav: np.ndarray, *, n_fft: int, hop: int, win_length: Optional[int] = None, window: str = "hann", n_mels: int = 128,
                    sample_rate: int = 16000, representation: str = "mel_spectrogram", center: bool = True,
                    normalize: bool = True) -> np.ndarray:
    """``AudioProcessor.__call__`` (audio_utils.py:106-172) for the spectrogram representations."""
    win_length = win_length or n_fft
    n = np.arange(win_length, dtype=np.float64)
    w = ((0.5 - 0.5 * np.cos(2 * np.pi * n / win_length)) if window == "hann" else (0.54 - 0.46 * np.cos(2 * np.pi * n / win_length))).astype(np.float32)
    x = stft_power(wav, n_fft, hop, win_length, w, center)
    if representation == "mel_spectrogram":
        x = np.einsum("fm,bft->bmt", htk_mel_fb(n_fft // 2 + 1, n_mels, sample_rate), x).astype(np.float32)     # MelScale: fb^T @ spec
    if normalize:
        x = np.log(x + np.float32(1e-6)).astype(np.float32)
        mn = x.min(axis=(1, 2), keepdims=True); mx = x.max(axis=(1, 2), keepdims=True)
        x = (x - mn) / (mx - mn + np.float32(1e-8))
    return x.astype(np.float32)


def eat_preprocess(wav: np.ndarray, *, sample_rate: int = 16000, target_length: int = 1024, n_mels: int = 128,
                   norm_mean: float = -4.268, norm_std: float = 4.569, frame_shift_ms: int = 10) -> np.ndarray:
    """``EATAudioProcessor.__call__`` (avex/models/eat/audio_processor.py:72-143): per clip ``mono - mono.mean()`` (:107),
    kaldi fbank with a Hann window, dither 0, NO 2**15 scale (:110-119; torchaudio.compliance.kaldi.fbank defaults: 25 ms
    frames, pre-emphasis 0.97, per-frame DC removal, snip_edges, power-of-two FFT, low_freq 20, log of max(., fp32 eps)),
    zero-pad / truncate to ``target_length`` frames (:121-126), ``(mel - norm_mean) / (2 norm_std)`` (:135), or per-sample
    statistics when the constants are (0, 1) (:131-134).  Returns ``(B, target_length, n_mels)``.

    PARITY UNPINNED against torchaudio itself: ``torchaudio`` (2.11.0 in the reference's uv.lock) is not installed in the
    development container, so no golden of this function exists.  It is pinned transitively: ``fbank`` above is the same
    kaldi algorithm (the reference's own ``_BatchedFbank`` is tested against torchaudio's, tests/unittests/test_batched_fbank.py)
    and IS pinned by the reference goldens; this function only swaps the window and the scale around it."""
    wav = np.asarray(wav, np.float32)
    if wav.ndim == 1:
        wav = wav[None]
    hop = int(round(sample_rate * frame_shift_ms / 1000))
    win = int(sample_rate * 25 / 1000)
    out = []
    for mono in wav:
        mono = mono - mono.mean(dtype=np.float32)                                   # :107
        mel = fbank(mono[None], n_mels=n_mels, win_length=win, hop_length=hop, window=hann_window(win))[0]
        t = mel.shape[0]
        if t < target_length:
            mel = np.pad(mel, ((0, target_length - t), (0, 0)))                      # :123
        else:
            mel = mel[:target_length]                                               # :125
        if norm_mean == 0.0 and norm_std == 1.0:
            std = mel.std(ddof=1)
            mel = (mel - mel.mean()) / ((std if std > 0 else 1.0) * 2)
        else:
            mel = (mel - np.float32(norm_mean)) / (np.float32(norm_std) * np.float32(2))   # :135
        out.append(mel.astype(np.float32))
    return np.stack(out, 0)


# --------------------------------------------------------------------------------------
# Encoder building blocks
# --------------------------------------------------------------------------------------
def layer_norm(x: np.ndarray, w: np.ndarray, b: np.ndarray, eps: float = 1e-5) -> np.ndarray:
    """``torch.nn.LayerNorm`` over the last dim (biased variance), fp32."""
    mu = x.mean(axis=-1, keepdims=True, dtype=np.float32)
    xc = x - mu
    var = (xc * xc).mean(axis=-1, keepdims=True, dtype=np.float32)
    return (xc / np.sqrt(var + np.float32(eps)) * w + b).astype(np.float32)


def gelu_erf(x: np.ndarray) -> np.ndarray:
    """Exact-erf GELU (modules.py:191-200 -> F.gelu; nn.GELU() at backbone.py:68)."""
    x = x.astype(np.float32, copy=False)
    t = _erf(x * np.float32(1.0 / math.sqrt(2.0))).astype(np.float32, copy=False)
    t += np.float32(1.0)
    t *= np.float32(0.5) * x          # same operations in the same order as 0.5 * x * (1 + erf(.)), without the temporaries
    return t


def ffn_hidden(x: np.ndarray, p: Mapping[str, np.ndarray], pre: str, activation: str) -> np.ndarray:
    """``activation_fn(fc1(x))`` for the activations ``get_activation_fn`` knows (modules.py:203-237), or the gated linear unit that
    replaces fc1 when ``activation_fn == "glu"`` (``GLU_Linear(E, F, "swish")``, backbone.py:296-297; modules.py:155-171: one Linear to
    2F, the first F columns times swish of the second F)."""
    if activation == "glu":
        y = linear(x, p[pre + "fc1.linear.weight"], p[pre + "fc1.linear.bias"])
        F = y.shape[-1] // 2
        g = y[..., F:]
        return (y[..., :F] * (g / (np.float32(1.0) + np.exp(-g)))).astype(np.float32)
    y = linear(x, p[pre + "fc1.weight"], p[pre + "fc1.bias"])
    if activation == "gelu":
        return gelu_erf(y)
    if activation == "relu":
        return np.maximum(y, np.float32(0.0))
    if activation in ("gelu_accurate", "gelu_fast"):      # modules.py:177-188
        a = np.float32(math.sqrt(2.0 / math.pi))
        return (np.float32(0.5) * y * (np.float32(1.0) + np.tanh(a * (y + np.float32(0.044715) * y * y * y)))).astype(np.float32)
    if activation == "tanh":
        return np.tanh(y).astype(np.float32)
    if activation == "linear":
        return y
    raise RuntimeError(f"--activation-fn {activation} not supported")      # modules.py:237


def linear(x: np.ndarray, w: np.ndarray, b: Optional[np.ndarray]) -> np.ndarray:
    y = x.astype(np.float32, copy=False) @ w.T.astype(np.float32, copy=False)
    if b is not None:
        y += b.astype(np.float32, copy=False)
    return y.astype(np.float32, copy=False)


def relative_position_bucket(rel: np.ndarray, num_buckets: int, max_distance: int) -> np.ndarray:
    """Bidirectional T5 buckets (backbone.py:438-473).  ``rel = j - i`` (memory - context)."""
    nb = num_buckets // 2
    out = (rel > 0).astype(np.int64) * nb
    a = np.abs(rel).astype(np.int64)
    max_exact = nb // 2
    is_small = a < max_exact
    with np.errstate(divide="ignore"):
        # torch: log(rel.float()/max_exact) / math.log(max_distance/max_exact) * (nb-max_exact) in fp32
        big = (np.log(a.astype(np.float32) / np.float32(max_exact))
               / np.float32(math.log(max_distance / max_exact)) * np.float32(nb - max_exact))
    big = np.where(is_small, 0, big)
    big = max_exact + big.astype(np.int64)   # .to(torch.long) truncates toward zero (values >= 0)
    big = np.minimum(big, nb - 1)
    return out + np.where(is_small, a, big)


def position_bias(table: np.ndarray, T: int, num_buckets: int, max_distance: int) -> np.ndarray:
    """``compute_bias`` (backbone.py:475-492): ``[H, T, T]`` from the shared ``(num_buckets, H)`` table."""
    ctx = np.arange(T)[:, None]
    mem = np.arange(T)[None, :]
    bucket = relative_position_bucket(mem - ctx, num_buckets, max_distance)
    return np.ascontiguousarray(table[bucket].transpose(2, 0, 1)).astype(np.float32)


def pos_conv_weight(g: np.ndarray, v: np.ndarray) -> np.ndarray:
    """weight_norm(dim=2): ``w = g * v / ||v||`` with the norm over dims (0,1) per tap
    (backbone.py:67; torch._weight_norm)."""
    norm = np.sqrt((v.astype(np.float32) ** 2).sum(axis=(0, 1), keepdims=True, dtype=np.float32))
    return (v * (g / norm)).astype(np.float32)


def pos_conv(x: np.ndarray, w: np.ndarray, bias: np.ndarray, groups: int) -> np.ndarray:
    """Grouped Conv1d(k, pad=k//2) + SamePad (drop last if k even) + GELU on ``x[B,T,C]``
    (backbone.py:52-68,172-173; modules.py:67-93).  Returns ``[B,T,C]``."""
    B, T, C = x.shape
    K = w.shape[2]
    cg = C // groups
    pad = K // 2
    xp = np.pad(x, ((0, 0), (pad, pad), (0, 0)))
    T_out = T + 2 * pad - K + 1
    if K % 2 == 0:
        T_out -= 1
    out = np.empty((B, T_out, C), np.float32)
    # im2col per group: cols[b,t,(k,c)] = xp[b,t+k,g*cg+c]
    tidx = np.arange(T_out)[:, None] + np.arange(K)[None, :]
    for g in range(groups):
        xg = xp[:, :, g * cg:(g + 1) * cg]                       # [B,Tp,cg]
        cols = xg[:, tidx, :].reshape(B, T_out, K * cg)          # [B,T,K*cg]
        wg = w[g * cg:(g + 1) * cg].transpose(2, 1, 0).reshape(K * cg, cg)  # [(k,c), o]
        out[:, :, g * cg:(g + 1) * cg] = cols @ wg
    out += bias
    return gelu_erf(out)


def attention(x: np.ndarray, p: Mapping[str, np.ndarray], pre: str, H: int, bias_hTT: Optional[np.ndarray],
              gru_rel_pos: bool, key_padding_mask: Optional[np.ndarray] = None) -> np.ndarray:
    """``_MultiheadAttention.forward`` (backbone.py:494-574) on ``x[B,T,E]`` (batch-first here;
    the reference runs ``(T,B,E)``, the arithmetic per (b,t) is identical)."""
    B, T, E = x.shape
    hd = E // H
    q = linear(x, p[pre + "q_proj.weight"], p[pre + "q_proj.bias"]).reshape(B, T, H, hd).transpose(0, 2, 1, 3)
    k = linear(x, p[pre + "k_proj.weight"], p[pre + "k_proj.bias"]).reshape(B, T, H, hd).transpose(0, 2, 1, 3)
    v = linear(x, p[pre + "v_proj.weight"], p[pre + "v_proj.bias"]).reshape(B, T, H, hd).transpose(0, 2, 1, 3)
    gate = None
    if bias_hTT is not None and gru_rel_pos:
        g8 = linear(q, p[pre + "grep_linear.weight"], p[pre + "grep_linear.bias"])   # [B,H,T,8]
        g2 = g8.reshape(B, H, T, 2, 4).sum(-1, dtype=np.float32)
        sg = (np.float32(1.0) / (np.float32(1.0) + np.exp(-g2))).astype(np.float32)
        gate_a, gate_b = sg[..., 0:1], sg[..., 1:2]
        grep_a = p[pre + "grep_a"].reshape(1, H, 1, 1)
        gate = gate_a * (gate_b * grep_a - np.float32(1.0)) + np.float32(2.0)        # backbone.py:550
    o = np.empty((B, H, T, hd), np.float32)
    kt = k.transpose(0, 1, 3, 2)
    tmp = np.empty((H, T, T), np.float32) if gate is not None else None
    # clip by clip, every step in place: the same operations per element in the same order as the whole-batch expressions
    # (scores * scale, + gate * bias, mask, - max, exp, / sum), without [B, H, T, T] temporaries -- the CPU baseline of bench.py times this
    for b in range(B):
        sc = np.matmul(q[b], kt[b])                                               # [H,T,T]
        sc *= np.float32(hd ** -0.5)                                              # SDPA scale (backbone.py:567)
        if bias_hTT is not None:
            if gate is not None:
                np.multiply(gate[b], bias_hTT, out=tmp)
                sc += tmp
            else:
                sc += bias_hTT
        if key_padding_mask is not None:
            sc[:, :, key_padding_mask[b]] = -np.inf
        sc -= sc.max(axis=-1, keepdims=True)
        np.exp(sc, out=sc)
        sc /= sc.sum(axis=-1, keepdims=True, dtype=np.float32)
        np.matmul(sc, v[b], out=o[b])
    o = o.transpose(0, 2, 1, 3).reshape(B, T, E)
    return linear(o, p[pre + "out_proj.weight"], p[pre + "out_proj.bias"])


def forward_padding_mask(n_feat: int, padding_mask: np.ndarray) -> np.ndarray:
    """``BEATs.forward_padding_mask`` (beats.py:283-302)."""
    extra = padding_mask.shape[1] % n_feat
    if extra > 0:
        padding_mask = padding_mask[:, :-extra]
    return padding_mask.reshape(padding_mask.shape[0], n_feat, -1).all(-1)


# --------------------------------------------------------------------------------------
# Whole path
# --------------------------------------------------------------------------------------
def layer_names(cfg: Mapping[str, object]) -> List[str]:
    """Hookable layers in ``named_modules()`` order (beats_model.py:206-227)."""
    names = []
    if int(cfg["embed_dim"]) != int(cfg["encoder_embed_dim"]):
        names.append("backbone.post_extract_proj")
    names += [f"backbone.encoder.layers.{i}.fc2" for i in range(int(cfg["encoder_layers"]))]
    return names


def beats_forward(wav: np.ndarray, sd: Mapping[str, np.ndarray], cfg: Mapping[str, object],
                  padding_mask: Optional[np.ndarray] = None, return_stages: bool = False
                  ) -> Tuple[np.ndarray, Dict[str, np.ndarray]]:
    """``beats_model.Model.forward`` in features mode (beats_model.py:232-264) ->
    ``BEATs.extract_features`` (beats.py:325-382) -> ``TransformerEncoder.extract_features``
    (backbone.py:151-221) -> layers (backbone.py:350-375).

    Returns ``(features[B,T',E], taps)`` where ``taps[name]`` is the raw module output a forward hook
    on ``name`` would capture, already batch-first ``[B,T',E]``; with ``return_stages`` extra
    intermediate activations are included under ``stage.*`` keys.
    """
    p = {k[len("backbone."):] if k.startswith("backbone.") else k: np.asarray(v, np.float32) for k, v in sd.items()}
    E = int(cfg["encoder_embed_dim"]); H = int(cfg["encoder_attention_heads"]); L = int(cfg["encoder_layers"])
    D = int(cfg["embed_dim"]); P = int(cfg["input_patch_size"]); G = int(cfg["conv_pos_groups"])
    alpha = np.float32(math.pow(2.0 * L, 0.25)) if bool(cfg.get("deep_norm", True)) else np.float32(1.0)
    pre_ln = bool(cfg.get("layer_norm_first", False))
    assert not (pre_ln and bool(cfg.get("deep_norm", True))), "deep_norm and layer_norm_first exclude each other (beats.py:275)"
    activation = str(cfg.get("activation_fn", "gelu"))
    taps: Dict[str, np.ndarray] = {}

    fb = beats_preprocess(wav, cfg)                                   # [B,frames,mel]
    B, NF, NM = fb.shape
    mask = None
    if padding_mask is not None:
        mask = forward_padding_mask(NF, np.asarray(padding_mask, bool))   # beats.py:346-347
    # Conv2d(1,D,P,stride P,no bias) == patch GEMM; token order t*(NM/P)+f (beats.py:350-352)
    nt, nf = NF // P, NM // P
    patches = fb[:, :nt * P, :nf * P].reshape(B, nt, P, nf, P).transpose(0, 1, 3, 2, 4).reshape(B, nt * nf, P * P)
    wpe = p["patch_embedding.weight"].reshape(D, P * P)
    feat = (patches @ wpe.T).astype(np.float32)
    if "patch_embedding.bias" in p:
        feat = feat + p["patch_embedding.bias"]
    feat = layer_norm(feat, p["layer_norm.weight"], p["layer_norm.bias"])            # beats.py:353
    T = feat.shape[1]
    if mask is not None:
        mask = forward_padding_mask(T, mask)                                          # beats.py:355-356
    if return_stages:
        taps["stage.fbank"] = fb; taps["stage.patch_ln"] = feat
    if "post_extract_proj.weight" in p:
        feat = linear(feat, p["post_extract_proj.weight"], p["post_extract_proj.bias"])   # beats.py:358-359
        taps["backbone.post_extract_proj"] = feat.copy()
    x = feat
    if mask is not None:
        x = np.where(mask[..., None], np.float32(0.0), x)                              # backbone.py:169-170
    wpc = pos_conv_weight(p["encoder.pos_conv.0.parametrizations.weight.original0"],
                          p["encoder.pos_conv.0.parametrizations.weight.original1"])
    xc = pos_conv(x, wpc, p["encoder.pos_conv.0.bias"], G)                             # backbone.py:172-173
    x = x + xc                                                                         # backbone.py:174
    if not pre_ln:
        x = layer_norm(x, p["encoder.layer_norm.weight"], p["encoder.layer_norm.bias"])    # backbone.py:176-177
    if return_stages:
        taps["stage.pos_conv"] = xc; taps["stage.enc_in"] = x

    bias_hTT = None
    if bool(cfg.get("relative_position_embedding", True)):
        bias_hTT = position_bias(p["encoder.layers.0.self_attn.relative_attention_bias.weight"], T,
                                 int(cfg["num_buckets"]), int(cfg["max_distance"]))
    for i in range(L):
        pre = f"encoder.layers.{i}."
        if pre_ln:                                                                      # backbone.py:328-348 (no DeepNorm scale here)
            xn = layer_norm(x, p[pre + "self_attn_layer_norm.weight"], p[pre + "self_attn_layer_norm.bias"])
            a = attention(xn, p, pre + "self_attn.", H, bias_hTT, bool(cfg.get("gru_rel_pos", True)), mask)
            x = x + a
            xn = layer_norm(x, p[pre + "final_layer_norm.weight"], p[pre + "final_layer_norm.bias"])
            y = linear(ffn_hidden(xn, p, pre, activation), p[pre + "fc2.weight"], p[pre + "fc2.bias"])
            taps[f"backbone.encoder.layers.{i}.fc2"] = y.copy()
            x = x + y
            if return_stages:
                taps[f"stage.attn{i}"] = a; taps[f"stage.layer{i}"] = x
            continue
        a = attention(x, p, pre + "self_attn.", H, bias_hTT, bool(cfg.get("gru_rel_pos", True)), mask)
        x = x * alpha + a                                                               # backbone.py:360
        x = layer_norm(x, p[pre + "self_attn_layer_norm.weight"], p[pre + "self_attn_layer_norm.bias"])
        h = ffn_hidden(x, p, pre, activation)                                           # backbone.py:365-368
        y = linear(h, p[pre + "fc2.weight"], p[pre + "fc2.bias"])                       # backbone.py:370 (hook tap)
        taps[f"backbone.encoder.layers.{i}.fc2"] = y.copy()
        x = x * alpha + y                                                               # backbone.py:372
        x = layer_norm(x, p[pre + "final_layer_norm.weight"], p[pre + "final_layer_norm.bias"])
        if return_stages:
            taps[f"stage.attn{i}"] = a; taps[f"stage.layer{i}"] = x
    if pre_ln:      # TransformerEncoder.forward, which BEATs.extract_features calls (beats.py:362-365), normalises after the stack
        x = layer_norm(x, p["encoder.layer_norm.weight"], p["encoder.layer_norm.bias"])    # backbone.py:146-147
    return x, taps


def aggregate(emb: np.ndarray, how: str) -> np.ndarray:
    """Per-layer aggregation over dim 1 (beats_model.py:403-417)."""
    if how == "mean":
        return emb.mean(axis=1, dtype=np.float32)
    if how == "max":
        return emb.max(axis=1)
    if how == "cls_token":
        return emb[:, 0, :]
    raise ValueError(f"Unsupported aggregation method: {how}")


def extract_embeddings(wav: np.ndarray, sd: Mapping[str, np.ndarray], cfg: Mapping[str, object],
                       hook_layers: Sequence[str], aggregation: str = "mean",
                       padding_mask: Optional[np.ndarray] = None):
    """``beats_model.Model.extract_embeddings`` (beats_model.py:279-429) for resolved layer names."""
    _, taps = beats_forward(wav, sd, cfg, padding_mask)
    embs = [taps[n] for n in hook_layers]
    if aggregation == "none":
        return embs[0] if len(embs) == 1 else embs
    embs = [aggregate(e, aggregation) for e in embs]
    return embs[0] if len(embs) == 1 else np.concatenate(embs, axis=1)


def pooled(features: np.ndarray) -> np.ndarray:
    """Headline embedding: ``features.mean(dim=1)`` (README:80; regression test :234)."""
    return features.mean(axis=1, dtype=np.float32)
