import numpy as np


def rel_l2(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def max_abs(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))))


def round_half(x, dtype):
    """Round an fp32 numpy array to the operand dtype and back (what the MFMA sees)."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(x, np.float32))
    td = torch.float16 if dtype in ("f16", 0) else torch.bfloat16
    return t.to(td).float().numpy()
