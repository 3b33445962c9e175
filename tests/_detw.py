"""Portable deterministic pseudo-random arrays (exact integer hashing, no RNG
library state) so golden fixtures need not store weights."""
import numpy as np


def det_array(tag: int, shape, scale: float = 0.05) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over='ignore'):
        x = np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + \
            np.uint64((tag + 1) * 0xD1B54A32D192ED03 % (1 << 64))
        x ^= x >> np.uint64(30); x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27); x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    u = (x >> np.uint64(40)).astype(np.float64) / float(1 << 24)      # [0,1)
    return ((u * 2.0 - 1.0) * scale * 1.7320508).astype(np.float32).reshape(shape)


def fill_state_dict(module, tag0: int = 0, scale: float = 0.05):
    """Deterministically overwrite every parameter (sorted-name order)."""
    import torch
    sd = module.state_dict()
    for i, k in enumerate(sorted(sd.keys())):
        v = sd[k]
        if not v.dtype.is_floating_point:
            continue
        a = det_array(tag0 + i, tuple(v.shape), scale)
        if k.endswith('running_var'):
            a = np.abs(a) + 1.0
        if ('norm' in k or '.0.' in k or '.3.' in k) and k.endswith('weight') and v.dim() == 1:
            a = a + 1.0          # BN / LN gains around 1
        sd[k] = torch.from_numpy(a.copy())
    module.load_state_dict(sd)
    return module
