"""Deterministic, platform-independent parameter fill (integer hash -> fp32), TEST INFRASTRUCTURE.

Golden fixtures only store *outputs*; the weights are regenerated bit-identically wherever the
test runs (build container with the real reference, GPU box with oracle + HIP model) from the
parameter NAME and element index, using exact uint64 arithmetic (no libm, no RNG state).
"""
import zlib

import numpy as np
import torch


def _hash_uniform(name: str, n: int) -> np.ndarray:
    """n values in [-1, 1), exactly representable in fp32 (24-bit mantissa grid)."""
    seed = np.uint64(zlib.crc32(name.encode()) + 0x9E3779B1)
    i = np.arange(n, dtype=np.uint64)
    x = (i + seed) * np.uint64(0x9E3779B97F4A7C15)
    x ^= x >> np.uint64(29)
    x *= np.uint64(0xBF58476D1CE4E5B9)
    x ^= x >> np.uint64(32)
    top = (x >> np.uint64(40)).astype(np.int64)          # 24 bits
    return (top.astype(np.float64) / float(1 << 23) - 1.0).astype(np.float32)


@torch.no_grad()
def fill_state(module: torch.nn.Module, gain: float = 1.0):
    """Fill every parameter: conv weights ~ U(-a, a), a = gain*sqrt(3/fan_in) ; norm weight 1 + 0.1u ;
    biases / norm bias 0.1u ; regressor scales 1 + 0.1u. Keyed by state-dict name."""
    for name, p in module.state_dict().items():
        u = torch.from_numpy(_hash_uniform(name, p.numel())).reshape(p.shape)
        if p.ndim == 5:
            fan_in = p[0].numel() if "up." not in name else p.shape[0] * p[0, 0].numel()
            v = u * float(gain * (3.0 / fan_in) ** 0.5)
        elif name.endswith("norm.weight") or name.endswith(".scale"):
            v = 1.0 + 0.1 * u
        else:
            v = 0.1 * u
        p.copy_(v.to(p.dtype))
    return module
