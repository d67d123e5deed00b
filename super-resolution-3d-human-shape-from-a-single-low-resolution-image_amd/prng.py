"""Counter-based PRNG shared by the golden generator (this container) and the
GPU box, so that bit-identical synthetic weights / images / points exist on
both sides without committing 95 MB of tensors.

value(name, seed, i) = splitmix64(fnv1a64(name) ^ (seed * GOLDEN) + i); the top
24 bits become a float32 in [0, 1).  Pure integer numpy: identical everywhere.
"""
import numpy as np

_MASK = (1 << 64) - 1
_GOLDEN = 0x9E3779B97F4A7C15


def fnv1a64(name: str) -> int:
    h = 0xCBF29CE484222325
    for b in name.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & _MASK
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = x + np.uint64(_GOLDEN)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(name: str, seed: int, n: int) -> np.ndarray:
    """n float32 values in [0,1), a pure function of (name, seed, index)."""
    key = (fnv1a64(name) ^ ((seed * _GOLDEN) & _MASK)) & _MASK
    with np.errstate(over="ignore"):
        ctr = np.uint64(key) + np.arange(n, dtype=np.uint64)
    bits = _splitmix64(ctr) >> np.uint64(40)
    return (bits.astype(np.float32) * np.float32(1.0 / (1 << 24))).astype(np.float32)


def uniform(name: str, seed: int, shape, lo: float, hi: float) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    u = uniform01(name, seed, n)
    out = (np.float32(lo) + u * np.float32(hi - lo)).astype(np.float32)
    return out.reshape(shape)
