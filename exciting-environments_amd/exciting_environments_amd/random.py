"""JAX-compatible counter-based keys for random resets.

The reference draws initial states with ``jax.random.uniform(key, shape=(S,), minval=-1, maxval=1)`` and keeps
``jax.random.split(key)[1]`` as the state's PRNGKey (e.g. pendulum_env.py:270-276). This module restates the public
algorithm behind those calls for jax==0.9.0 defaults (``threefry2x32`` PRNG, ``jax_threefry_partitionable=True``):

  * threefry2x32, 20 rounds — Salmon et al., "Parallel random numbers: as easy as 1, 2, 3" (SC'11). Pinned by the
    Random123 known-answer vectors (tests/test_random_keys.py).
  * ``PRNGKey(seed)`` = (seed >> 32, seed & 0xffffffff); ``split(key, n)[i]`` = threefry(key, (0, i));
    32-bit ``random_bits`` for element i = hi ^ lo of threefry(key, (0, i)), 64-bit = (hi << 32) | lo;
    ``uniform`` = bitcast((bits >> (nbits - nmant)) | bits_of(1.0)) - 1.0, then ``max(minval, u*(maxval-minval)+minval)``.

JAX is not available in this build environment, so the wiring above is **parity unpinned**: it follows JAX's published
source as recalled, corroborated only by the documented ``split(key(0))`` example. Keys are int64 tensors of shape
[..., 2] holding uint32 words (torch has no general uint32 arithmetic).
"""
from __future__ import annotations

import torch

_M32 = 0xFFFFFFFF
_ROT = ((13, 15, 26, 6), (17, 29, 16, 24))


def _rotl(x, r):
    return ((x << r) | (x >> (32 - r))) & _M32


def threefry2x32(k0, k1, c0, c1):
    """Threefry-2x32-20 block function on int64 tensors (or Python ints) holding uint32 words."""
    as_t = lambda v: v if isinstance(v, torch.Tensor) else torch.tensor(v, dtype=torch.int64)
    k0, k1, c0, c1 = (as_t(v).to(torch.int64) & _M32 for v in (k0, k1, c0, c1))
    ks = (k0, k1, (k0 ^ k1 ^ 0x1BD11BDA) & _M32)
    x0, x1 = (c0 + ks[0]) & _M32, (c1 + ks[1]) & _M32
    for i in range(5):
        for r in _ROT[i % 2]:
            x0 = (x0 + x1) & _M32
            x1 = _rotl(x1, r) ^ x0
        x0 = (x0 + ks[(i + 1) % 3]) & _M32
        x1 = (x1 + ks[(i + 2) % 3] + (i + 1)) & _M32
    return x0, x1


def PRNGKey(seed: int, device=None) -> torch.Tensor:
    """jax.random.PRNGKey(seed) as an int64 [2] tensor of uint32 words."""
    seed = int(seed)
    return torch.tensor([(seed >> 32) & _M32, seed & _M32], dtype=torch.int64, device=device)


def split(key: torch.Tensor, num: int = 2) -> torch.Tensor:
    """jax.random.split(key, num): [..., 2] -> [..., num, 2]."""
    key = torch.as_tensor(key).to(torch.int64)
    idx = torch.arange(num, dtype=torch.int64, device=key.device)
    k0, k1 = key[..., 0:1], key[..., 1:2]
    b0, b1 = threefry2x32(k0, k1, torch.zeros_like(idx), idx)
    return torch.stack([b0, b1], dim=-1)


def random_bits(key: torch.Tensor, n: int, bit_width: int = 32) -> torch.Tensor:
    """n words per key: [..., 2] -> [..., n] (int64 holding uint32 words, or the low 64 bits as int64 for bit_width 64)."""
    key = torch.as_tensor(key).to(torch.int64)
    idx = torch.arange(n, dtype=torch.int64, device=key.device)
    b0, b1 = threefry2x32(key[..., 0:1], key[..., 1:2], torch.zeros_like(idx), idx)
    if bit_width == 32:
        return b0 ^ b1
    if bit_width == 64:
        return (b0 << 32) | b1  # wraps into the sign bit of int64: only the bit pattern matters
    raise ValueError("bit_width must be 32 or 64")


def uniform(key: torch.Tensor, n: int, dtype=torch.float32, minval=0.0, maxval=1.0) -> torch.Tensor:
    """jax.random.uniform(key, (n,), dtype, minval, maxval) for every key of a [..., 2] batch -> [..., n]."""
    if dtype == torch.float32:
        bits = random_bits(key, n, 32)
        fb = ((bits >> 9) | 0x3F800000).to(torch.int32)
        u = fb.view(torch.float32) - 1.0
    elif dtype == torch.float64:
        bits = random_bits(key, n, 64)
        mant = (bits >> 12) & ((1 << 52) - 1)  # logical shift of the 64-bit pattern
        u = (mant | 0x3FF0000000000000).view(torch.float64) - 1.0
    else:
        raise TypeError("uniform: float32 or float64")
    lo = torch.as_tensor(minval, dtype=dtype, device=u.device)
    hi = torch.as_tensor(maxval, dtype=dtype, device=u.device)
    return torch.maximum(lo, u * (hi - lo) + lo)


def is_key(x) -> bool:
    return isinstance(x, torch.Tensor) and x.dtype in (torch.int64, torch.int32, torch.uint32) and x.ndim >= 1 and x.shape[-1] == 2
