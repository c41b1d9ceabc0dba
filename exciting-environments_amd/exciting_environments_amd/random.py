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


# ---------------------------------------------------------------------------------------------------------------------
# jax.random.randint / ball and the samplers ball is built from (normal, exponential, gamma, rademacher), restated from
# JAX's published source (jax/_src/random.py) as recalled — PARITY UNPINNED like the wiring above: no JAX is available to
# check a single value against. What IS guaranteed by construction: every function consumes exactly the key it is given
# (no hidden state), so the key STREAM of an environment (state.PRNGKey after a random reset, the GymWrapper's reference
# generator) advances through the same split() tree as the reference's. All functions are vectorised over a leading batch
# of keys [..., 2] (the reference vmaps the same scalar code over the batch).
def _u32(x):
    return x & _M32


def _umod64(x: torch.Tensor, span: int) -> torch.Tensor:
    """x mod span for uint64 bit patterns held in int64 tensors (torch has no uint64 arithmetic): a negative int64 s stands for
    s + 2**64, so its residue is (s mod span + 2**64 mod span) mod span."""
    r = torch.remainder(x, span)
    return torch.where(x < 0, torch.remainder(r + ((1 << 64) % span), span), r)


def randint(key: torch.Tensor, n: int, minval: int, maxval: int, x64: bool = False) -> torch.Tensor:
    """jax.random.randint(key, (n,), minval, maxval) for every key of a [..., 2] batch -> int64 tensor [..., n]. Two draws
    (higher / lower bits from split(key)) reduce the modulo bias (jax/_src/random.py _randint).
    x64=False: the default int32 dtype — 32-bit draws, uint32 wrap-around arithmetic.
    x64=True: what the same call returns under jax_enable_x64 (the reference's tests run that way; gym_wrapper.py:183-188 passes no
    dtype, so the default int type is int64) — 64-bit draws, uint64 arithmetic. Spans below 2**31 (hold-step ranges); parity of
    this form is unpinned (no published value), it is checked against a big-integer restatement in tests/test_oracle_rng.py."""
    minval, maxval = int(minval), int(maxval)
    ks = split(key)
    if x64:
        span = (maxval - minval) if maxval > minval else 1
        if span >= (1 << 31):
            raise NotImplementedError("randint(x64=True): span must be below 2**31")
        higher, lower = random_bits(ks[..., 0, :], n, 64), random_bits(ks[..., 1, :], n, 64)
        multiplier = (1 << 32) % span
        multiplier = (multiplier * multiplier) % span
        off = torch.remainder(_umod64(higher, span) * multiplier + _umod64(lower, span), span)  # < span**2 + span < 2**63: no wrap
        return off + minval
    higher, lower = random_bits(ks[..., 0, :], n, 32), random_bits(ks[..., 1, :], n, 32)
    span = (maxval - minval) & _M32
    if maxval <= minval:
        span = 1
    multiplier = (1 << 16) % span
    multiplier = ((multiplier * multiplier) & _M32) % span
    off = _u32(_u32((higher % span) * multiplier) + (lower % span)) % span
    return off + minval


def _uniform_scalar(key, dtype, minval=0.0, maxval=1.0):
    return uniform(key, 1, dtype, minval, maxval)[..., 0]


def normal(key: torch.Tensor, dtype=torch.float32) -> torch.Tensor:
    """jax.random.normal(key, (), dtype): sqrt(2) * erf_inv(uniform(key, (), minval=nextafter(-1, 0), maxval=1))."""
    lo = torch.nextafter(torch.tensor(-1.0, dtype=dtype), torch.tensor(0.0, dtype=dtype)).item()
    u = _uniform_scalar(key, dtype, lo, 1.0)
    return torch.erfinv(u) * torch.tensor(2.0, dtype=dtype).sqrt().to(u.device)


def exponential(key: torch.Tensor, dtype=torch.float32) -> torch.Tensor:
    """jax.random.exponential(key, (), dtype): -log1p(-uniform(key))."""
    return -torch.log1p(-_uniform_scalar(key, dtype))


def rademacher(key: torch.Tensor, n: int, dtype=torch.float32) -> torch.Tensor:
    """jax.random.rademacher(key, (n,), dtype): 2 * bernoulli(key, 0.5, (n,)) - 1, bernoulli = uniform(key, (n,)) < p."""
    return (uniform(key, n, dtype) < 0.5).to(dtype) * 2 - 1


def _gamma_one(key: torch.Tensor, alpha: float, dtype) -> torch.Tensor:
    """random._gamma_one (Marsaglia & Tsang, with the alpha < 1 boost) for a batch of keys [M, 2], scalar alpha. The two
    nested lax.while_loops of the source become masked iterations: a lane keeps its values once its own loop has ended."""
    dev = key.device
    t = lambda v: torch.tensor(v, dtype=dtype, device=dev)
    one, third = t(1.0), t(1.0 / 3.0)
    boost_mask = alpha >= 1.0
    a = t(alpha) if boost_mask else t(alpha) + one
    d = a - third
    c = third / torch.sqrt(d)
    sp = split(key)
    k, subkey = sp[:, 0, :], sp[:, 1, :]
    M = key.shape[0]
    X, V, U = torch.zeros(M, dtype=dtype, device=dev), torch.ones(M, dtype=dtype, device=dev), torch.full((M,), 2.0, dtype=dtype, device=dev)
    active = torch.ones(M, dtype=torch.bool, device=dev)
    for _ in range(200):  # P(reject) per round is a few percent; 200 rounds never bind in practice
        if not bool(active.any()):
            break
        s3 = split(k, 3)
        k_next, kk, u_key = s3[:, 0, :], s3[:, 1, :], s3[:, 2, :]
        x, v = torch.zeros(M, dtype=dtype, device=dev), torch.full((M,), -1.0, dtype=dtype, device=dev)
        inner = active.clone()
        for _ in range(200):
            if not bool(inner.any()):
                break
            s2 = split(kk)
            xn = normal(s2[:, 1, :], dtype)
            vn = one + xn * c
            kk = torch.where(inner[:, None], s2[:, 0, :], kk)
            x, v = torch.where(inner, xn, x), torch.where(inner, vn, v)
            inner = inner & (v <= 0)
        Xn, Vn, Un = x * x, v * v * v, _uniform_scalar(u_key, dtype)
        k = torch.where(active[:, None], k_next, k)
        X, V, U = torch.where(active, Xn, X), torch.where(active, Vn, V), torch.where(active, Un, U)
        cond = (U >= one - t(0.0331) * (X * X)) & (torch.log(U) >= X * t(0.5) + d * ((one - V) + torch.log(V)))
        active = active & cond
    samples = one - _uniform_scalar(subkey, dtype)
    boost = torch.ones(M, dtype=dtype, device=dev) if boost_mask else torch.pow(samples, one / t(alpha))
    return d * V * boost


def gamma(key: torch.Tensor, alpha: float, n: int, dtype=torch.float32) -> torch.Tensor:
    """jax.random.gamma(key, alpha, (n,), dtype): one sub-key per sample (split(key, n)), each through _gamma_one."""
    key = torch.as_tensor(key).to(torch.int64)
    lead = key.shape[:-1]
    keys = split(key, n).reshape(-1, 2)
    return _gamma_one(keys, float(alpha), dtype).reshape(tuple(lead) + (n,))


def generalized_normal(key: torch.Tensor, p: float, n: int, dtype=torch.float32) -> torch.Tensor:
    """jax.random.generalized_normal(key, p, (n,), dtype): rademacher * gamma(1/p) ** (1/p) from split(key)."""
    ks = split(key)
    g = gamma(ks[..., 0, :], 1.0 / p, n, dtype)
    r = rademacher(ks[..., 1, :], n, dtype)
    return r * torch.pow(g, torch.tensor(1.0 / p, dtype=dtype, device=g.device))


def ball(key: torch.Tensor, d: int, p: float = 2, dtype=torch.float32) -> torch.Tensor:
    """jax.random.ball(key, d, p, (), dtype): a point uniform in the unit p-ball, [..., d] for a [..., 2] batch of keys
    (the PMSM random current draw, pmsm_env.py:405-406)."""
    ks = split(key)
    g = generalized_normal(ks[..., 0, :], p, d, dtype)
    e = exponential(ks[..., 1, :], dtype)
    inv_p = torch.tensor(1.0 / p, dtype=dtype, device=g.device)
    return g / torch.pow(torch.pow(g.abs(), torch.tensor(float(p), dtype=dtype, device=g.device)).sum(-1) + e, inv_p)[..., None]
