// GymWrapper.update_ref / generate_new_ref (reference exciting_environments/gym_wrapper.py:170-192) as one kernel, one thread
// per environment: where the hold counter reached zero, draw a random initial state from the environment's key
// (init_state(rng): e.g. pendulum_env.py:270-276, PMSM pmsm_env.py:402-456), copy its controlled fields into the reference,
// split the key once more for the new hold time (jax.random.randint) and keep the other half as the new key; then count down.
// The samplers restate JAX's published algorithms exactly like random.py (its host twin — same split tree, same bit
// manipulation; parity with JAX itself is unpinned): threefry2x32, split, random_bits, uniform, randint, normal (erf_inv),
// exponential, gamma (Marsaglia-Tsang rejection with the alpha < 1 boost), rademacher, generalized_normal, ball.
#pragma once
#include "models.hpp"

namespace excenv {

struct Key {
  uint32_t k0, k1;
};

__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

// Threefry-2x32, 20 rounds (Salmon et al., SC'11) — same schedule as random.threefry2x32
__device__ __forceinline__ void threefry2x32(Key key, uint32_t c0, uint32_t c1, uint32_t& o0, uint32_t& o1) {
  const uint32_t ks[3] = {key.k0, key.k1, key.k0 ^ key.k1 ^ 0x1BD11BDAu};
  uint32_t x0 = c0 + ks[0], x1 = c1 + ks[1];
  constexpr int R[2][4] = {{13, 15, 26, 6}, {17, 29, 16, 24}};
#pragma unroll
  for (int i = 0; i < 5; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      x0 += x1;
      x1 = rotl32(x1, R[i % 2][j]) ^ x0;
    }
    x0 += ks[(i + 1) % 3];
    x1 += ks[(i + 2) % 3] + (uint32_t)(i + 1);
  }
  o0 = x0;
  o1 = x1;
}

// jax.random.split(key, n)[i]
__device__ __forceinline__ Key rng_split(Key key, uint32_t i) {
  Key r;
  threefry2x32(key, 0u, i, r.k0, r.k1);
  return r;
}
__device__ __forceinline__ uint32_t rng_bits32(Key key, uint32_t i) {
  uint32_t a, b;
  threefry2x32(key, 0u, i, a, b);
  return a ^ b;
}
__device__ __forceinline__ uint64_t rng_bits64(Key key, uint32_t i) {
  uint32_t a, b;
  threefry2x32(key, 0u, i, a, b);
  return ((uint64_t)a << 32) | b;
}

// jax.random.uniform(key, (n,), dtype, minval, maxval)[i]
__device__ __forceinline__ float rng_uniform(Key key, uint32_t i, float lo, float hi) {
  const float u = __uint_as_float((rng_bits32(key, i) >> 9) | 0x3F800000u) - 1.0f;
  const float v = u * (hi - lo) + lo;
  return (v > lo) ? v : lo;
}
__device__ __forceinline__ double rng_uniform(Key key, uint32_t i, double lo, double hi) {
  const double u = __longlong_as_double((long long)((rng_bits64(key, i) >> 12) | 0x3FF0000000000000ull)) - 1.0;
  const double v = u * (hi - lo) + lo;
  return (v > lo) ? v : lo;
}

// jax.random.randint(key, (1,), lo, hi)[0], int32 form (two 32-bit draws, uint32 wrap-around arithmetic)
__device__ __forceinline__ int64_t rng_randint(Key key, int32_t lo, int32_t hi) {
  const uint32_t higher = rng_bits32(rng_split(key, 0), 0), lower = rng_bits32(rng_split(key, 1), 0);
  uint32_t span = (uint32_t)(hi - lo);
  if (hi <= lo) span = 1u;
  uint32_t mult = 65536u % span;
  mult = (mult * mult) % span;
  const uint32_t off = ((higher % span) * mult + (lower % span)) % span;
  return (int64_t)lo + (int64_t)off;
}

// the same under jax_enable_x64 (int64 form: two 64-bit draws, uint64 wrap-around arithmetic) — what the reference draws when its
// arrays are float64 (x64 is the only way to get them, and it makes int64 the default int type); T selects the form
__device__ __forceinline__ int64_t rng_randint64(Key key, int64_t lo, int64_t hi) {
  const uint64_t higher = rng_bits64(rng_split(key, 0), 0), lower = rng_bits64(rng_split(key, 1), 0);
  uint64_t span = (uint64_t)hi - (uint64_t)lo;
  if (hi <= lo) span = 1u;
  uint64_t mult = ((uint64_t)1 << 32) % span;
  mult = (mult * mult) % span;
  const uint64_t off = ((higher % span) * mult + (lower % span)) % span;
  return (int64_t)((uint64_t)lo + off);
}

__device__ __forceinline__ float xerfinv(float x) { return ::erfinvf(x); }
__device__ __forceinline__ double xerfinv(double x) { return ::erfinv(x); }
__device__ __forceinline__ float xlog(float x) { return ::logf(x); }
__device__ __forceinline__ double xlog(double x) { return ::log(x); }
__device__ __forceinline__ float xlog1p(float x) { return ::log1pf(x); }
__device__ __forceinline__ double xlog1p(double x) { return ::log1p(x); }
__device__ __forceinline__ float xpow(float x, float y) { return ::powf(x, y); }
__device__ __forceinline__ double xpow(double x, double y) { return ::pow(x, y); }
__device__ __forceinline__ float next_after_minus_one(float) { return -0.99999994f; }           // nextafter(-1, 0)
__device__ __forceinline__ double next_after_minus_one(double) { return -0.99999999999999989; }  // nextafter(-1, 0)

// jax.random.normal(key, ()): sqrt(2) * erf_inv(uniform(key, (), nextafter(-1, 0), 1))
template <typename T> __device__ __forceinline__ T rng_normal(Key key) {
  const T u = rng_uniform(key, 0, next_after_minus_one(T(0)), T(1));
  return xerfinv(u) * xsqrt(T(2));
}
// jax.random.exponential(key, ())
template <typename T> __device__ __forceinline__ T rng_exponential(Key key) { return -xlog1p(-rng_uniform(key, 0, T(0), T(1))); }

// random._gamma_one (Marsaglia & Tsang with the alpha < 1 boost), the nested while_loops as plain loops
template <typename T> __device__ T rng_gamma_one(Key key, T alpha_orig) {
  const T one = T(1), third = T(1.0 / 3.0);
  const bool boost_mask = alpha_orig >= one;
  const T alpha = boost_mask ? alpha_orig : alpha_orig + one;
  const T d = alpha - third;
  const T c = third / xsqrt(d);
  Key k = rng_split(key, 0);
  const Key subkey = rng_split(key, 1);
  T X = T(0), V = one, U = T(2);
  for (int it = 0; it < 200; ++it) {  // the acceptance test fails a few percent of the time; 200 rounds never bind
    const bool again = (U >= one - T(0.0331) * (X * X)) && (xlog(U) >= X * T(0.5) + d * ((one - V) + xlog(V)));
    if (!again) break;
    const Key k_next = rng_split(k, 0);
    Key kk = rng_split(k, 1);
    const Key u_key = rng_split(k, 2);
    T x = T(0), v = T(-1);
    for (int jt = 0; jt < 200 && v <= T(0); ++jt) {
      const Key sub = rng_split(kk, 1);
      kk = rng_split(kk, 0);
      x = rng_normal<T>(sub);
      v = one + x * c;
    }
    X = x * x;
    V = v * v * v;
    U = rng_uniform(u_key, 0, T(0), T(1));
    k = k_next;
  }
  const T samples = one - rng_uniform(subkey, 0, T(0), T(1));
  const T boost = boost_mask ? one : xpow(samples, one / alpha_orig);
  return d * V * boost;
}

// jax.random.ball(key, 2, p = 2): a point uniform in the unit disc
template <typename T> __device__ void rng_ball2(Key key, T (&out)[2]) {
  const Key k1 = rng_split(key, 0), k2 = rng_split(key, 1);
  // generalized_normal(k1, 2, (2,)) = rademacher(keys[1], (2,)) * gamma(keys[0], 1/2, (2,)) ** (1/2)
  const Key kg = rng_split(k1, 0), kr = rng_split(k1, 1);
  T g[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const T gam = rng_gamma_one<T>(rng_split(kg, j), T(0.5));
    const T r = (rng_uniform(kr, j, T(0), T(1)) < T(0.5)) ? T(1) : T(-1);
    g[j] = r * xsqrt(gam);
  }
  const T e = rng_exponential<T>(k2);
  const T nrm = xsqrt(g[0] * g[0] + g[1] * g[1] + e);
  out[0] = g[0] / nrm;
  out[1] = g[1] / nrm;
}

// init_state(key) of one environment (e.g. pendulum_env.py:270-276; PMSM pmsm_env.py:402-456): physical state + the key
// that becomes the state's PRNGKey leaf.
template <class M, typename T> __device__ __forceinline__ void init_state_from_key(Key key, const Ctx<T, M>& c, T (&phys)[M::S], Key& leaf) {
  constexpr int S = M::S;
  if constexpr (M::IS_PMSM) {
    const Key s1a = rng_split(key, 0), s1b = rng_split(key, 1);
    const T sn0 = rng_uniform(s1b, 0, T(-1), T(1)), sn1 = rng_uniform(s1b, 1, T(-1), T(1));
    const Key s2b = rng_split(s1a, 1);
    leaf = rng_split(s1a, 0);
    T disc[2];
    rng_ball2<T>(s2b, disc);
    T i_max = xabs(c.smin[3]);
    i_max = (xabs(c.smax[3]) > i_max) ? xabs(c.smax[3]) : i_max;
    i_max = (xabs(c.smin[4]) > i_max) ? xabs(c.smin[4]) : i_max;
    i_max = (xabs(c.smax[4]) > i_max) ? xabs(c.smax[4]) : i_max;
    auto relu = [](T x) { return (x > T(0)) ? x : T(0); };
    const T xd = disc[0] * i_max, xq = disc[1] * i_max;
    const T i_d = xd - T(2) * relu(xd - c.smax[3]) + T(2) * relu(-xd + c.smin[3]);
    const T i_q = xq - T(2) * relu(xq - c.smax[4]) + T(2) * relu(-xq + c.smin[4]);
    phys[0] = T(0);
    phys[1] = T(0);
    phys[2] = (sn0 + T(1)) / T(2) * (c.smax[2] - c.smin[2]) + c.smin[2];
    phys[3] = i_d;
    phys[4] = i_q;
    phys[5] = M::torque(i_d, i_q, c);
    phys[6] = (sn1 + T(1)) / T(2) * (c.smax[6] - c.smin[6]) + c.smin[6];
  } else {  // uniform(key, (S,), -1 (tank: 0), 1) denormalised; PRNGKey leaf = split(key)[1]
    const T lo = (M::ID == EXCENV_FLUID_TANK) ? T(0) : T(-1);
#pragma unroll
    for (int j = 0; j < S; ++j) phys[j] = denormalize(rng_uniform(key, j, lo, T(1)), c.smin[j], c.smax[j]);
    leaf = rng_split(key, 1);
  }
}

template <typename T, class M> struct RefGenArgs {
  KProps<T, M> kp;
  int64_t B;
  int32_t n_control;
  int32_t control_idx[EXCENV_MAX_CONTROL];
  T* reference[EXCENV_MAX_CONTROL];  // out: [B] reference leaf of controlled field j
  int64_t* keys;                     // out: [B][2] uint32 key words held in int64 (the Python mirror's key tensors)
  int64_t* hold;                     // out: [B] hold counters
  // inputs (== the outputs for the in-place form)
  const T* reference_in[EXCENV_MAX_CONTROL];
  const int64_t* keys_in;
  const int64_t* hold_in;
  int32_t hold_min, hold_max;
};

template <class M, typename T> __global__ void __launch_bounds__(BLOCK) update_ref_kernel(const RefGenArgs<T, M> ka) {
  constexpr int S = M::S;
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= ka.B) return;
  int64_t h = ka.hold_in[i];
  const bool copy = ka.keys != ka.keys_in;  // out-of-place: environments that are not due carry their values over
  if (h == 0) {
    Ctx<T, M> c;
    load_ctx<true, T, M, false>(c, ka.kp, i, T(0), T(0), T(0));
    Key key{(uint32_t)ka.keys_in[2 * i], (uint32_t)ka.keys_in[2 * i + 1]};
    T phys[S];
    Key leaf;
    init_state_from_key<M, T>(key, c, phys, leaf);
#pragma unroll
    for (int j = 0; j < EXCENV_MAX_CONTROL; ++j) {
      if (j < ka.n_control) {
        const int f = ka.control_idx[j];
        T v = phys[0];
#pragma unroll
        for (int q = 1; q < S; ++q) v = (f == q) ? phys[q] : v;
        ka.reference[j][i] = v;
      }
    }
    const Key k_new = rng_split(leaf, 0), sub = rng_split(leaf, 1);
    h = (sizeof(T) == 8) ? rng_randint64(sub, ka.hold_min, ka.hold_max) : rng_randint(sub, ka.hold_min, ka.hold_max);
    ka.keys[2 * i] = (int64_t)k_new.k0;
    ka.keys[2 * i + 1] = (int64_t)k_new.k1;
  } else if (copy) {
#pragma unroll
    for (int j = 0; j < EXCENV_MAX_CONTROL; ++j)
      if (j < ka.n_control) ka.reference[j][i] = ka.reference_in[j][i];
    ka.keys[2 * i] = ka.keys_in[2 * i];
    ka.keys[2 * i + 1] = ka.keys_in[2 * i + 1];
  }
  ka.hold[i] = h - 1;
}

// CoreEnvironment.vmap_init_state(rng) with one key per environment: physical state leaves + PRNGKey leaf
template <typename T, class M> struct RandomStateArgs {
  KProps<T, M> kp;
  int64_t B;
  const int64_t* keys;  // [B][2]
  T* state_out[M::S];
  int64_t* key_leaf;    // [B][2]
};

template <class M, typename T> __global__ void __launch_bounds__(BLOCK) random_state_kernel(const RandomStateArgs<T, M> ka) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= ka.B) return;
  Ctx<T, M> c;
  load_ctx<true, T, M, false>(c, ka.kp, i, T(0), T(0), T(0));
  const Key key{(uint32_t)ka.keys[2 * i], (uint32_t)ka.keys[2 * i + 1]};
  T phys[M::S];
  Key leaf;
  init_state_from_key<M, T>(key, c, phys, leaf);
#pragma unroll
  for (int j = 0; j < M::S; ++j) ka.state_out[j][i] = phys[j];
  ka.key_leaf[2 * i] = (int64_t)leaf.k0;
  ka.key_leaf[2 * i + 1] = (int64_t)leaf.k1;
}

}  // namespace excenv
