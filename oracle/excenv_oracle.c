/*
 * ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the product path.
 *
 * Plain-C CPU restatement of the reference's batched ODE step path
 * (exciting_environments/core_env.py vmap_step :533-569 / vmap_sim_ahead :571-616 and the six
 * environments' _ode / _ode_solver_step / _ode_solver_simulate_ahead / generate_observation).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library —
 * as the checker / the timed CPU baseline, never as a fallback for the HIP path.
 *
 * Third-party arithmetic restated here: diffrax==0.7.0 (reference pyproject.toml:27) Euler.step and the
 * fixed-step explicit-RK stage loop; jax.numpy remainder / clip / sign semantics (jax==0.9.0).
 *
 * Pinning: Euler, fp64, step path is pinned by the reference's own golden fixtures
 * (tests/envs/ENV/data/ files, copied to tests/golden/) at the reference's tolerances
 * (tests/test_oracle_golden.py). Tsit5 / RK4: "parity unpinned" — no reference test or fixture
 * exists; pinned only by tableau order conditions and convergence order (tests/test_oracle_rk.py).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stddef.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "../include/excenv.h"

typedef struct {
  int id, S, A, O, P;
} env_info_t;

static const env_info_t ENVS[EXCENV_NUM_ENVS] = {
    {EXCENV_PENDULUM, 2, 1, 2, 3},  {EXCENV_MASS_SPRING_DAMPER, 2, 1, 2, 3}, {EXCENV_CART_POLE, 4, 1, 4, 6},
    {EXCENV_ACROBOT, 4, 1, 4, 9},   {EXCENV_FLUID_TANK, 1, 1, 1, 4},         {EXCENV_PMSM, 7, 2, 8, 7},
};

/* Classic RK4 and Tsitouras 5(4) (first six stages; b7 = 0 and the FSAL stage is not needed at fixed
 * step). Coefficients: SURVEY.md Appendix B / Ch. Tsitouras, Comput. Math. Appl. 62 (2011). */
static const double RK4_C[6] = {0.0, 0.5, 0.5, 1.0, 0, 0};
static const double RK4_A[6][6] = {{0}, {0.5}, {0.0, 0.5}, {0.0, 0.0, 1.0}, {0}, {0}};
static const double RK4_B[6] = {1.0 / 6.0, 1.0 / 3.0, 1.0 / 3.0, 1.0 / 6.0, 0, 0};

static const double TSIT5_C[6] = {0.0, 0.161, 0.327, 0.9, 0.9800255409045097, 1.0};
static const double TSIT5_A[6][6] = {
    {0},
    {0.161},
    {-0.008480655492356989, 0.335480655492357},
    {2.8971530571054935, -6.359448489975075, 4.3622954328695815},
    {5.325864828439257, -11.74888356406283, 7.4955393428898365, -0.09249506636175525},
    {5.86145544294642, -12.92096931784711, 8.159367898576159, -0.071584973281401, -0.028269050394068383},
};
static const double TSIT5_B[6] = {0.09646076681806523, 0.01, 0.4798896504144996,
                                  1.379008574103742,   -3.290069515436081, 2.324710524099774};

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)

/* ---- jax.random, the part that does not depend on the working precision (the rest: oracle_rng.inc) ---------------------
 * Threefry-2x32, 20 rounds: Salmon, Moraes, Dror, Shaw, "Parallel random numbers: as easy as 1, 2, 3" (SC'11), the rotation
 * constants and key schedule of Random123's threefry2x32_R(20, ...) == jax/_src/prng.py threefry2x32 (apply_round, rotations
 * [13, 15, 26, 6] / [17, 29, 16, 24], ks[2] = k0 ^ k1 ^ 0x1BD11BDA, five groups of four rounds with a key injection each). */
static inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

static void threefry2x32(const uint32_t key[2], uint32_t c0, uint32_t c1, uint32_t out[2]) {
  static const int R[2][4] = {{13, 15, 26, 6}, {17, 29, 16, 24}};
  const uint32_t ks[3] = {key[0], key[1], key[0] ^ key[1] ^ 0x1BD11BDAu};
  uint32_t x0 = c0 + ks[0], x1 = c1 + ks[1];
  for (int i = 0; i < 5; ++i) {
    for (int j = 0; j < 4; ++j) {
      x0 += x1;
      x1 = rotl32(x1, R[i % 2][j]);
      x1 ^= x0;
    }
    x0 += ks[(i + 1) % 3];
    x1 += ks[(i + 2) % 3] + (uint32_t)(i + 1);
  }
  out[0] = x0;
  out[1] = x1;
}

/* jax.random.split(key, n)[i] — threefry_split, partitionable form: the counter of element i of an iota over the output shape is
 * the 64-bit index split into (high, low) words = (0, i); the new key is the pair of output words */
static inline void rng_split_i(const uint32_t key[2], uint32_t i, uint32_t out[2]) { threefry2x32(key, 0u, i, out); }
/* threefry_random_bits, partitionable form, element i of a shape with fewer than 2^32 elements: 32-bit draws are the XOR of the
 * two output words, 64-bit draws are (word0 << 32) | word1 */
static inline uint32_t rng_bits32(const uint32_t key[2], uint32_t i) {
  uint32_t o[2];
  threefry2x32(key, 0u, i, o);
  return o[0] ^ o[1];
}
static inline uint64_t rng_bits64(const uint32_t key[2], uint32_t i) {
  uint32_t o[2];
  threefry2x32(key, 0u, i, o);
  return ((uint64_t)o[0] << 32) | o[1];
}
/* jax.random.randint(key, (n,), minval, maxval)[i], default int32 dtype (random._randint): two 32-bit draws from the halves of
 * split(key), span / multiplier arithmetic in uint32 with wrap-around */
static int32_t rng_randint_i32(const uint32_t key[2], uint32_t i, int32_t minval, int32_t maxval) {
  uint32_t k1[2], k2[2];
  rng_split_i(key, 0, k1);
  rng_split_i(key, 1, k2);
  const uint32_t higher = rng_bits32(k1, i), lower = rng_bits32(k2, i);
  uint32_t span = (uint32_t)maxval - (uint32_t)minval;
  if (maxval <= minval) span = 1u;
  uint32_t mult = (1u << 16) % span; /* 2 ** (nbits / 2) % span */
  mult = (mult * mult) % span;       /* == 2 ** nbits % span */
  const uint32_t off = ((higher % span) * mult + (lower % span)) % span;
  return (int32_t)((uint32_t)minval + off);
}
/* The same function as JAX evaluates it with jax_enable_x64 (the reference's own tests run that way, tests/test_gym_wrapper.py;
 * gym_wrapper.py:183-188 passes no dtype, so randint draws the default int type: int64): nbits = 64, two 64-bit draws, span /
 * multiplier arithmetic in uint64 with wrap-around (jax/_src/random.py _randint). PARITY UNPINNED: no published x64 randint value
 * is known to the author; tests/test_oracle_rng.py checks it against an independent big-integer restatement of the same source. */
static int64_t rng_randint_i64(const uint32_t key[2], uint32_t i, int64_t minval, int64_t maxval) {
  uint32_t k1[2], k2[2];
  rng_split_i(key, 0, k1);
  rng_split_i(key, 1, k2);
  const uint64_t higher = rng_bits64(k1, i), lower = rng_bits64(k2, i);
  uint64_t span = (uint64_t)maxval - (uint64_t)minval;
  if (maxval <= minval) span = 1u;
  uint64_t mult = ((uint64_t)1 << 32) % span; /* 2 ** (nbits / 2) % span */
  mult = (mult * mult) % span;                /* == 2 ** nbits % span (uint64 product, wraps like lax.mul) */
  const uint64_t off = ((higher % span) * mult + (lower % span)) % span;
  return (int64_t)((uint64_t)minval + off);
}
/* erf_inv in double precision: Winitzki's closed-form start, then Newton on libm's erf (|y| < 0.5) or on erfc of the tail
 * (no cancellation in 1 - |y|: exact for |y| >= 0.5). */
static double erfinv_d(double y) {
  if (y != y || y <= -1.0 || y >= 1.0) return (y == 1.0) ? INFINITY : ((y == -1.0) ? -INFINITY : NAN);
  if (y == 0.0) return y;
  const double ay = fabs(y), a = 0.147, two_over_sqrt_pi = 1.1283791670955126;
  const double ln1 = log1p(-ay * ay), t = 2.0 / (3.141592653589793 * a) + 0.5 * ln1;
  double x = sqrt(sqrt(t * t - ln1 / a) - t);
  for (int it = 0; it < 8; ++it) {
    const double fx = (ay < 0.5) ? (erf(x) - ay) : ((1.0 - ay) - erfc(x));
    const double step = fx / (two_over_sqrt_pi * exp(-x * x));
    x -= step;
    if (fabs(step) <= 1e-17 * fabs(x)) break;
  }
  return (y < 0) ? -x : x;
}

/* ---- float instantiation ---- */
/* oracle-only semantics value (not part of include/excenv.h: the product offers EXCENV_SEM_STEP / EXCENV_SEM_AHEAD) */
#define ORACLE_SEM_AHEAD_ACCUMULATED_T 2

#define REAL float
#define FN(x) CAT(x, _f32)
#define R_SIN sinf
#define R_COS cosf
#define R_SQRT sqrtf
#define R_ATAN2 atan2f
#define R_FMOD fmodf
#define R_FMA fmaf
#define R_ABS fabsf
#define ORACLE_REAL_IS_FLOAT 1
#include "oracle_body.inc"
#include "oracle_rng.inc"
#undef ORACLE_REAL_IS_FLOAT
#undef REAL
#undef FN
#undef R_SIN
#undef R_COS
#undef R_SQRT
#undef R_ATAN2
#undef R_FMOD
#undef R_FMA
#undef R_ABS

/* ---- double instantiation ---- */
#define REAL double
#define FN(x) CAT(x, _f64)
#define R_SIN sin
#define R_COS cos
#define R_SQRT sqrt
#define R_ATAN2 atan2
#define R_FMOD fmod
#define R_FMA fma
#define R_ABS fabs
#include "oracle_body.inc"
#include "oracle_rng.inc"
#undef REAL
#undef FN

static int check_common(int env, int solver, int dtype, int64_t B) {
  if (env < 0 || env >= EXCENV_NUM_ENVS) return EXCENV_EINVAL;
  if (solver < 0 || solver >= EXCENV_NUM_SOLVERS) return EXCENV_EINVAL;
  if (dtype != EXCENV_F32 && dtype != EXCENV_F64) return EXCENV_EINVAL;
  if (B < 0) return EXCENV_EINVAL;
  return EXCENV_OK;
}

int oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void oracle_set_num_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

int oracle_env_dims(int env, int32_t* S, int32_t* A, int32_t* O, int32_t* P) {
  if (env < 0 || env >= EXCENV_NUM_ENVS) return EXCENV_EINVAL;
  *S = ENVS[env].S; *A = ENVS[env].A; *O = ENVS[env].O; *P = ENVS[env].P;
  return EXCENV_OK;
}

/* host-pointer twin of excenv_step */
int oracle_step(int env, int solver, int dtype, int64_t B, const excenv_props_t* props,
                const excenv_control_t* control, double tau, const void* const* state_in, const void* action,
                void* const* state_out, void* obs) {
  int rc = check_common(env, solver, dtype, B);
  if (rc) return rc;
  if (!props || !state_in || !action || !state_out || !obs) return EXCENV_ENULL;
  if (props->pmsm_lut && env != EXCENV_PMSM) return EXCENV_EINVAL;
  if (control && control->n_control == 0) control = NULL;
  return dtype == EXCENV_F32
             ? oracle_step_f32(&ENVS[env], solver, B, props, control, tau, state_in, action, state_out, obs)
             : oracle_step_f64(&ENVS[env], solver, B, props, control, tau, state_in, action, state_out, obs);
}

/* host-pointer twin of excenv_gym_step */
int oracle_gym_step(int env, int solver, int dtype, int64_t B, const excenv_props_t* props,
                    const excenv_control_t* control, double tau, const void* const* state_in, const void* action,
                    void* const* state_out, void* obs, void* reward, uint8_t* terminated, uint8_t* truncated) {
  int rc = check_common(env, solver, dtype, B);
  if (rc) return rc;
  if (!props || !state_in || !action || !state_out || !obs || !reward || !terminated || !truncated) return EXCENV_ENULL;
  if (control && control->n_control == 0) control = NULL;
  return dtype == EXCENV_F32 ? oracle_gym_step_f32(&ENVS[env], solver, B, props, control, tau, state_in, action, state_out,
                                                   obs, reward, terminated, truncated)
                             : oracle_gym_step_f64(&ENVS[env], solver, B, props, control, tau, state_in, action, state_out,
                                                   obs, reward, terminated, truncated);
}

/* host-pointer twin of excenv_rew_trunc_term (env-major trajectories) */
int oracle_rew_trunc_term(int env, int dtype, int64_t B, int64_t rows, const excenv_props_t* props,
                          const excenv_control_t* control, const void* const* state_traj, void* reward,
                          uint8_t* terminated, uint8_t* truncated) {
  int rc = check_common(env, 0, dtype, B);
  if (rc) return rc;
  if (rows < 0) return EXCENV_EINVAL;
  if (!props || !state_traj || !reward || !terminated || !truncated) return EXCENV_ENULL;
  if (control && control->n_control == 0) control = NULL;
  return dtype == EXCENV_F32
             ? oracle_rew_trunc_term_f32(&ENVS[env], B, rows, props, control, state_traj, reward, terminated, truncated)
             : oracle_rew_trunc_term_f64(&ENVS[env], B, rows, props, control, state_traj, reward, terminated, truncated);
}

/* host-pointer twin of excenv_sim_ahead */
int oracle_sim_ahead(int env, int solver, int dtype, int64_t B, int64_t K, int32_t substeps,
                     const excenv_props_t* props, const excenv_control_t* control, double obs_stepsize,
                     double env_tau, const void* const* state_in, const void* actions, int action_layout,
                     void* obs_traj, void* const* state_traj, int traj_layout, void* const* last_state,
                     int semantics) {
  int rc = check_common(env, solver, dtype, B);
  if (rc) return rc;
  if (K < 0 || substeps < 1) return EXCENV_EINVAL;
  if (env == EXCENV_PMSM && substeps != 1) return EXCENV_EUNSUPPORTED;
  if (semantics != EXCENV_SEM_STEP && semantics != EXCENV_SEM_AHEAD && semantics != ORACLE_SEM_AHEAD_ACCUMULATED_T) return EXCENV_EINVAL;
  if (!props || !state_in || (!actions && K > 0) || !obs_traj || !last_state) return EXCENV_ENULL;
  if (control && control->n_control == 0) control = NULL;
  return dtype == EXCENV_F32
             ? oracle_sim_ahead_f32(&ENVS[env], solver, B, K, substeps, props, control, obs_stepsize, env_tau,
                                    state_in, actions, action_layout, obs_traj, state_traj, traj_layout,
                                    last_state, semantics)
             : oracle_sim_ahead_f64(&ENVS[env], solver, B, K, substeps, props, control, obs_stepsize, env_tau,
                                    state_in, actions, action_layout, obs_traj, state_traj, traj_layout,
                                    last_state, semantics);
}

/* ---- jax.random restated (oracle_rng.inc): exported for the tests -------------------------------------------------------- */
void oracle_threefry2x32(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t* out) {
  const uint32_t key[2] = {k0, k1};
  threefry2x32(key, c0, c1, out);
}

/* keys [n][2] (uint32 words held in int64, like the product's key tensors) -> out [n][num][2] = jax.random.split(key, num) */
void oracle_split(int64_t n, const int64_t* keys, int32_t num, int64_t* out) {
  for (int64_t i = 0; i < n; ++i) {
    const uint32_t key[2] = {(uint32_t)keys[2 * i], (uint32_t)keys[2 * i + 1]};
    for (int32_t j = 0; j < num; ++j) {
      uint32_t o[2];
      rng_split_i(key, (uint32_t)j, o);
      out[(i * num + j) * 2] = (int64_t)o[0];
      out[(i * num + j) * 2 + 1] = (int64_t)o[1];
    }
  }
}

/* out [n][m] = jax.random.bits(key, (m,)) with 32-bit (uint32 in int64) or 64-bit (bit pattern in int64) words */
void oracle_random_bits(int64_t n, const int64_t* keys, int32_t m, int32_t bit_width, int64_t* out) {
  for (int64_t i = 0; i < n; ++i) {
    const uint32_t key[2] = {(uint32_t)keys[2 * i], (uint32_t)keys[2 * i + 1]};
    for (int32_t j = 0; j < m; ++j)
      out[i * m + j] = (bit_width == 64) ? (int64_t)rng_bits64(key, (uint32_t)j) : (int64_t)rng_bits32(key, (uint32_t)j);
  }
}

/* out [n][m] = jax.random.randint(key, (m,), minval, maxval), int32 form */
void oracle_randint(int64_t n, const int64_t* keys, int32_t m, int32_t minval, int32_t maxval, int64_t* out) {
  for (int64_t i = 0; i < n; ++i) {
    const uint32_t key[2] = {(uint32_t)keys[2 * i], (uint32_t)keys[2 * i + 1]};
    for (int32_t j = 0; j < m; ++j) out[i * m + j] = (int64_t)rng_randint_i32(key, (uint32_t)j, minval, maxval);
  }
}

/* out [n][m] = jax.random.randint(key, (m,), minval, maxval) under jax_enable_x64 (int64 form) */
void oracle_randint64(int64_t n, const int64_t* keys, int32_t m, int64_t minval, int64_t maxval, int64_t* out) {
  for (int64_t i = 0; i < n; ++i) {
    const uint32_t key[2] = {(uint32_t)keys[2 * i], (uint32_t)keys[2 * i + 1]};
    for (int32_t j = 0; j < m; ++j) out[i * m + j] = rng_randint_i64(key, (uint32_t)j, minval, maxval);
  }
}

double oracle_erfinv(double y) { return erfinv_d(y); }

/* which: 0 uniform(key, (m,), lo, hi) -> [n][m]; 1 normal(key, ()) -> [n]; 2 exponential(key, ()) -> [n];
 * 3 gamma(key, lo, (m,)) -> [n][m]; 4 ball(key, 2) -> [n][2] */
int oracle_rng_sample(int which, int dtype, int64_t n, const int64_t* keys, int32_t m, double lo, double hi, void* out) {
  if (which < 0 || which > 4 || (dtype != EXCENV_F32 && dtype != EXCENV_F64) || n < 0 || !keys || !out) return EXCENV_EINVAL;
  if (dtype == EXCENV_F32) oracle_rng_sample_f32(which, n, keys, m, lo, hi, (float*)out);
  else oracle_rng_sample_f64(which, n, keys, m, lo, hi, (double*)out);
  return EXCENV_OK;
}

/* host-pointer twin of excenv_random_state */
int oracle_random_state(int env, int dtype, int64_t B, const excenv_props_t* props, const int64_t* keys, void* const* state_out,
                        int64_t* key_leaf) {
  int rc = check_common(env, 0, dtype, B);
  if (rc) return rc;
  if (!props || !keys || !state_out || !key_leaf) return EXCENV_ENULL;
  return dtype == EXCENV_F32 ? oracle_random_state_f32(&ENVS[env], B, props, keys, state_out, key_leaf)
                             : oracle_random_state_f64(&ENVS[env], B, props, keys, state_out, key_leaf);
}

/* host-pointer twin of excenv_update_ref_to */
int oracle_update_ref(int env, int dtype, int64_t B, const excenv_props_t* props, int32_t n_control, const int32_t* control_idx,
                      const void* const* reference_in, const int64_t* keys_in, const int64_t* hold_in, void* const* reference_out,
                      int64_t* keys_out, int64_t* hold_out, int32_t hold_min, int32_t hold_max) {
  int rc = check_common(env, 0, dtype, B);
  if (rc) return rc;
  if (n_control < 0 || n_control > EXCENV_MAX_CONTROL) return EXCENV_EINVAL;
  if (!props || !keys_in || !hold_in || !keys_out || !hold_out || (n_control > 0 && (!control_idx || !reference_in || !reference_out)))
    return EXCENV_ENULL;
  return dtype == EXCENV_F32 ? oracle_update_ref_f32(&ENVS[env], B, props, n_control, control_idx, reference_in, keys_in, hold_in,
                                                     reference_out, keys_out, hold_out, hold_min, hold_max)
                             : oracle_update_ref_f64(&ENVS[env], B, props, n_control, control_idx, reference_in, keys_in, hold_in,
                                                     reference_out, keys_out, hold_out, hold_min, hold_max);
}
