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

/* ---- float instantiation ---- */
#define REAL float
#define FN(x) CAT(x, _f32)
#define R_SIN sinf
#define R_COS cosf
#define R_SQRT sqrtf
#define R_ATAN2 atan2f
#define R_FMOD fmodf
#define R_FMA fmaf
#define R_ABS fabsf
#include "oracle_body.inc"
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
  if (semantics != EXCENV_SEM_STEP && semantics != EXCENV_SEM_AHEAD) return EXCENV_EINVAL;
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
