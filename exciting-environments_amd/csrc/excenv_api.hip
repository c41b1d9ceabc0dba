// extern "C" entry points of libexcenv_hip.so (declared in include/excenv.h): argument validation,
// per-thread error string, dispatch into the per-environment launch tables. No device allocation,
// no synchronisation — only kernel enqueues on the caller's stream.
#include <cstdarg>
#include <cstdio>
#include "launch.hpp"

namespace excenv {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

EnvVTable vtable_pendulum();
EnvVTable vtable_msd();
EnvVTable vtable_cartpole();
EnvVTable vtable_acrobot();
EnvVTable vtable_tank();
EnvVTable vtable_pmsm();

static const EnvVTable* table(int env) {
  static const EnvVTable T[EXCENV_NUM_ENVS] = {vtable_pendulum(), vtable_msd(),  vtable_cartpole(),
                                               vtable_acrobot(),  vtable_tank(), vtable_pmsm()};
  if (env < 0 || env >= EXCENV_NUM_ENVS) return nullptr;
  return &T[env];
}

static int g_vec_pref = 0;
static int g_lds_pad = 0;

static int check_common(const char* fn, int env, int solver, int dtype, int64_t B) {
  if (!table(env)) { set_error("%s: bad env id %d", fn, env); return EXCENV_EINVAL; }
  if (solver < 0 || solver >= EXCENV_NUM_SOLVERS) { set_error("%s: bad solver id %d", fn, solver); return EXCENV_EINVAL; }
  if (dtype != EXCENV_F32 && dtype != EXCENV_F64) { set_error("%s: bad dtype id %d", fn, dtype); return EXCENV_EINVAL; }
  if (B < 0 || B > ((int64_t)1 << 31) * BLOCK) { set_error("%s: bad batch size %lld", fn, (long long)B); return EXCENV_EINVAL; }
  return EXCENV_OK;
}

static int check_control(const char* fn, int env, const excenv_control_t*& c) {
  if (c && c->n_control == 0) c = nullptr;
  if (!c) return EXCENV_OK;
  if (c->n_control < 0 || c->n_control > EXCENV_MAX_CONTROL) { set_error("%s: bad n_control %d", fn, c->n_control); return EXCENV_EINVAL; }
  for (int j = 0; j < c->n_control; ++j) {
    if (c->control_idx[j] < 0 || c->control_idx[j] >= table(env)->S) { set_error("%s: control_idx[%d] out of range", fn, j); return EXCENV_EINVAL; }
    if (!c->reference[j]) { set_error("%s: reference[%d] is NULL", fn, j); return EXCENV_ENULL; }
  }
  return EXCENV_OK;
}

}  // namespace excenv

using namespace excenv;

extern "C" {

int excenv_abi_version(void) { return EXCENV_ABI_VERSION; }

const char* excenv_last_error(void) { return g_err; }

int excenv_env_dims(int env, int32_t* S, int32_t* A, int32_t* O, int32_t* P) {
  const EnvVTable* t = table(env);
  if (!t) { set_error("excenv_env_dims: bad env id %d", env); return EXCENV_EINVAL; }
  if (S) *S = t->S;
  if (A) *A = t->A;
  if (O) *O = t->O;
  if (P) *P = t->P;
  return EXCENV_OK;
}

int64_t excenv_step_bytes(int env, int dtype) {
  const EnvVTable* t = table(env);
  if (!t) return -1;
  const int64_t w = dtype == EXCENV_F64 ? 8 : 4;
  return w * (t->S + t->A + t->S + t->O);
}

int64_t excenv_sim_ahead_bytes(int env, int dtype, int with_state_traj) {
  const EnvVTable* t = table(env);
  if (!t) return -1;
  const int64_t w = dtype == EXCENV_F64 ? 8 : 4;
  return w * (t->A + t->O + (with_state_traj ? t->S : 0));
}

/* tuning knob (not part of the reference surface): key 0 = envs per lane for lane-major trajectories
 * (0 auto, 1/2/4 forced). Returns the previous value. */
int excenv_set_tuning(int key, int value) {
  if (key == 0) { int old = g_vec_pref; g_vec_pref = value; return old; }
  if (key == 1) { int old = g_lds_pad; g_lds_pad = value < 0 ? 0 : value; return old; }
  return EXCENV_EINVAL;
}

int excenv_step(int env, int solver, int dtype, int64_t B, const excenv_props_t* props,
                const excenv_control_t* control, double tau, const void* const* state_in, const void* action,
                void* const* state_out, void* obs, void* stream) {
  if (int rc = check_common("excenv_step", env, solver, dtype, B)) return rc;
  if (!props || !state_in || !action || !state_out || !obs) { set_error("excenv_step: NULL argument"); return EXCENV_ENULL; }
  if (int rc = check_control("excenv_step", env, control)) return rc;
  StepCall sc{solver, dtype, B, props, control, tau, state_in, action, state_out, obs, (hipStream_t)stream};
  return table(env)->step(sc);
}

int excenv_sim_ahead(int env, int solver, int dtype, int64_t B, int64_t K, int32_t substeps,
                     const excenv_props_t* props, const excenv_control_t* control, double obs_stepsize,
                     double env_tau, const void* const* state_in, const void* actions, int action_layout,
                     void* obs_traj, void* const* state_traj, int traj_layout, void* const* last_state,
                     int semantics, void* stream) {
  if (int rc = check_common("excenv_sim_ahead", env, solver, dtype, B)) return rc;
  if (K < 0 || substeps < 1) { set_error("excenv_sim_ahead: bad K=%lld or substeps=%d", (long long)K, substeps); return EXCENV_EINVAL; }
  if (semantics != EXCENV_SEM_STEP && semantics != EXCENV_SEM_AHEAD) { set_error("excenv_sim_ahead: bad semantics %d", semantics); return EXCENV_EINVAL; }
  if (action_layout < EXCENV_LAYOUT_ENV_MAJOR || action_layout > EXCENV_LAYOUT_TILED ||
      traj_layout < EXCENV_LAYOUT_ENV_MAJOR || traj_layout > EXCENV_LAYOUT_TILED) {
    set_error("excenv_sim_ahead: bad layout id");
    return EXCENV_EINVAL;
  }
  if (!props || !state_in || (!actions && K > 0) || !obs_traj || !last_state) { set_error("excenv_sim_ahead: NULL argument"); return EXCENV_ENULL; }
  if (int rc = check_control("excenv_sim_ahead", env, control)) return rc;
  SimCall sc{solver, dtype, B, K, substeps, props, control, obs_stepsize, env_tau, state_in, actions, action_layout,
             obs_traj, state_traj, traj_layout, last_state, semantics, g_vec_pref, g_lds_pad, (hipStream_t)stream};
  return table(env)->sim(sc);
}

int excenv_probe_math(int which, int dtype, int64_t n, const void* in, void* out, void* stream) {
  if (which < 0 || which > 2 || n < 0 || (dtype != EXCENV_F32 && dtype != EXCENV_F64)) { set_error("excenv_probe_math: bad argument"); return EXCENV_EINVAL; }
  if (!in || !out) { set_error("excenv_probe_math: NULL argument"); return EXCENV_ENULL; }
  if (n == 0) return EXCENV_OK;
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  if (dtype == EXCENV_F32)
    hipLaunchKernelGGL((probe_kernel<float>), grid, block, 0, (hipStream_t)stream, which, n, (const float*)in, (float*)out);
  else
    hipLaunchKernelGGL((probe_kernel<double>), grid, block, 0, (hipStream_t)stream, which, n, (const double*)in, (double*)out);
  return check_launch("excenv_probe_math");
}

}  // extern "C"
