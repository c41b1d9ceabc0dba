// extern "C" entry points of libexcenv_hip.so (declared in include/excenv.h): argument validation,
// per-thread error string, dispatch into the per-environment launch tables. No device allocation,
// no synchronisation — only kernel enqueues on the caller's stream.
#include <cstdarg>
#include <cstdio>
#include <dlfcn.h>
#include "launch.hpp"

static_assert(EXCENV_FAULT == 0, "EXCENV_FAULT builds exist for tools/isa_guards_selftest.sh only (single objects, never linked into the library)");

namespace excenv {

static thread_local char g_err[512] = "";
thread_local const char* g_last_launch = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int launch_transpose(int dtype, int64_t M, int64_t N, const void* in, void* out, hipStream_t stream);

EnvVTable vtable_pendulum();
EnvVTable vtable_msd();
EnvVTable vtable_cartpole();
EnvVTable vtable_acrobot();
EnvVTable vtable_tank();
EnvVTable vtable_pmsm();
EnvVTable vtable_pmsm_sat();

static const EnvVTable* table(int env) {
  static const EnvVTable T[EXCENV_NUM_ENVS + 1] = {vtable_pendulum(), vtable_msd(),  vtable_cartpole(), vtable_acrobot(),
                                                   vtable_tank(),     vtable_pmsm(), vtable_pmsm_sat()};
  if (env < 0 || env > EXCENV_NUM_ENVS) return nullptr;
  return &T[env];
}

static const EnvVTable* table_public(int env) { return (env >= 0 && env < EXCENV_NUM_ENVS) ? table(env) : nullptr; }

// Launch table for a call: PMSM with LUTs attached runs the saturated model's instantiations.
static const EnvVTable* table_for(int env, const excenv_props_t* props, int* rc) {
  *rc = EXCENV_OK;
  if (env < 0 || env >= EXCENV_NUM_ENVS) return nullptr;
  if (props && props->pmsm_lut) {
    const excenv_pmsm_lut_t* l = props->pmsm_lut;
    if (env != EXCENV_PMSM) { set_error("pmsm_lut is only valid for EXCENV_PMSM"); *rc = EXCENV_EINVAL; return nullptr; }
    if (l->n_d < 2 || l->n_q < 2 || !l->grid_d || !l->grid_q || !l->tables) {
      set_error("pmsm_lut: need n_d, n_q >= 2 and non-NULL grid / table pointers");
      *rc = EXCENV_EINVAL;
      return nullptr;
    }
    return table(EXCENV_NUM_ENVS);
  }
  return table(env);
}

static const excenv_launch_opts_t kDefaultOpts = {0, 0, 0, 0};  // envs_per_lane, env_major_mode, lds_pad_bytes, flags

static int check_opts(const char* fn, const excenv_launch_opts_t*& o) {
  if (!o) o = &kDefaultOpts;
  const int v = o->envs_per_lane;
  if (!(v == 0 || v == 1 || v == 2 || v == 4)) { set_error("%s: opts.envs_per_lane must be 0, 1, 2 or 4 (got %d)", fn, v); return EXCENV_EINVAL; }
  if (o->env_major_mode < 0 || o->env_major_mode > 3) { set_error("%s: opts.env_major_mode must be 0, 1, 2 or 3", fn); return EXCENV_EINVAL; }
  if (o->lds_pad_bytes < 0 || o->lds_pad_bytes > 150 * 1024) { set_error("%s: opts.lds_pad_bytes out of range", fn); return EXCENV_EINVAL; }
  if ((o->flags & ~EXCENV_OPT_NO_FUSED_ACTIONS) != 0) { set_error("%s: opts.flags has unknown bits set (0x%x)", fn, (unsigned)o->flags); return EXCENV_EINVAL; }
  return EXCENV_OK;
}

static int check_common(const char* fn, int env, int solver, int dtype, int64_t B) {
  if (env < 0 || env >= EXCENV_NUM_ENVS) { set_error("%s: bad env id %d", fn, env); return EXCENV_EINVAL; }
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
    if (c->control_idx[j] < 0 || c->control_idx[j] >= table_public(env)->S) { set_error("%s: control_idx[%d] out of range", fn, j); return EXCENV_EINVAL; }
    if (!c->reference[j]) { set_error("%s: reference[%d] is NULL", fn, j); return EXCENV_ENULL; }
  }
  return EXCENV_OK;
}

}  // namespace excenv

using namespace excenv;

extern "C" {

int excenv_abi_version(void) { return EXCENV_ABI_VERSION; }

const char* excenv_last_error(void) { return g_err; }

const char* excenv_last_launch(void) { return g_last_launch; }

int excenv_env_dims(int env, int32_t* S, int32_t* A, int32_t* O, int32_t* P) {
  const EnvVTable* t = table_public(env);
  if (!t) { set_error("excenv_env_dims: bad env id %d", env); return EXCENV_EINVAL; }
  if (S) *S = t->S;
  if (A) *A = t->A;
  if (O) *O = t->O;
  if (P) *P = t->P;
  return EXCENV_OK;
}

int64_t excenv_step_bytes(int env, int dtype) {
  const EnvVTable* t = table_public(env);
  if (!t) return -1;
  const int64_t w = dtype == EXCENV_F64 ? 8 : 4;
  return w * (t->S + t->A + t->S + t->O);
}

int64_t excenv_sim_ahead_bytes(int env, int dtype, int with_state_traj) {
  const EnvVTable* t = table_public(env);
  if (!t) return -1;
  const int64_t w = dtype == EXCENV_F64 ? 8 : 4;
  return w * (t->A + t->O + (with_state_traj ? t->S : 0));
}

int excenv_step(int env, int solver, int dtype, int64_t B, const excenv_props_t* props,
                const excenv_control_t* control, double tau, const void* const* state_in, const void* action,
                void* const* state_out, void* obs, const excenv_launch_opts_t* opts, void* stream) {
  if (int rc = check_common("excenv_step", env, solver, dtype, B)) return rc;
  if (!props || !state_in || !action || !state_out || !obs) { set_error("excenv_step: NULL argument"); return EXCENV_ENULL; }
  if (int rc = check_control("excenv_step", env, control)) return rc;
  if (int rc = check_opts("excenv_step", opts)) return rc;
  StepCall sc{opts->envs_per_lane, solver, dtype, B, props, control, tau, state_in, action, state_out, obs, nullptr, nullptr, nullptr,
              (hipStream_t)stream};
  int rc;
  const EnvVTable* vt = table_for(env, props, &rc);
  return vt ? vt->step(sc) : rc;
}

int32_t excenv_truncated_width(int env, int32_t n_control) {
  const EnvVTable* t = table_public(env);
  if (!t || n_control < 0) return -1;
  return (env == EXCENV_FLUID_TANK || env == EXCENV_PMSM) ? 1 : t->O + n_control;
}

int excenv_gym_step(int env, int solver, int dtype, int64_t B, const excenv_props_t* props,
                    const excenv_control_t* control, double tau, const void* const* state_in, const void* action,
                    void* const* state_out, void* obs, void* reward, uint8_t* terminated, uint8_t* truncated,
                    const excenv_launch_opts_t* opts, void* stream) {
  if (int rc = check_common("excenv_gym_step", env, solver, dtype, B)) return rc;
  if (!props || !state_in || !action || !state_out || !obs || !reward || !terminated || !truncated) {
    set_error("excenv_gym_step: NULL argument");
    return EXCENV_ENULL;
  }
  if (int rc = check_control("excenv_gym_step", env, control)) return rc;
  if (int rc = check_opts("excenv_gym_step", opts)) return rc;
  StepCall sc{1, solver, dtype, B, props, control, tau, state_in, action, state_out, obs, reward, terminated, truncated,
              (hipStream_t)stream};
  int rc;
  const EnvVTable* vt = table_for(env, props, &rc);
  return vt ? vt->step(sc) : rc;
}

static inline int64_t align_up(int64_t x) { return (x + 255) & ~(int64_t)255; }

int64_t excenv_sim_ahead_workspace_bytes(int env, int dtype, int64_t B, int64_t K, int32_t substeps, int32_t n_control,
                                         int action_layout, int traj_layout, int with_state_traj) {
  const EnvVTable* t = table_public(env);
  if (!t || B < 0 || K < 0 || substeps < 1 || n_control < 0) return -1;
  const int64_t w = dtype == EXCENV_F64 ? 8 : 4, N = K * substeps;
  int64_t bytes = 0;
  if (action_layout == EXCENV_LAYOUT_ENV_MAJOR) bytes += align_up(w * K * t->A * B);
  if (traj_layout == EXCENV_LAYOUT_ENV_MAJOR) {
    bytes += align_up(w * (N + 1) * (t->O + n_control) * B);
    if (with_state_traj) bytes += (int64_t)t->S * align_up(w * (N + 1) * B);
  }
  return bytes;
}

int excenv_sim_ahead_fuses_actions(int env, int solver, int dtype, int64_t B, int64_t K, const excenv_props_t* props,
                                   int32_t n_control, int with_gym, int action_layout, int traj_layout, const void* actions,
                                   const excenv_launch_opts_t* opts) {
  if (check_common("excenv_sim_ahead_fuses_actions", env, solver, dtype, B) || !props) return 0;
  if (check_opts("excenv_sim_ahead_fuses_actions", opts)) return 0;
  int trc;
  const EnvVTable* t = table_for(env, props, &trc);
  if (!t) return 0;
  (void)n_control;  // control columns are filled behind the lean kernel (control_fill_kernel): no reason not to fuse
  return aem_applies(env, props->pmsm_lut != nullptr, props_batched(props, t->P, t->S, t->A) || with_gym != 0, t->A,
                     dtype == EXCENV_F64 ? 8 : 4, B, K, solver, opts->envs_per_lane, action_layout, traj_layout, opts->flags, actions)
             ? 1 : 0;
}

int excenv_transpose(int dtype, int64_t M, int64_t N, const void* in, void* out, void* stream) {
  if ((dtype != EXCENV_F32 && dtype != EXCENV_F64) || M < 0 || N < 0) { set_error("excenv_transpose: bad argument"); return EXCENV_EINVAL; }
  if ((!in || !out) && M * N > 0) { set_error("excenv_transpose: NULL argument"); return EXCENV_ENULL; }
  int rc = launch_transpose(dtype, M, N, in, out, (hipStream_t)stream);
  if (rc) set_error("excenv_transpose: launch failed");
  return rc;
}

int excenv_sim_ahead_ws(int env, int solver, int dtype, int64_t B, int64_t K, int32_t substeps,
                        const excenv_props_t* props, const excenv_control_t* control, double obs_stepsize,
                        double env_tau, const void* const* state_in, const void* actions, int action_layout,
                        void* obs_traj, void* const* state_traj, int traj_layout, void* const* last_state,
                        int semantics, const excenv_traj_gym_t* gym, void* workspace, int64_t workspace_bytes,
                        const excenv_launch_opts_t* opts, void* stream) {
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
  if (int rc = check_opts("excenv_sim_ahead", opts)) return rc;
  int trc;
  const EnvVTable* t = table_for(env, props, &trc);
  if (!t) return trc;
  const int nc = control ? control->n_control : 0;
  const int64_t wbytes = dtype == EXCENV_F64 ? 8 : 4;
  // one decision for the whole call: fused env-major kernel / workspace + transposes / generic strides
  // (the fused kernel reads the action array in whole 16-byte pieces: it must start on one and consist of whole ones —
  // otherwise the generic-stride / workspace paths take the call)
  const bool fused_em = B > 0 && K > 0 && aligned16(obs_traj) && aligned16(actions) && (B * K * (int64_t)t->A * wbytes) % 16 == 0 &&
                        em_fused_eligible(opts->env_major_mode, action_layout, traj_layout, substeps, gym != nullptr, t->A, t->S,
                                          t->O, (size_t)wbytes);
  const int64_t need = excenv_sim_ahead_workspace_bytes(env, dtype, B, K, substeps, nc, action_layout, traj_layout,
                                                        state_traj != nullptr);
  // row-major actions + lane-major trajectories: the trajectory kernel reads the actions itself (no workspace, no extra pass)
  bool refs_ok = true;  // control columns next to fused actions need every reference array (control_fill_kernel reads them)
  for (int j = 0; j < nc; ++j) refs_ok &= control->reference[j] != nullptr;
  const bool fused_actions =
      aem_applies(env, props->pmsm_lut != nullptr, props_batched(props, t->P, t->S, t->A) || !refs_ok || gym != nullptr, t->A,
                  (size_t)wbytes, B, K, solver, opts->envs_per_lane, action_layout, traj_layout, opts->flags, actions) &&
      aligned16(obs_traj);
  const bool via_ws = !fused_em && !fused_actions && !gym && workspace && need > 0 && workspace_bytes >= need && B > 0 &&
                      (action_layout == EXCENV_LAYOUT_ENV_MAJOR || traj_layout == EXCENV_LAYOUT_ENV_MAJOR);
  const int em = !fused_em ? 1 : (opts->env_major_mode == 2 ? 3 : (opts->env_major_mode == 3 ? 4 : 2));
  if (!via_ws) {
    SimCall sc{solver, dtype, B, K, substeps, props, control, obs_stepsize, env_tau, state_in, actions, action_layout,
               obs_traj, state_traj, traj_layout, last_state, semantics, opts->envs_per_lane, opts->lds_pad_bytes, em, gym,
               (hipStream_t)stream};
    sc.flags = opts->flags;
    return t->sim(sc);
  }
  // env-major buffers + workspace: transpose in, run the coalesced lane-major kernel, transpose out
  const int64_t w = wbytes, N = K * substeps, OW = t->O + nc;
  char* ws = (char*)workspace;
  const void* k_actions = actions;
  int k_alayout = action_layout, k_tlayout = traj_layout;
  void* k_obs = obs_traj;
  void* k_straj[EXCENV_MAX_STATE] = {nullptr};
  void* const* k_straj_p = state_traj;
  hipStream_t st = (hipStream_t)stream;
  if (action_layout == EXCENV_LAYOUT_ENV_MAJOR) {
    if (int rc = launch_transpose(dtype, B, K * t->A, actions, ws, st)) { set_error("excenv_sim_ahead: action transpose failed"); return rc; }
    k_actions = ws;
    k_alayout = EXCENV_LAYOUT_LANE_MAJOR;
    ws += align_up(w * K * t->A * B);
  }
  if (traj_layout == EXCENV_LAYOUT_ENV_MAJOR) {
    k_obs = ws;
    ws += align_up(w * (N + 1) * OW * B);
    if (state_traj) {
      for (int j = 0; j < t->S; ++j) { k_straj[j] = ws; ws += align_up(w * (N + 1) * B); }
      k_straj_p = k_straj;
    }
    k_tlayout = EXCENV_LAYOUT_LANE_MAJOR;
  }
  SimCall sc{solver, dtype, B, K, substeps, props, control, obs_stepsize, env_tau, state_in, k_actions, k_alayout,
             k_obs, k_straj_p, k_tlayout, last_state, semantics, opts->envs_per_lane, opts->lds_pad_bytes, 1, nullptr, st};
  if (int rc = t->sim(sc)) return rc;
  g_last_launch = "transposition workspace + sim_ahead_kernel";
  if (traj_layout == EXCENV_LAYOUT_ENV_MAJOR) {
    if (int rc = launch_transpose(dtype, (N + 1) * OW, B, k_obs, obs_traj, st)) { set_error("excenv_sim_ahead: obs transpose failed"); return rc; }
    if (state_traj)
      for (int j = 0; j < t->S; ++j)
        if (int rc = launch_transpose(dtype, N + 1, B, k_straj[j], state_traj[j], st)) { set_error("excenv_sim_ahead: state transpose failed"); return rc; }
  }
  return EXCENV_OK;
}

int excenv_sim_ahead(int env, int solver, int dtype, int64_t B, int64_t K, int32_t substeps,
                     const excenv_props_t* props, const excenv_control_t* control, double obs_stepsize,
                     double env_tau, const void* const* state_in, const void* actions, int action_layout,
                     void* obs_traj, void* const* state_traj, int traj_layout, void* const* last_state,
                     int semantics, const excenv_traj_gym_t* gym, const excenv_launch_opts_t* opts, void* stream) {
  return excenv_sim_ahead_ws(env, solver, dtype, B, K, substeps, props, control, obs_stepsize, env_tau, state_in, actions,
                             action_layout, obs_traj, state_traj, traj_layout, last_state, semantics, gym, nullptr, 0, opts,
                             stream);
}

int excenv_rew_trunc_term(int env, int dtype, int64_t B, int64_t rows, const excenv_props_t* props,
                          const excenv_control_t* control, const int64_t* ref_strides, const void* const* state_traj,
                          int64_t state_env_stride, int64_t state_row_stride, void* reward, uint8_t* terminated,
                          uint8_t* truncated, int out_layout, void* stream) {
  if (int rc = check_common("excenv_rew_trunc_term", env, 0, dtype, B)) return rc;
  if (rows < 1) { set_error("excenv_rew_trunc_term: rows must be >= 1 (row 0 is the initial state)"); return EXCENV_EINVAL; }
  if (out_layout != EXCENV_LAYOUT_ENV_MAJOR && out_layout != EXCENV_LAYOUT_LANE_MAJOR) { set_error("excenv_rew_trunc_term: bad out_layout"); return EXCENV_EINVAL; }
  if (!props || !state_traj || !truncated || (rows > 1 && (!reward || !terminated))) { set_error("excenv_rew_trunc_term: NULL argument"); return EXCENV_ENULL; }
  if (int rc = check_control("excenv_rew_trunc_term", env, control)) return rc;
  int trc;
  const EnvVTable* t = table_for(env, props, &trc);
  if (!t) return trc;
  TrajGymCall gc{dtype, B, rows, props, control, ref_strides, state_traj, state_env_stride, state_row_stride, reward,
                 terminated, truncated, out_layout, (hipStream_t)stream};
  return t->traj_gym(gc);
}

int excenv_state_from_observation(int env, int dtype, int64_t B, const excenv_props_t* props, int32_t n_control,
                                  const int32_t* control_idx, const void* obs, void* const* state_out,
                                  void* const* reference_out, void* stream) {
  if (int rc = check_common("excenv_state_from_observation", env, 0, dtype, B)) return rc;
  if (n_control < 0 || n_control > EXCENV_MAX_CONTROL) { set_error("excenv_state_from_observation: bad n_control %d", n_control); return EXCENV_EINVAL; }
  if (!props || !obs || !state_out || (n_control > 0 && (!control_idx || !reference_out))) { set_error("excenv_state_from_observation: NULL argument"); return EXCENV_ENULL; }
  for (int j = 0; j < n_control; ++j)
    if (control_idx[j] < 0 || control_idx[j] >= table_public(env)->S) { set_error("excenv_state_from_observation: control_idx[%d] out of range", j); return EXCENV_EINVAL; }
  int trc;
  const EnvVTable* t = table_for(env, props, &trc);
  if (!t) return trc;
  FromObsCall fc{dtype, B, props, n_control, control_idx, obs, state_out, reference_out, (hipStream_t)stream};
  return t->from_obs(fc);
}

int excenv_observe(int env, int dtype, int64_t B, const excenv_props_t* props, const excenv_control_t* control,
                   const void* const* state, void* obs, void* stream) {
  if (int rc = check_common("excenv_observe", env, 0, dtype, B)) return rc;
  if (!props || !state || !obs) { set_error("excenv_observe: NULL argument"); return EXCENV_ENULL; }
  if (control) {
    if (control->n_control < 0 || control->n_control > EXCENV_MAX_CONTROL) { set_error("excenv_observe: bad n_control %d", control->n_control); return EXCENV_EINVAL; }
    for (int j = 0; j < control->n_control; ++j)
      if (control->control_idx[j] < 0 || control->control_idx[j] >= table_public(env)->S) { set_error("excenv_observe: control_idx[%d] out of range", j); return EXCENV_EINVAL; }
  }
  int trc;
  const EnvVTable* t = table_for(env, props, &trc);
  if (!t) return trc;
  ObserveCall oc{dtype, B, props, control, state, obs, (hipStream_t)stream};
  return t->observe(oc);
}

int excenv_update_ref(int env, int dtype, int64_t B, const excenv_props_t* props, int32_t n_control,
                      const int32_t* control_idx, void* const* reference, int64_t* keys, int64_t* hold,
                      int32_t hold_steps_min, int32_t hold_steps_max, void* stream) {
  if (int rc = check_common("excenv_update_ref", env, 0, dtype, B)) return rc;
  if (n_control < 0 || n_control > EXCENV_MAX_CONTROL) { set_error("excenv_update_ref: bad n_control %d", n_control); return EXCENV_EINVAL; }
  if (!props || !keys || !hold || (n_control > 0 && (!control_idx || !reference))) { set_error("excenv_update_ref: NULL argument"); return EXCENV_ENULL; }
  for (int j = 0; j < n_control; ++j)
    if (control_idx[j] < 0 || control_idx[j] >= table_public(env)->S) { set_error("excenv_update_ref: control_idx[%d] out of range", j); return EXCENV_EINVAL; }
  int trc;
  const EnvVTable* t = table_for(env, props, &trc);
  if (!t) return trc;
  RefGenCall rc{dtype, B, props, n_control, control_idx, reference, keys, hold, hold_steps_min, hold_steps_max, (hipStream_t)stream};
  return t->update_ref(rc);
}

int excenv_update_ref_to(int env, int dtype, int64_t B, const excenv_props_t* props, int32_t n_control,
                         const int32_t* control_idx, const void* const* reference_in, const int64_t* keys_in,
                         const int64_t* hold_in, void* const* reference_out, int64_t* keys_out, int64_t* hold_out,
                         int32_t hold_steps_min, int32_t hold_steps_max, void* stream) {
  if (int rc = check_common("excenv_update_ref_to", env, 0, dtype, B)) return rc;
  if (n_control < 0 || n_control > EXCENV_MAX_CONTROL) { set_error("excenv_update_ref_to: bad n_control %d", n_control); return EXCENV_EINVAL; }
  if (!props || !keys_in || !hold_in || !keys_out || !hold_out || (n_control > 0 && (!control_idx || !reference_in || !reference_out))) {
    set_error("excenv_update_ref_to: NULL argument");
    return EXCENV_ENULL;
  }
  if (keys_in == keys_out || hold_in == hold_out) { set_error("excenv_update_ref_to: outputs must not alias the inputs (use excenv_update_ref)"); return EXCENV_EINVAL; }
  for (int j = 0; j < n_control; ++j) {
    if (control_idx[j] < 0 || control_idx[j] >= table_public(env)->S) { set_error("excenv_update_ref_to: control_idx[%d] out of range", j); return EXCENV_EINVAL; }
    if (reference_in[j] == reference_out[j]) { set_error("excenv_update_ref_to: outputs must not alias the inputs (use excenv_update_ref)"); return EXCENV_EINVAL; }
  }
  int trc;
  const EnvVTable* t = table_for(env, props, &trc);
  if (!t) return trc;
  RefGenCall rc{dtype, B, props, n_control, control_idx, reference_out, keys_out, hold_out, hold_steps_min, hold_steps_max,
                (hipStream_t)stream, reference_in, keys_in, hold_in};
  return t->update_ref(rc);
}

int excenv_random_state(int env, int dtype, int64_t B, const excenv_props_t* props, const int64_t* keys,
                        void* const* state_out, int64_t* key_leaf, void* stream) {
  if (int rc = check_common("excenv_random_state", env, 0, dtype, B)) return rc;
  if (!props || !keys || !state_out || !key_leaf) { set_error("excenv_random_state: NULL argument"); return EXCENV_ENULL; }
  int trc;
  const EnvVTable* t = table_for(env, props, &trc);
  if (!t) return trc;
  RandomStateCall rc{dtype, B, props, keys, state_out, key_leaf, (hipStream_t)stream};
  return t->random_state(rc);
}

// ---- the one collective of the path: reassemble observations on every rank -------------------------------
// RCCL is resolved at run time (dlopen: librccl.so, the library torch.distributed's "nccl" backend uses on ROCm) so that
// single-GPU users of libexcenv_hip.so do not need it; the communicator is the caller's.
namespace {
typedef int (*nccl_allgather_fn)(const void*, void*, size_t, int, void*, hipStream_t);
struct RcclEntry {
  nccl_allgather_fn fn = nullptr;
  const char* err = nullptr;
  RcclEntry() {
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void* h = nullptr;
    for (const char* n : names) {
      h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (h) break;
    }
    if (!h) err = "librccl.so could not be loaded";
    else {
      fn = (nccl_allgather_fn)dlsym(h, "ncclAllGather");
      if (!fn) err = "librccl.so has no ncclAllGather";
    }
  }
};
nccl_allgather_fn resolve_allgather(const char** why) {
  static const RcclEntry entry;  // initialised once, thread-safely (C++11 function-local static); read-only afterwards
  if (why) *why = entry.err;
  return entry.fn;
}
}  // namespace

int excenv_allgather(void* nccl_comm, int dtype, const void* send, void* recv, int64_t count_per_rank, void* stream) {
  if (dtype != EXCENV_F32 && dtype != EXCENV_F64) { set_error("excenv_allgather: bad dtype id %d", dtype); return EXCENV_EINVAL; }
  if (count_per_rank < 0) { set_error("excenv_allgather: bad count %lld", (long long)count_per_rank); return EXCENV_EINVAL; }
  if (!nccl_comm || ((!send || !recv) && count_per_rank > 0)) { set_error("excenv_allgather: NULL argument"); return EXCENV_ENULL; }
  if (count_per_rank == 0) return EXCENV_OK;
  const char* why = nullptr;
  nccl_allgather_fn fn = resolve_allgather(&why);
  if (!fn) { set_error("excenv_allgather: %s", why ? why : "RCCL unavailable"); return EXCENV_EUNSUPPORTED; }
  const int nccl_type = dtype == EXCENV_F32 ? 7 : 8;  // ncclFloat32 / ncclFloat64 (rccl.h)
  const int rc = fn(send, recv, (size_t)count_per_rank, nccl_type, nccl_comm, (hipStream_t)stream);
  if (rc != 0) { set_error("excenv_allgather: ncclAllGather failed with ncclResult_t %d", rc); return EXCENV_EHIP; }
  return EXCENV_OK;
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

int excenv_probe_div(int dtype, int64_t n, const void* num, const void* den, void* out_fast, void* out_ref, void* stream) {
  if (n < 0 || (dtype != EXCENV_F32 && dtype != EXCENV_F64)) { set_error("excenv_probe_div: bad argument"); return EXCENV_EINVAL; }
  if (!num || !den || !out_fast || !out_ref) { set_error("excenv_probe_div: NULL argument"); return EXCENV_ENULL; }
  if (n == 0) return EXCENV_OK;
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  if (dtype == EXCENV_F32)
    hipLaunchKernelGGL((probe_div_kernel<float>), grid, block, 0, (hipStream_t)stream, n, (const float*)num, (const float*)den,
                       (float*)out_fast, (float*)out_ref);
  else
    hipLaunchKernelGGL((probe_div_kernel<double>), grid, block, 0, (hipStream_t)stream, n, (const double*)num,
                       (const double*)den, (double*)out_fast, (double*)out_ref);
  return check_launch("excenv_probe_div");
}

}  // extern "C"
