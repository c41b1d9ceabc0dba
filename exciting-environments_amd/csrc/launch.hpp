// Host-side glue between the C ABI (include/excenv.h) and the templated kernels: converts the untyped
// call into typed kernel arguments, picks the instantiation, enqueues it on the caller's stream.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include "kernels_emr.hpp"
#include "refgen.hpp"

namespace excenv {

void set_error(const char* fmt, ...);
// Set by EXCENV_LAUNCH_DYN when raising a kernel's dynamic-LDS limit failed (the launch is then skipped and
// check_launch reports the stored message instead of a generic launch error). Per thread, like the error string.
static thread_local bool g_attr_failed = false;
// Which form of the trajectory kernel the last excenv_sim_ahead[_ws] call of this thread enqueued (excenv_last_launch(): tests
// assert that the path they mean to check is the one that ran; like the error string it is per thread and purely informational).
extern thread_local const char* g_last_launch;  // defined in excenv_api.hip

struct StepCall {
  int vec_pref;  // 0 auto, else forced envs per lane
  int solver, dtype;
  int64_t B;
  const excenv_props_t* props;
  const excenv_control_t* control;  // nullptr when n_control == 0
  double tau;
  const void* const* state_in;
  const void* action;
  void* const* state_out;
  void* obs;
  void* reward;       // optional gym outputs
  void* terminated;
  void* truncated;
  hipStream_t stream;
};

struct SimCall {
  int solver, dtype;
  int64_t B, K;
  int32_t substeps;
  const excenv_props_t* props;
  const excenv_control_t* control;
  double obs_stepsize, env_tau;
  const void* const* state_in;
  const void* actions;
  int action_layout;
  void* obs_traj;
  void* const* state_traj;  // may be nullptr
  int traj_layout;
  void* const* last_state;
  int semantics;
  int vec_pref;  // 0 auto, else forced envs-per-lane (1, 2, 4)
  int lds_pad;   // dynamic LDS bytes per workgroup (occupancy shaping experiments; 0 = none)
  int em_mode;   // env-major buffers: 1 = not fused (generic strides / workspace); fused: 2 = pick the kernel, 3 = LDS-ring kernel,
                 // 4 = register-ring kernel whenever its preconditions hold (no batch-size heuristic)
  const excenv_traj_gym_t* gym;  // optional reward / terminated / truncated trajectories
  hipStream_t stream;
  int flags = 0;  // excenv_launch_opts_t.flags
};

// Row-major actions [B][K][A] with lane-major trajectories — what a reference-shaped vmap_sim_ahead call with the library's
// default outputs is: the widest lean instantiation reads them itself through a per-wave LDS piece ring (kernels.hpp, AEM)
// instead of a transposition pass in front of the launch. Decided once per call (excenv_api.hip decides with the same function
// whether a workspace transposition is needed at all). EXCENV_AEM=0 in the environment switches it off (A/B measurements).
static inline bool aem_enabled() {
  static const int on = [] { const char* e = std::getenv("EXCENV_AEM"); return (e && e[0] == '0') ? 0 : 1; }();
  return on != 0;
}
static inline int auto_envs_per_lane(int64_t B, int vmax);
static inline bool widest_form_pays(int64_t B, int vmax);
static inline bool aem_applies(int env_id, bool has_lut, bool general, int A, size_t elem, int64_t B, int64_t K, int solver,
                               int vec_pref, int action_layout, int traj_layout, int flags, const void* actions) {
  const int vmax = 16 / (int)elem;
  if (!aem_enabled() || (flags & EXCENV_OPT_NO_FUSED_ACTIONS) || has_lut || general) return false;
  if (action_layout != EXCENV_LAYOUT_ENV_MAJOR || traj_layout != EXCENV_LAYOUT_LANE_MAJOR) return false;
  if (K < 1 || (vmax % A) != 0 || (K * A) % vmax != 0) return false;  // whole 16-byte pieces per row
  if ((B % (64 * vmax)) != 0) return false;  // whole waves: the lanes of a wave fetch action windows for each other
  if ((reinterpret_cast<uintptr_t>(actions) & 15u) != 0) return false;
  if ((int64_t)EXCENV_BLOCK * vmax * K * A >= ((int64_t)1 << 32)) return false;             // 32-bit element offsets inside a workgroup
  // only where the batch takes the widest form anyway (launch_sim picks the same way)
  int want = vec_pref > 0 ? vec_pref : (widest_form_pays(B, vmax) ? vmax : 1);
  if (vec_pref == 0 && env_id == EXCENV_ACROBOT && solver != EXCENV_EULER && want > 2) want = 2;
  return want == vmax;
}
static inline bool props_batched(const excenv_props_t* p, int P, int S, int A) {
  bool b = false;
  for (int j = 0; j < P; ++j) b |= p->static_params[j].per_env != nullptr;
  for (int j = 0; j < S; ++j) b |= p->state_min[j].per_env != nullptr || p->state_max[j].per_env != nullptr;
  for (int j = 0; j < A; ++j) b |= p->action_min[j].per_env != nullptr || p->action_max[j].per_env != nullptr;
  return b;
}

// The fused env-major kernel applies when both layouts are env-major, substeps == 1, the caller did not opt out, no gym
// trajectories are requested and the time tile fits LDS. Decided once per call (excenv_api.hip) and handed to launch_sim.
// EXCENV_EM_RING=0 in the environment keeps the LDS-ring kernel (A/B measurements)
static inline bool emr_enabled() {
  static const int on = [] { const char* e = std::getenv("EXCENV_EM_RING"); return (e && e[0] == '0') ? 0 : 1; }();
  return on != 0;
}

static inline bool em_fused_eligible(int em_mode, int action_layout, int traj_layout, int32_t substeps, bool with_gym,
                                     int A, int S, int O, size_t elem) {
  return em_mode != 1 && action_layout == EXCENV_LAYOUT_ENV_MAJOR && traj_layout == EXCENV_LAYOUT_ENV_MAJOR &&
         substeps == 1 && !with_gym && (elem == 8 ? em_lds_elems<double>(A, S, O) : em_lds_elems<float>(A, S, O)) * elem <= 150 * 1024;
}

struct TrajGymCall {
  int dtype;
  int64_t B, rows;
  const excenv_props_t* props;
  const excenv_control_t* control;
  const int64_t* ref_strides;  // [n_control][2] element strides (env, row) of each reference array
  const void* const* state_traj;
  int64_t s_sb, s_sk;  // element strides (env, row) of every state leaf
  void* reward;
  uint8_t* terminated;
  uint8_t* truncated;
  int out_layout;
  hipStream_t stream;
};

struct FromObsCall {
  int dtype;
  int64_t B;
  const excenv_props_t* props;
  int32_t n_control;
  const int32_t* control_idx;
  const void* obs;
  void* const* state_out;
  void* const* reference_out;
  hipStream_t stream;
};

struct ObserveCall {
  int dtype;
  int64_t B;
  const excenv_props_t* props;
  const excenv_control_t* control;
  const void* const* state;
  void* obs;
  hipStream_t stream;
};

struct RefGenCall {
  int dtype;
  int64_t B;
  const excenv_props_t* props;
  int32_t n_control;
  const int32_t* control_idx;
  void* const* reference;
  int64_t* keys;
  int64_t* hold;
  int32_t hold_min, hold_max;
  hipStream_t stream;
  // inputs of the out-of-place form (nullptr: in place)
  const void* const* reference_in = nullptr;
  const int64_t* keys_in = nullptr;
  const int64_t* hold_in = nullptr;
};

struct RandomStateCall {
  int dtype;
  int64_t B;
  const excenv_props_t* props;
  const int64_t* keys;
  void* const* state_out;
  int64_t* key_leaf;
  hipStream_t stream;
};

struct EnvVTable {
  int S, A, O, P;
  int (*step)(const StepCall&);
  int (*sim)(const SimCall&);
  int (*traj_gym)(const TrajGymCall&);
  int (*from_obs)(const FromObsCall&);
  int (*update_ref)(const RefGenCall&);
  int (*random_state)(const RandomStateCall&);
  int (*observe)(const ObserveCall&);
};

template <typename T, class M>
static bool fill_props(KProps<T, M>& kp, const excenv_props_t* p) {
  bool batched = false;
  int n = 0;
  auto put = [&](const excenv_param_t& q) {
    kp.scalar[n] = (T)q.value;
    kp.ptr[n] = (const T*)q.per_env;
    batched |= (q.per_env != nullptr);
    ++n;
  };
  for (int j = 0; j < M::P; ++j) put(p->static_params[j]);
  for (int j = 0; j < M::S; ++j) put(p->state_min[j]);
  for (int j = 0; j < M::S; ++j) put(p->state_max[j]);
  for (int j = 0; j < M::A; ++j) put(p->action_min[j]);
  for (int j = 0; j < M::A; ++j) put(p->action_max[j]);
  kp.lut_gd = kp.lut_gq = kp.lut_tab = nullptr;
  kp.lut_nd = kp.lut_nq = 0;
  kp.lut_lds = 0;
  if (p->pmsm_lut) {
    kp.lut_gd = (const T*)p->pmsm_lut->grid_d;
    kp.lut_gq = (const T*)p->pmsm_lut->grid_q;
    kp.lut_tab = (const T*)p->pmsm_lut->tables;
    kp.lut_nd = p->pmsm_lut->n_d;
    kp.lut_nq = p->pmsm_lut->n_q;
  }
  return batched;
}

// Launch with dynamic LDS; above the default 64 KiB limit the kernel's attribute is raised first (gfx950: 160 KiB per CU).
#define EXCENV_LAUNCH_DYN(KERNEL, GRID, BLOCK, LDS, STREAM, ARGS)                                                         \
  do {                                                                                                                   \
    bool excenv_attr_ok = true;                                                                                          \
    if ((LDS) > 64 * 1024) {                                                                                             \
      const hipError_t excenv_e = hipFuncSetAttribute(reinterpret_cast<const void*>(&KERNEL),                            \
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDS));            \
      if (excenv_e != hipSuccess) {                                                                                      \
        set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize = %d) failed: %s", (int)(LDS),                         \
                  hipGetErrorString(excenv_e));                                                                          \
        g_attr_failed = true;                                                                                            \
        excenv_attr_ok = false;                                                                                          \
      }                                                                                                                  \
    }                                                                                                                    \
    if (excenv_attr_ok) hipLaunchKernelGGL(KERNEL, GRID, BLOCK, LDS, STREAM, ARGS);                                      \
  } while (0)

// Dynamic LDS for the saturated model's tables: staged when they fit LDS (<= 150 KiB, leaving room for one workgroup).
// `other`: further dynamic LDS of the launch (returned with the tables' share); `static_bytes`: static LDS of the kernel itself
// (step_kernel's dense observation staging) — it counts against the workgroup's limit but is not part of the dynamic size.
template <typename T, class M> static size_t lut_lds_bytes(KProps<T, M>& kp, size_t other, size_t static_bytes = 0) {
  if constexpr (!M::HAS_LUT) return other;
  const size_t need = ((size_t)kp.lut_nd * kp.lut_nq * 8 + 2 * (size_t)(kp.lut_nd + kp.lut_nq)) * sizeof(T);  // tables, grids, cell-width reciprocals
  if (kp.lut_tab && need + other + static_bytes <= 150 * 1024) {
    kp.lut_lds = 1;
    return need + other;
  }
  kp.lut_lds = 0;
  return other;
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

static inline int check_launch(const char* what) {
  if (g_attr_failed) {  // message already set by EXCENV_LAUNCH_DYN
    g_attr_failed = false;
    (void)hipGetLastError();
    return EXCENV_EHIP;
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: HIP launch failed: %s", what, hipGetErrorString(e));
    return EXCENV_EHIP;
  }
  return EXCENV_OK;
}

// PMSM dead time: a broadcast non-negative integer. The reference's state holds ONE buffered action, so any deadtime > 0 is a
// one-step delay (PMSM.step, pmsm_env.py:866-875) while the clip's angle advance uses the full (deadtime + 0.5) * tau
// (pmsm_env.py:599-604) — returned here, folded in double like the Python expression. `ahead`: the reference's sim_ahead
// override assembles deadtime + K buffer rows for K + 1 saved rows (pmsm_env.py:765-791), which only works for deadtime 0 / 1
// (its own vmap rejects the shapes otherwise), so EXCENV_SEM_AHEAD takes 0 / 1 only; EXCENV_SEM_STEP (K exact steps) takes any.
template <class M> static int pmsm_coef(const excenv_props_t* p, double env_tau, double* coef, bool ahead = false) {
  *coef = 0.0;
  if constexpr (M::IS_PMSM) {
    const excenv_param_t& d = p->static_params[6];
    if (d.per_env) {
      set_error("PMSM: static_params.deadtime must be a scalar, not a per-env array");
      return EXCENV_EUNSUPPORTED;
    }
    if (!(d.value >= 0.0) || d.value != (double)(int64_t)d.value || d.value > 1e6) {
      set_error("PMSM: deadtime must be a non-negative integer (got %g)", d.value);
      return EXCENV_EUNSUPPORTED;
    }
    if (ahead && d.value > 1.0) {
      set_error("PMSM: EXCENV_SEM_AHEAD supports deadtime 0 or 1 (got %g): the reference's sim_ahead builds K + deadtime buffer "
                "rows for K + 1 saved rows (pmsm_env.py:765-791) and fails for more; use EXCENV_SEM_STEP", d.value);
      return EXCENV_EUNSUPPORTED;
    }
    *coef = (d.value + 0.5) * env_tau;
  }
  return EXCENV_OK;
}

// Envs per lane for a batch: the widest 16-byte form that still leaves at least one wave per SIMD on the chip
// (256 CUs x 4 SIMDs = 1024 waves of 64 lanes); small batches run one env per lane so that the per-step dependent chain
// of a wave is as short as possible (batch sweep).
// Environments per lane by batch size. Two per lane from one wave per SIMD on the chip (1024 SIMDs x 64 lanes), FOUR only from two
// waves per SIMD: at B = 2^18 four per lane leave one 256-thread workgroup per CU — round 4, same-buffers A/B and fresh processes:
// PMSM Euler 0.411 -> 0.325 ms with two per lane, pendulum 1.259 -> 0.963, cart-pole 0.228 -> 0.153; from 2^19 on four win.
static inline int auto_envs_per_lane(int64_t B, int vmax) {
  int v = vmax;
  while (v > 1 && (B / v) < (int64_t)1024 * 64 * (v >= 4 ? 2 : 1)) v >>= 1;
  return v;
}
// The forms that exist only at the widest lane width (row-major actions read by the kernel, lean gym outputs) keep the earlier bound:
// they beat what the call would fall back to (a transposition pass, the one-environment general kernel) from one wave per SIMD on.
static inline bool widest_form_pays(int64_t B, int vmax) { return (B / vmax) >= (int64_t)1024 * 64; }

template <class M, typename T> static int launch_step(const StepCall& sc) {
  StepArgs<T, M> ka;
  std::memset(&ka, 0, sizeof(ka));
  const bool batched = fill_props<T, M>(ka.kp, sc.props);
  double coef;
  if (int rc = pmsm_coef<M>(sc.props, sc.tau, &coef)) return rc;
  ka.B = sc.B;
  for (int j = 0; j < M::S; ++j) {
    if (!sc.state_in[j] || !sc.state_out[j]) { set_error("excenv_step: state pointer %d is NULL", j); return EXCENV_ENULL; }
    ka.state_in[j] = (const T*)sc.state_in[j];
    ka.state_out[j] = (T*)sc.state_out[j];
  }
  ka.action = (const T*)sc.action;
  ka.obs = (T*)sc.obs;
  ka.n_control = sc.control ? sc.control->n_control : 0;
  for (int j = 0; j < ka.n_control; ++j) {
    ka.control_idx[j] = sc.control->control_idx[j];
    ka.reference[j] = (const T*)sc.control->reference[j];
    ka.obs_reference[j] = sc.control->obs_reference[j] ? (const T*)sc.control->obs_reference[j] : ka.reference[j];
  }
  ka.dt = (T)sc.tau;
  ka.env_tau = (T)sc.tau;
  ka.adv_coef = (T)coef;
  ka.reward = (T*)sc.reward;
  ka.terminated = (uint8_t*)sc.terminated;
  ka.truncated = (uint8_t*)sc.truncated;
  if (!aligned16(ka.action) || !aligned16(ka.obs)) {
    set_error("excenv_step: action and obs must be 16-byte aligned");
    return EXCENV_EINVAL;
  }
  if (sc.B == 0) return EXCENV_OK;
  constexpr int VMAX = 16 / (int)sizeof(T);
  const bool general = batched || ka.n_control > 0 || ka.reward != nullptr;
  int V = 1;
  if (!general) {
    bool ok = true;
    for (int j = 0; j < M::S; ++j) ok &= aligned16(ka.state_in[j]) && aligned16(ka.state_out[j]);
    int want = sc.vec_pref > 0 ? sc.vec_pref : 1;  // measured: one env per lane is fastest on this path
    if (want > VMAX) want = VMAX;
    while (want > 1 && (sc.B % want) != 0) want >>= 1;
    if (ok) V = want;
  }
  const int64_t lanes = sc.B / V;
  const dim3 grid((unsigned)((lanes + BLOCK - 1) / BLOCK)), block(BLOCK);
  // the DENSE observation path of step_kernel (kernels.hpp) stages BLOCK rows in static LDS: tables that would not fit next to
  // it stay in global memory (L2) instead of failing the launch
  constexpr int VWs = 16 / (int)sizeof(T);
  const bool dense = !general && V == 1 && (M::O % VWs) == 0 && (M::O / VWs) > 1;
  const size_t step_lds = lut_lds_bytes<T, M>(ka.kp, 0, dense ? sizeof(T) * BLOCK * M::O : 0);
#define EXCENV_STEP_LAUNCH(SOLV, GEN, VV) EXCENV_LAUNCH_DYN((step_kernel<M, T, SOLV, GEN, VV>), grid, block, step_lds, sc.stream, ka)
#define EXCENV_STEP_CASE(SOLV)                                                 \
  case SOLV:                                                                   \
    if (general) EXCENV_STEP_LAUNCH(SOLV, true, 1);                            \
    else if (V == 2) EXCENV_STEP_LAUNCH(SOLV, false, 2);                       \
    else if (V == 1) EXCENV_STEP_LAUNCH(SOLV, false, 1);                       \
    else { if constexpr (sizeof(T) == 4) EXCENV_STEP_LAUNCH(SOLV, false, 4); } \
    break;
  switch (sc.solver) {
    EXCENV_STEP_CASE(EXCENV_EULER)
    EXCENV_STEP_CASE(EXCENV_RK4)
    EXCENV_STEP_CASE(EXCENV_TSIT5)
    default: set_error("bad solver id %d", sc.solver); return EXCENV_EINVAL;
  }
#undef EXCENV_STEP_CASE
#undef EXCENV_STEP_LAUNCH
  return check_launch("excenv_step");
}

// Threads per workgroup of the plain lean trajectory kernel (kernels.hpp, NT): 1024 with one barrier per row for the Euler kernels of
// the small models, BLOCK everywhere else — measured per workload (profiles/r04_pattern_sweep.md): pendulum Euler fp32 -10.6 %, fp64
// -8 %, MSD Euler fp32 -7 %, fp64 -6 %, tank Euler fp32 -6 % (fp64 +3 %: not taken); RK4 / Tsit5 of the same models +3 ... +9 %,
// cart-pole / acrobot Euler within 3 % either way, PMSM (256 registers) not possible.
#ifndef EXCENV_ROW_SYNC_MIN_BATCH
#define EXCENV_ROW_SYNC_MIN_BATCH ((int64_t)1 << 17)
#endif
constexpr int64_t ROW_SYNC_MIN_BATCH = EXCENV_ROW_SYNC_MIN_BATCH;
static inline int row_sync_mode() {  // EXCENV_ROW_SYNC = 0: off, 1: barrier only, default 2: rows through LDS where possible
  static const int mode = [] { const char* e = std::getenv("EXCENV_ROW_SYNC"); return (e && e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : 2; }();
  return mode;
}
static inline bool row_sync_enabled() { return row_sync_mode() != 0; }
static inline bool wide_enabled() {  // EXCENV_WIDE=0: the 256-thread form everywhere (A/B measurements, counter passes)
  static const int on = [] { const char* e = std::getenv("EXCENV_WIDE"); return (e && e[0] == '0') ? 0 : 1; }();
  return on != 0;
}
constexpr int WIDE_THREADS = 1024;
constexpr int64_t WIDE_MIN_WORKGROUPS = 256;  // at least one wide workgroup per CU of the MI355X, else the narrow form fills the chip better
template <class M, typename T> constexpr bool sim_wide_ok(int solver) {
  return solver == EXCENV_EULER && !M::HAS_LUT &&
         (M::ID == EXCENV_PENDULUM || M::ID == EXCENV_MASS_SPRING_DAMPER || (M::ID == EXCENV_FLUID_TANK && sizeof(T) == 4));
}

// with the gym outputs' code the fp64 pendulum instantiations need 146 ... 150 registers: they would spill under the 1024-thread bound
// (fp32: 118 ... 123 since round 5 — wide like its plain launch: 2.43 -> 2.2 ms for the gym trajectories of B = 2^22, K = 100)
template <class M, typename T> constexpr bool sim_wide_gym_ok(int solver) {
  return sim_wide_ok<M, T>(solver) && (M::ID != EXCENV_PENDULUM || sizeof(T) == 4);
}

template <class M, typename T, int SOLVER, bool AHEAD> static void launch_sim_v(const SimCall& sc_in, const SimArgs<T, M>& ka_in,
                                                                                 bool general, int V, bool aem = false, bool lgym = false,
                                                                                 int nt = BLOCK) {
  SimArgs<T, M> ka = ka_in;
  SimCall sc = sc_in;
  sc.lds_pad = (int)lut_lds_bytes<T, M>(ka.kp, (size_t)sc_in.lds_pad);
  const int64_t lanes = sc.B / V;
  const dim3 grid((unsigned)((lanes + nt - 1) / nt)), block(nt);
  if constexpr (sim_wide_ok<M, T>(SOLVER)) {
    if (nt == WIDE_THREADS && !lgym) {  // the widest lean form in 1024-thread workgroups, one barrier per row (launch_sim decides)
      constexpr int VA = 16 / (int)sizeof(T);
      if (ka.straj[0] == nullptr) EXCENV_LAUNCH_DYN((sim_ahead_kernel<M, T, SOLVER, AHEAD, false, VA, 0, false, false, false, WIDE_THREADS>), grid, block, (size_t)sc_in.lds_pad, sc.stream, ka);
      else EXCENV_LAUNCH_DYN((sim_ahead_kernel<M, T, SOLVER, AHEAD, false, VA, 1, false, false, false, WIDE_THREADS>), grid, block, (size_t)sc_in.lds_pad, sc.stream, ka);
      return;
    }
  }
  if constexpr (!M::HAS_LUT) {
    if (lgym) {  // the gym trajectories out of the widest lean form (V == 16 / sizeof(T)); half of it (two environments per lane in
      // fp32) was built and measured in round 4: PMSM 5.80 -> 7.26 ms, cart-pole 3.55 -> 4.34, acrobot 3.55 -> 4.11 — removed
      constexpr int VA = 16 / (int)sizeof(T);
      if constexpr (sim_wide_gym_ok<M, T>(SOLVER)) {
        if (nt == WIDE_THREADS) {
          if (ka.straj[0] == nullptr) EXCENV_LAUNCH_DYN((sim_ahead_kernel<M, T, SOLVER, AHEAD, false, VA, 0, false, false, true, WIDE_THREADS>), grid, block, (size_t)sc_in.lds_pad, sc.stream, ka);
          else EXCENV_LAUNCH_DYN((sim_ahead_kernel<M, T, SOLVER, AHEAD, false, VA, 1, false, false, true, WIDE_THREADS>), grid, block, (size_t)sc_in.lds_pad, sc.stream, ka);
          return;
        }
      }
      if (ka.straj[0] == nullptr) EXCENV_LAUNCH_DYN((sim_ahead_kernel<M, T, SOLVER, AHEAD, false, VA, 0, false, false, true>), grid, block, (size_t)sc_in.lds_pad, sc.stream, ka);
      else EXCENV_LAUNCH_DYN((sim_ahead_kernel<M, T, SOLVER, AHEAD, false, VA, 1, false, false, true>), grid, block, (size_t)sc_in.lds_pad, sc.stream, ka);
      return;
    }
  }
  if constexpr (!M::HAS_LUT && (16 / (int)sizeof(T)) % M::A == 0) {
    if (aem) {  // row-major actions through the per-wave LDS piece ring: V == 16 / sizeof(T) (aem_applies)
      constexpr int VA = 16 / (int)sizeof(T);
      const size_t lds = aem_lds_bytes<M, T, VA>() + (size_t)sc_in.lds_pad;
      if (ka.straj[0] == nullptr) EXCENV_LAUNCH_DYN((sim_ahead_kernel<M, T, SOLVER, AHEAD, false, VA, 0, false, true>), grid, block, lds, sc.stream, ka);
      else EXCENV_LAUNCH_DYN((sim_ahead_kernel<M, T, SOLVER, AHEAD, false, VA, 1, false, true>), grid, block, lds, sc.stream, ka);
      return;
    }
  }
  // look-up models: one instantiation per place the tables live in (LDS when they fit, lut_lds_bytes above)
#define EXCENV_SIM_LAUNCH(GEN, VV, ST)                                                                                          \
  do {                                                                                                                          \
    if constexpr (M::HAS_LUT) {                                                                                                 \
      if (ka.kp.lut_lds) {                                                                                                      \
        EXCENV_LAUNCH_DYN((sim_ahead_kernel<M, T, SOLVER, AHEAD, GEN, VV, ST, true>), grid, block, (size_t)sc.lds_pad, sc.stream, ka); \
        return;                                                                                                                 \
      }                                                                                                                         \
    }                                                                                                                           \
    EXCENV_LAUNCH_DYN((sim_ahead_kernel<M, T, SOLVER, AHEAD, GEN, VV, ST, false>), grid, block, (size_t)sc.lds_pad, sc.stream, ka); \
    return;                                                                                                                     \
  } while (0)
  if (general) {
    if (ka.truncated == nullptr) EXCENV_SIM_LAUNCH(true, 1, -2);  // no gym trajectories: the instantiation without their code
    EXCENV_SIM_LAUNCH(true, 1, -1);
  }
  if (ka.straj[0] == nullptr) {  // observations only: its own instantiations (no state stores between the action loads and their waits)
    if constexpr (sizeof(T) == 4) {
      if (V == 4) EXCENV_SIM_LAUNCH(false, 4, 0);
    }
    if (V == 2) EXCENV_SIM_LAUNCH(false, 2, 0);
    EXCENV_SIM_LAUNCH(false, 1, 0);
  }
  if constexpr (sizeof(T) == 4) {
    if (V == 4) EXCENV_SIM_LAUNCH(false, 4, 1);
  }
  if (V == 2) EXCENV_SIM_LAUNCH(false, 2, 1);
  EXCENV_SIM_LAUNCH(false, 1, 1);
#undef EXCENV_SIM_LAUNCH
}

template <class M, typename T> static int launch_sim(const SimCall& sc) {
  SimArgs<T, M> ka;
  std::memset(&ka, 0, sizeof(ka));
  const bool batched = fill_props<T, M>(ka.kp, sc.props);
  double coef;
  if (int rc = pmsm_coef<M>(sc.props, sc.env_tau, &coef, sc.semantics == EXCENV_SEM_AHEAD)) return rc;
  if (M::IS_PMSM && sc.substeps != 1) {
    set_error("PMSM: obs_stepsize must equal action_stepsize (reference pmsm_env.py:787)");
    return EXCENV_EUNSUPPORTED;
  }
  ka.B = sc.B;
  ka.K = sc.K;
  ka.substeps = sc.substeps;
  ka.n_control = sc.control ? sc.control->n_control : 0;
  const int64_t N = sc.K * sc.substeps;
  const int64_t OW = M::O + ka.n_control;
  const bool with_gym = sc.gym != nullptr;
  // control_state columns alone (broadcast properties, no gym outputs, lane-major / tiled trajectories) do not need the
  // one-environment-per-lane GENERAL kernel: they are constant along the trajectory and are filled by control_fill_kernel
  // after the lean kernel has written everything else (same bytes, +1 launch, 0.52 -> 0.7 of the HBM roof at B = 2^22)
  // the gym trajectories come out of the widest lean form too (kernels.hpp, LGYM) when everything is lane-major, the batch
  // runs that form anyway and the flag / reward / reference arrays allow vector accesses
  constexpr int VMAXG = 16 / (int)sizeof(T);
  bool traj_aligned = aligned16(sc.obs_traj);  // the trajectory arrays alone (row_sync == 2 below)
  bool vec_ok = true;  // every pointer 16-byte aligned; the general instantiation stays at one environment per lane
  for (int j = 0; j < M::S; ++j) {
    if (!sc.state_in[j] || !sc.last_state[j]) { set_error("excenv_sim_ahead: state pointer %d is NULL", j); return EXCENV_ENULL; }
    ka.state_in[j] = (const T*)sc.state_in[j];
    ka.last_state[j] = (T*)sc.last_state[j];
    ka.straj[j] = sc.state_traj ? (T*)sc.state_traj[j] : nullptr;
    if (sc.state_traj && !sc.state_traj[j]) { set_error("excenv_sim_ahead: state_traj pointer %d is NULL", j); return EXCENV_ENULL; }
    vec_ok &= aligned16(ka.state_in[j]) && aligned16(ka.last_state[j]) && aligned16(ka.straj[j]);
    traj_aligned &= aligned16(ka.straj[j]);
  }
  // Everything that keeps a call from the widest lane form keeps it from the lean gym form too, and the call then takes the general
  // instantiation like any other gym call (round 4 returned an "internal error" for two such cases: acrobot RK4 / Tsit5 with default
  // options — the lane-width cap below — and state / trajectory / action arrays that are not 16-byte aligned).
  bool lean_gym = with_gym && !batched && !M::HAS_LUT && ka.n_control <= M::S && sc.action_layout == EXCENV_LAYOUT_LANE_MAJOR &&
                  sc.traj_layout == EXCENV_LAYOUT_LANE_MAJOR && sc.B > 0 && (sc.B % VMAXG) == 0 &&
                  (sc.vec_pref > 0 ? sc.vec_pref == VMAXG : widest_form_pays(sc.B, VMAXG)) &&
                  vec_ok && aligned16(sc.actions) && aligned16(sc.obs_traj) &&
                  !(sc.vec_pref == 0 && M::ID == EXCENV_ACROBOT && sc.solver != EXCENV_EULER) &&
                  aligned16(sc.gym->reward) && ((uintptr_t)sc.gym->terminated % VMAXG) == 0 && ((uintptr_t)sc.gym->truncated % VMAXG) == 0;
  for (int j = 0; lean_gym && j < ka.n_control; ++j) lean_gym = sc.control->reference[j] != nullptr && aligned16(sc.control->reference[j]);
  // the four-leaf models in fp64 with an RK solver would need more than 256 registers in that form (one wave per SIMD): general
  if (sizeof(T) == 8 && M::S == 4 && sc.solver != EXCENV_EULER) lean_gym = false;
  // row-major actions the lean kernel can read itself (AEM) are no reason for the general kernel either: the control columns are
  // filled behind it all the same (round 5; before, a control_state alone sent a plain [B, K, A] call to the transposition pass)
  const bool aem_candidate = !batched && !with_gym && vec_ok && aligned16(sc.obs_traj) &&
                             aem_applies(M::ID, M::HAS_LUT, false, M::A, sizeof(T), sc.B, sc.K, sc.solver, sc.vec_pref, sc.action_layout,
                                         sc.traj_layout, sc.flags, sc.actions);
  bool split_control = !batched && (!with_gym || lean_gym) && ka.n_control > 0 && sc.traj_layout != EXCENV_LAYOUT_ENV_MAJOR &&
                       (sc.action_layout != EXCENV_LAYOUT_ENV_MAJOR || aem_candidate) && sc.B > 0;
  if (split_control) {
    for (int j = 0; j < ka.n_control; ++j) split_control &= sc.control->reference[j] != nullptr;
  }
  bool general = batched || (ka.n_control > 0 && !split_control) || (with_gym && !lean_gym);
  ka.actions = (const T*)sc.actions;
  ka.obs = (T*)sc.obs_traj;
  constexpr int64_t TILE = EXCENV_TILE;  // envs per tile of the tiled layout (== one workgroup at V = TILE/BLOCK)
  if (sc.action_layout == EXCENV_LAYOUT_ENV_MAJOR) { ka.a_sb = sc.K * M::A; ka.a_sk = M::A; ka.a_sc = 1; }
  else if (sc.action_layout == EXCENV_LAYOUT_TILED) { ka.a_sb = 1; ka.a_sk = (int64_t)M::A * TILE; ka.a_sc = TILE; }
  else { ka.a_sb = 1; ka.a_sk = (int64_t)M::A * sc.B; ka.a_sc = sc.B; }
  if (sc.traj_layout == EXCENV_LAYOUT_ENV_MAJOR) {
    ka.o_sb = (N + 1) * OW; ka.o_sk = OW; ka.o_sc = 1;
    ka.s_sb = N + 1; ka.s_sk = 1;
  } else if (sc.traj_layout == EXCENV_LAYOUT_TILED) {
    ka.o_sb = 1; ka.o_sk = OW * TILE; ka.o_sc = TILE;
    ka.s_sb = 1; ka.s_sk = TILE;
  } else {
    ka.o_sb = 1; ka.o_sk = OW * sc.B; ka.o_sc = sc.B;
    ka.s_sb = 1; ka.s_sk = sc.B;
  }
  const bool tiled_a = sc.action_layout == EXCENV_LAYOUT_TILED, tiled_t = sc.traj_layout == EXCENV_LAYOUT_TILED;
  if ((tiled_a || tiled_t) && (sc.B % TILE) != 0) {
    set_error("excenv_sim_ahead: the tiled layout needs batch_size %% %lld == 0", (long long)TILE);
    return EXCENV_EINVAL;
  }
  if (with_gym) {
    if (!sc.gym->reward || !sc.gym->terminated || !sc.gym->truncated) {
      set_error("excenv_sim_ahead: gym trajectories need all of reward, terminated and truncated");
      return EXCENV_ENULL;
    }
    if (tiled_t) { set_error("excenv_sim_ahead: gym trajectories are not available in the tiled layout"); return EXCENV_EUNSUPPORTED; }
    const int64_t TW = (M::IS_PMSM || M::ID == EXCENV_FLUID_TANK) ? 1 : OW;
    ka.reward = (T*)sc.gym->reward;
    ka.terminated = sc.gym->terminated;
    ka.truncated = sc.gym->truncated;
    if (sc.traj_layout == EXCENV_LAYOUT_ENV_MAJOR) {
      ka.g_sb = N; ka.g_sk = 1;
      ka.t_sb = (N + 1) * TW; ka.t_sk = TW; ka.t_sc = 1;
    } else {
      ka.g_sb = 1; ka.g_sk = sc.B;
      ka.t_sb = TW; ka.t_sk = TW * sc.B; ka.t_sc = 1;  // lane-major flags: [row][B][TW], an environment's flags adjacent (ABI 7)
    }
  }
  for (int j = 0; j < ka.n_control; ++j) {
    ka.control_idx[j] = sc.control->control_idx[j];
    ka.reference[j] = (const T*)sc.control->reference[j];
  }
  ka.dt = (T)sc.obs_stepsize;
  ka.env_tau = (T)sc.env_tau;
  ka.adv_coef = (T)coef;
  ka.lin_stop = (T)(sc.env_tau * (double)(sc.K > 0 ? sc.K - 1 : 0));
  if (sc.B == 0) return EXCENV_OK;
  {  // per-lane offsets are 32-bit: 256 lanes * env stride * element size must stay below 2^31
    const int64_t lim = ((int64_t)1 << 31) / (BLOCK * (int64_t)sizeof(T));
    if (ka.a_sb >= lim || ka.o_sb >= lim || ka.s_sb >= lim) {
      set_error("excenv_sim_ahead: env-major trajectory too long for one call ((N+1)*O must be < %lld); chunk K",
                (long long)lim);
      return EXCENV_EUNSUPPORTED;
    }
  }

  if (sc.em_mode >= 2 && !em_fused_eligible(0, sc.action_layout, sc.traj_layout, sc.substeps, sc.gym != nullptr, M::A, M::S, M::O,
                                            sizeof(T))) {  // em_mode >= 2 is produced by excenv_sim_ahead_ws only; never trust it blindly
    set_error("excenv_sim_ahead: internal error: fused env-major kernel selected for an ineligible call");
    return EXCENV_EINVAL;
  }
  if constexpr (emr_supported<M, T>()) {
  if ((sc.em_mode == 2 || sc.em_mode == 4) && !general && emr_enabled()) {
    // register-ring form (kernels_emr.hpp): whole-line stores. Needs 128-byte aligned trajectory arrays and enough environments
    // to fill waves whose lanes are P environments apart.
    const bool ahead = sc.semantics == EXCENV_SEM_AHEAD;
    const int64_t W = ahead ? emr_rows<M, T, true>() : emr_rows<M, T, false>();  // steps per window
    auto period = [](int64_t x, int64_t m) { int64_t g = m, y = x % m; while (y) { const int64_t t = g % y; g = y; y = t; } return m / g; };
    const int64_t P = period(sc.K + 1, W);
    // action rows must consist of whole 16-byte pieces (they are fetched as 64-byte windows by LDS-direct loads)
    bool ok = ((uintptr_t)ka.obs % 128) == 0 && (sc.em_mode == 4 || sc.B >= 16 * EM_LANES * P) &&
              (sc.K * M::A * (int64_t)sizeof(T)) % 16 == 0 && aligned16(ka.actions) &&
              EM_LANES * P * (sc.K + 1) * M::O * (int64_t)sizeof(T) < ((int64_t)1 << 31);  // 32-bit lane offsets
    for (int j = 0; j < M::S; ++j) ok &= ka.straj[j] == nullptr || ((uintptr_t)ka.straj[j] % 128) == 0;
    if (ok) {
      SimArgs<T, M> kr = ka;
      kr.a_wg = P;
      const size_t emr_lds = ahead ? emr_lds_bytes<M, T, true>() : emr_lds_bytes<M, T, false>();
      const int64_t per = EM_LANES * P;
      const dim3 grid((unsigned)(((sc.B + per - 1) / per) * P)), block(EM_LANES);
#define EXCENV_EMR_CASE(SOLV)                                                                                                 \
  case SOLV:                                                                                                                  \
    if (sc.semantics == EXCENV_SEM_AHEAD) EXCENV_LAUNCH_DYN((sim_ahead_emr_kernel<M, T, SOLV, true>), grid, block, emr_lds, sc.stream, kr); \
    else EXCENV_LAUNCH_DYN((sim_ahead_emr_kernel<M, T, SOLV, false>), grid, block, emr_lds, sc.stream, kr);             \
    break;
      switch (sc.solver) {
        EXCENV_EMR_CASE(EXCENV_EULER)
        EXCENV_EMR_CASE(EXCENV_RK4)
        EXCENV_EMR_CASE(EXCENV_TSIT5)
        default: set_error("bad solver id %d", sc.solver); return EXCENV_EINVAL;
      }
#undef EXCENV_EMR_CASE
      g_last_launch = "sim_ahead_emr_kernel";
      return check_launch("excenv_sim_ahead (env-major fused, register ring)");
    }
  }
  }
  if (sc.em_mode >= 2) {  // decided by the caller (em_fused_eligible): fused env-major kernel, one wave per 64 envs,
                          // TK steps staged in LDS, per-env contiguous runs written out
    const size_t lds = em_lds_elems<T>(M::A, M::S, M::O) * sizeof(T);
    const dim3 grid((unsigned)((sc.B + EM_LANES - 1) / EM_LANES)), block(EM_LANES);
#define EXCENV_EM_CASE(SOLV)                                                                                             \
  case SOLV:                                                                                                             \
    if (sc.semantics == EXCENV_SEM_AHEAD) {                                                                              \
      if (general) EXCENV_LAUNCH_DYN((sim_ahead_em_kernel<M, T, SOLV, true, true>), grid, block, lds, sc.stream, ka);   \
      else EXCENV_LAUNCH_DYN((sim_ahead_em_kernel<M, T, SOLV, true, false>), grid, block, lds, sc.stream, ka);          \
    } else {                                                                                                             \
      if (general) EXCENV_LAUNCH_DYN((sim_ahead_em_kernel<M, T, SOLV, false, true>), grid, block, lds, sc.stream, ka);  \
      else EXCENV_LAUNCH_DYN((sim_ahead_em_kernel<M, T, SOLV, false, false>), grid, block, lds, sc.stream, ka);         \
    }                                                                                                                    \
    break;
    switch (sc.solver) {
      EXCENV_EM_CASE(EXCENV_EULER)
      EXCENV_EM_CASE(EXCENV_RK4)
      EXCENV_EM_CASE(EXCENV_TSIT5)
      default: set_error("bad solver id %d", sc.solver); return EXCENV_EINVAL;
    }
#undef EXCENV_EM_CASE
    g_last_launch = general ? "sim_ahead_em_kernel (general)" : "sim_ahead_em_kernel";
    return check_launch("excenv_sim_ahead (env-major fused)");
  }
  const bool aem = vec_ok && aligned16(ka.obs) &&
                   aem_applies(M::ID, M::HAS_LUT, general, M::A, sizeof(T), sc.B, sc.K, sc.solver, sc.vec_pref, sc.action_layout,
                               sc.traj_layout, sc.flags, ka.actions);
  vec_ok &= (sc.action_layout != EXCENV_LAYOUT_ENV_MAJOR || aem) && (sc.traj_layout != EXCENV_LAYOUT_ENV_MAJOR);
  vec_ok &= aligned16(ka.actions) && aligned16(ka.obs);
  constexpr int VMAX = 16 / (int)sizeof(T);
  int V = 1;
  if (general) vec_ok = false;  // one environment per lane (two, each with its own property set, measured no faster: DESIGN.md §4.1)
  if (vec_ok) {
    int want = sc.vec_pref > 0 ? sc.vec_pref : ((aem || lean_gym) ? VMAX : auto_envs_per_lane(sc.B, VMAX));
    // acrobot RK4 / Tsit5 is VALU-bound with the largest register footprint of all instantiations: two envs per lane keep
    // a third wave per SIMD resident (measured +7 % over four)
    if (sc.vec_pref == 0 && M::ID == EXCENV_ACROBOT && sc.solver != EXCENV_EULER && want > 2) want = 2;
    // look-up models: the interpolation code per environment is large (instruction cache) and keeps six table values per
    // environment live across the step (V = 4 needs > 256 registers): measured best at two environments per lane for Euler
    // and one for RK4 / Tsit5
    if (sc.vec_pref == 0 && M::HAS_LUT) {
      const int cap = (sc.solver == EXCENV_EULER) ? 2 : 1;
      if (want > cap) want = cap;
    }
    // PMSM observations only in fp32: the arithmetic of a step is the full launch's, the bytes are 40 of 68 — VALU floor and memory
    // floor meet (2.7 / 2.8 ms) and what counts is how well they overlap: two environments per lane (116 registers, four waves
    // per SIMD) instead of four (186, two waves). Same-buffers A/B: Euler 3.555 -> 3.157 ms (0.59 -> 0.66 of the roof), RK4 4.457 ->
    // 4.010, Tsit5 5.435 -> 4.784; with full outputs four stay faster (RK4 5.64 vs 6.19, Tsit5 6.25 vs 6.37).
    // Round 5, after the instruction diet (same-buffers A/B, one / two / four per lane): Euler 3.34 / 3.49 / 3.46 ms — one; RK4 3.79 /
    // 3.72 / 4.04 and Tsit5 4.32 / 4.19 / 4.73 — two.
    if (sc.vec_pref == 0 && M::IS_PMSM && !M::HAS_LUT && sizeof(T) == 4 && ka.straj[0] == nullptr && !aem && !lean_gym) {
      const int cap = (sc.solver == EXCENV_EULER) ? 1 : 2;
      if (want > cap) want = cap;
    }
    // cart-pole RK4 / Tsit5 and pendulum Tsit5 in fp32: the same trade (registers for a resident wave) — same-buffers A/B with two
    // instead of four environments per lane: cart-pole RK4 4.646 -> 4.323 ms, Tsit5 7.046 -> 6.091, pendulum Tsit5 2.637 -> 2.477
    // (pendulum RK4, mass-spring-damper, tank: four stay faster or equal)
    if (sc.vec_pref == 0 && sizeof(T) == 4 && !aem && !lean_gym && want > 2 &&
        ((M::ID == EXCENV_CART_POLE && sc.solver != EXCENV_EULER) || (M::ID == EXCENV_PENDULUM && sc.solver == EXCENV_TSIT5)))
      want = 2;
    if (want > VMAX) want = VMAX;
    while (want > 1 && (sc.B % want) != 0) want >>= 1;
    V = want;
  }
  if (lean_gym && (general || V != VMAX)) { set_error("excenv_sim_ahead: internal error: lean gym outputs need %d environments per lane", VMAX); return EXCENV_EINVAL; }
  if (aem && V != VMAX) { set_error("excenv_sim_ahead: internal error: fused row-major actions need %d environments per lane", VMAX); return EXCENV_EINVAL; }
  if (tiled_a || tiled_t) {  // a workgroup must not straddle tiles
    constexpr int VT = (int)(TILE / BLOCK);
    if (VT > VMAX || general || !vec_ok) {
      if (TILE % BLOCK != 0) { set_error("tiled layout: TILE %% BLOCK != 0"); return EXCENV_EINVAL; }
      V = 1;
    } else {
      V = VT;
    }
  }
  // one environment per lane at a batch that fills the chip several times over: the four waves of a workgroup store each row
  // together (kernels.hpp row_sync). EXCENV_ROW_SYNC=0 switches it off (A/B measurements).
  // (not with the gym outputs' code in the loop: with that much arithmetic per row lockstep costs more than the stores gain —
  // PMSM 7.06 -> 8.59 ms, pendulum 4.64 -> 5.28, acrobot 7.0 -> 8.0 measured)
  ka.row_sync = (V == 1 && !with_gym && sc.traj_layout == EXCENV_LAYOUT_LANE_MAJOR && sc.B >= ROW_SYNC_MIN_BATCH && row_sync_enabled()) ? 1 : 0;
  size_t row_lds = 0;
  if (ka.row_sync && !M::HAS_LUT && !aem && (sc.B % BLOCK) == 0 && traj_aligned && row_sync_mode() >= 2) {
    // whole workgroups and aligned arrays: the rows leave through LDS as 16-byte stores (kernels.hpp, row_sync == 2)
    const size_t ns = (size_t)OW + (ka.straj[0] ? M::S : 0);
    const size_t bytes = 2 * ns * BLOCK * sizeof(T);
    if (bytes <= ((size_t)64 << 10)) { ka.row_sync = 2; row_lds = bytes; }
  }
  int nt = BLOCK;
  if (wide_enabled() && sim_wide_ok<M, T>(sc.solver) && (!lean_gym || sim_wide_gym_ok<M, T>(sc.solver)) && !general && !aem && !tiled_a && !tiled_t && V == VMAX &&
      sc.B / V >= WIDE_THREADS * WIDE_MIN_WORKGROUPS)
    nt = WIDE_THREADS;
  {  // element offset of workgroup w's first env in each stream
    const int64_t wg_envs = (int64_t)nt * V;
    auto wg_off = [&](int layout, int64_t sb, int64_t per_tile) -> int64_t {
      if (layout == EXCENV_LAYOUT_TILED) return (wg_envs == TILE) ? per_tile : -1;
      return wg_envs * sb;
    };
    ka.a_wg = wg_off(sc.action_layout, ka.a_sb, sc.K * M::A * TILE);
    ka.o_wg = wg_off(sc.traj_layout, ka.o_sb, (N + 1) * OW * TILE);
    ka.s_wg = wg_off(sc.traj_layout, ka.s_sb, (N + 1) * TILE);
    if (ka.a_wg < 0 || ka.o_wg < 0 || ka.s_wg < 0) {
      set_error("excenv_sim_ahead: tiled layout needs unbatched properties, no gym trajectories, 16-byte aligned buffers and the %d-byte dtype (control columns are filled by a second launch)", 4);
      return EXCENV_EUNSUPPORTED;
    }
  }
  SimCall scl = sc;
  scl.lds_pad += (int)row_lds;
#define EXCENV_SIM_CASE(SOLV)                                                         \
  case SOLV:                                                                          \
    if (sc.semantics == EXCENV_SEM_AHEAD) launch_sim_v<M, T, SOLV, true>(scl, ka, general, V, aem && V == VMAX, lean_gym, nt);  \
    else launch_sim_v<M, T, SOLV, false>(scl, ka, general, V, aem && V == VMAX, lean_gym, nt);                         \
    break;
  switch (sc.solver) {
    EXCENV_SIM_CASE(EXCENV_EULER)
    EXCENV_SIM_CASE(EXCENV_RK4)
    EXCENV_SIM_CASE(EXCENV_TSIT5)
    default: set_error("bad solver id %d", sc.solver); return EXCENV_EINVAL;
  }
#undef EXCENV_SIM_CASE
  g_last_launch = general ? "sim_ahead_kernel (general)" : (lean_gym ? (nt > BLOCK ? "sim_ahead_kernel (lean, gym outputs, 1024 threads)" : "sim_ahead_kernel (lean, gym outputs)") : aem ? "sim_ahead_kernel (row-major actions fused)" : (V == 1 ? "sim_ahead_kernel (V=1)" : (V == 2 ? (nt > BLOCK ? "sim_ahead_kernel (V=2, 1024 threads)" : "sim_ahead_kernel (V=2)") : (nt > BLOCK ? "sim_ahead_kernel (V=4, 1024 threads)" : "sim_ahead_kernel (V=4)"))));
  if (int rc = check_launch("excenv_sim_ahead")) return rc;
  if (split_control && !general) {
    ControlFillArgs<T, M> fa;
    std::memset(&fa, 0, sizeof(fa));
    fa.kp = ka.kp;
    fa.B = sc.B;
    fa.rows = N + 1;
    fa.n_control = ka.n_control;
    fa.tiled = tiled_t ? 1 : 0;
    for (int j = 0; j < ka.n_control; ++j) {
      fa.control_idx[j] = ka.control_idx[j];
      fa.reference[j] = ka.reference[j];
    }
    fa.obs = ka.obs;
    fa.o_sk = ka.o_sk;
    fa.o_sc = ka.o_sc;
    fa.tile_pitch = (N + 1) * OW * TILE;
    bool v4 = (sizeof(T) == 4) && (sc.B % 4 == 0) && aligned16(ka.obs);
    for (int j = 0; j < ka.n_control; ++j) v4 &= aligned16(fa.reference[j]);
    const int64_t lanes = v4 ? sc.B / 4 : sc.B;
    const dim3 fgrid((unsigned)((lanes + BLOCK - 1) / BLOCK), (unsigned)((N + 1 < 64) ? N + 1 : 64)), fblock(BLOCK);
    if (v4) {
      if constexpr (sizeof(T) == 4) hipLaunchKernelGGL((control_fill_kernel<M, T, 4>), fgrid, fblock, 0, sc.stream, fa);
    } else {
      hipLaunchKernelGGL((control_fill_kernel<M, T, 1>), fgrid, fblock, 0, sc.stream, fa);
    }
    return check_launch("excenv_sim_ahead (control columns)");
  }
  return EXCENV_OK;
}

template <class M, typename T> static int launch_traj_gym(const TrajGymCall& gc) {
  TrajGymArgs<T, M> ka;
  std::memset(&ka, 0, sizeof(ka));
  fill_props<T, M>(ka.kp, gc.props);
  ka.B = gc.B;
  ka.rows = gc.rows;
  ka.n_control = gc.control ? gc.control->n_control : 0;
  for (int j = 0; j < ka.n_control; ++j) {
    ka.control_idx[j] = gc.control->control_idx[j];
    ka.reference[j] = (const T*)gc.control->reference[j];
    ka.r_sb[j] = gc.ref_strides ? gc.ref_strides[2 * j] : 1;
    ka.r_sk[j] = gc.ref_strides ? gc.ref_strides[2 * j + 1] : 0;
  }
  for (int j = 0; j < M::S; ++j) {
    if (!gc.state_traj[j]) { set_error("excenv_rew_trunc_term: state_traj pointer %d is NULL", j); return EXCENV_ENULL; }
    ka.straj[j] = (const T*)gc.state_traj[j];
  }
  ka.s_sb = gc.s_sb;
  ka.s_sk = gc.s_sk;
  ka.reward = (T*)gc.reward;
  ka.terminated = gc.terminated;
  ka.truncated = gc.truncated;
  const int64_t N = gc.rows - 1;
  const int64_t TW = (M::IS_PMSM || M::ID == EXCENV_FLUID_TANK) ? 1 : M::O + ka.n_control;
  if (gc.out_layout == EXCENV_LAYOUT_ENV_MAJOR) {
    ka.g_sb = N; ka.g_sk = 1;
    ka.t_sb = gc.rows * TW; ka.t_sk = TW; ka.t_sc = 1;
  } else {
    ka.g_sb = 1; ka.g_sk = gc.B;
    ka.t_sb = TW; ka.t_sk = TW * gc.B; ka.t_sc = 1;  // [row][B][TW]
  }
  if (gc.B == 0 || gc.rows == 0) return EXCENV_OK;
  ka.fast_is_env = (gc.s_sb == 1 || gc.rows == 1) ? 1 : 0;  // lanes run along the contiguous index of the state arrays
  const int64_t nfast = ka.fast_is_env ? gc.B : gc.rows, nslow = ka.fast_is_env ? gc.rows : gc.B;
  const int64_t blocks = ((nfast + BLOCK - 1) / BLOCK) * nslow;
  if (blocks >= ((int64_t)1 << 31)) { set_error("excenv_rew_trunc_term: trajectory too large for one launch"); return EXCENV_EUNSUPPORTED; }
  hipLaunchKernelGGL((traj_gym_kernel<M, T>), dim3((unsigned)blocks), dim3(BLOCK), 0, gc.stream, ka);
  return check_launch("excenv_rew_trunc_term");
}

template <class M, typename T> static int launch_from_obs(const FromObsCall& fc) {
  FromObsArgs<T, M> ka;
  std::memset(&ka, 0, sizeof(ka));
  fill_props<T, M>(ka.kp, fc.props);
  ka.B = fc.B;
  ka.n_control = fc.n_control;
  ka.obs = (const T*)fc.obs;
  for (int j = 0; j < M::S; ++j) {
    if (!fc.state_out[j]) { set_error("excenv_state_from_observation: state_out pointer %d is NULL", j); return EXCENV_ENULL; }
    ka.state_out[j] = (T*)fc.state_out[j];
  }
  for (int j = 0; j < fc.n_control; ++j) {
    if (!fc.reference_out || !fc.reference_out[j]) { set_error("excenv_state_from_observation: reference_out pointer %d is NULL", j); return EXCENV_ENULL; }
    ka.control_idx[j] = fc.control_idx[j];
    ka.reference_out[j] = (T*)fc.reference_out[j];
  }
  if (fc.B == 0) return EXCENV_OK;
  hipLaunchKernelGGL((from_obs_kernel<M, T>), dim3((unsigned)((fc.B + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, fc.stream, ka);
  return check_launch("excenv_state_from_observation");
}

template <class M, typename T> static int launch_observe(const ObserveCall& oc) {
  ObserveArgs<T, M> ka;
  std::memset(&ka, 0, sizeof(ka));
  fill_props<T, M>(ka.kp, oc.props);
  ka.B = oc.B;
  ka.n_control = oc.control ? oc.control->n_control : 0;
  ka.obs = (T*)oc.obs;
  for (int j = 0; j < M::S; ++j) {
    if (!oc.state[j]) { set_error("excenv_observe: state pointer %d is NULL", j); return EXCENV_ENULL; }
    ka.state[j] = (const T*)oc.state[j];
  }
  for (int j = 0; j < ka.n_control; ++j) {
    if (!oc.control->reference[j]) { set_error("excenv_observe: reference pointer %d is NULL", j); return EXCENV_ENULL; }
    ka.control_idx[j] = oc.control->control_idx[j];
    ka.reference[j] = (const T*)oc.control->reference[j];
  }
  if (oc.B == 0) return EXCENV_OK;
  hipLaunchKernelGGL((observe_kernel<M, T>), dim3((unsigned)((oc.B + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, oc.stream, ka);
  return check_launch("excenv_observe");
}

template <class M, typename T> static int launch_update_ref(const RefGenCall& rc) {
  RefGenArgs<T, M> ka;
  std::memset(&ka, 0, sizeof(ka));
  fill_props<T, M>(ka.kp, rc.props);
  ka.B = rc.B;
  ka.n_control = rc.n_control;
  for (int j = 0; j < rc.n_control; ++j) {
    if (!rc.reference[j]) { set_error("excenv_update_ref: reference pointer %d is NULL", j); return EXCENV_ENULL; }
    ka.control_idx[j] = rc.control_idx[j];
    ka.reference[j] = (T*)rc.reference[j];
    ka.reference_in[j] = rc.reference_in ? (const T*)rc.reference_in[j] : (const T*)rc.reference[j];
    if (!ka.reference_in[j]) { set_error("excenv_update_ref: reference_in pointer %d is NULL", j); return EXCENV_ENULL; }
  }
  ka.keys = rc.keys;
  ka.hold = rc.hold;
  ka.keys_in = rc.keys_in ? rc.keys_in : rc.keys;
  ka.hold_in = rc.hold_in ? rc.hold_in : rc.hold;
  ka.hold_min = rc.hold_min;
  ka.hold_max = rc.hold_max;
  if (rc.B == 0) return EXCENV_OK;
  hipLaunchKernelGGL((update_ref_kernel<M, T>), dim3((unsigned)((rc.B + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, rc.stream, ka);
  return check_launch("excenv_update_ref");
}

template <class M, typename T> static int launch_random_state(const RandomStateCall& rc) {
  RandomStateArgs<T, M> ka;
  std::memset(&ka, 0, sizeof(ka));
  fill_props<T, M>(ka.kp, rc.props);
  ka.B = rc.B;
  ka.keys = rc.keys;
  ka.key_leaf = rc.key_leaf;
  for (int j = 0; j < M::S; ++j) {
    if (!rc.state_out[j]) { set_error("excenv_random_state: state_out pointer %d is NULL", j); return EXCENV_ENULL; }
    ka.state_out[j] = (T*)rc.state_out[j];
  }
  if (rc.B == 0) return EXCENV_OK;
  hipLaunchKernelGGL((random_state_kernel<M, T>), dim3((unsigned)((rc.B + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, rc.stream, ka);
  return check_launch("excenv_random_state");
}

template <template <typename> class MT> struct EnvEntry {
  static int step(const StepCall& sc) {
    return sc.dtype == EXCENV_F32 ? launch_step<MT<float>, float>(sc) : launch_step<MT<double>, double>(sc);
  }
  static int sim(const SimCall& sc) {
    return sc.dtype == EXCENV_F32 ? launch_sim<MT<float>, float>(sc) : launch_sim<MT<double>, double>(sc);
  }
  static int traj_gym(const TrajGymCall& gc) {
    return gc.dtype == EXCENV_F32 ? launch_traj_gym<MT<float>, float>(gc) : launch_traj_gym<MT<double>, double>(gc);
  }
  static int from_obs(const FromObsCall& fc) {
    return fc.dtype == EXCENV_F32 ? launch_from_obs<MT<float>, float>(fc) : launch_from_obs<MT<double>, double>(fc);
  }
  static int update_ref(const RefGenCall& rc) {
    return rc.dtype == EXCENV_F32 ? launch_update_ref<MT<float>, float>(rc) : launch_update_ref<MT<double>, double>(rc);
  }
  static int random_state(const RandomStateCall& rc) {
    return rc.dtype == EXCENV_F32 ? launch_random_state<MT<float>, float>(rc) : launch_random_state<MT<double>, double>(rc);
  }
  static int observe(const ObserveCall& oc) {
    return oc.dtype == EXCENV_F32 ? launch_observe<MT<float>, float>(oc) : launch_observe<MT<double>, double>(oc);
  }
  static EnvVTable vtable() {
    return EnvVTable{MT<float>::S, MT<float>::A, MT<float>::O, MT<float>::P, &step, &sim, &traj_gym, &from_obs, &update_ref,
                     &random_state, &observe};
  }
};

}  // namespace excenv
