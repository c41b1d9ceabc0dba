/*
 * excenv.h — C ABI of the MI355X-native batched ODE stepper behind
 * exciting-environments' vmap_step / vmap_sim_ahead hot path.
 *
 * The reference has no FFI of its own (it is pure Python on JAX/diffrax); the
 * boundary it exposes is the object returned by EnvironmentRegistry.X.make()
 * (reference exciting_environments/registration.py:21-32). Each entry point below
 * names the reference method whose arithmetic it replaces. All pointers are raw
 * device pointers (hipMalloc'ed / torch CUDA storage), sizes are plain integers,
 * `stream` is a hipStream_t passed as void*. No function throws, allocates device
 * memory, or synchronises the device: they only enqueue kernels on `stream`
 * (hipGraph-capturable). Return value: 0 on success, a negative EXCENV_E* code
 * otherwise; excenv_last_error() gives the per-thread message.
 *
 * Data layout (struct-of-arrays): every state field / per-env parameter is its
 * own contiguous [B] array. Trajectories come in three layouts:
 *   EXCENV_LAYOUT_ENV_MAJOR  : element (b,k,c) at ((b*K)+k)*C + c  — the reference's
 *                              row-major jnp arrays actions[B,K,A], observations[B,K+1,O],
 *                              state leaves [B,K+1] (core_env.py:571-616).
 *   EXCENV_LAYOUT_LANE_MAJOR : element (b,k,c) at ((k*C)+c)*B + b  — one lane per env,
 *                              lane-adjacent envs address-adjacent (fully coalesced).
 *   EXCENV_LAYOUT_TILED      : lane-major inside tiles of EXCENV_TILE envs: element (b,k,c) at
 *                              (b/TILE)*K*C*TILE + ((k*C)+c)*TILE + b%TILE — every workgroup owns one tile and
 *                              writes ONE sequential stream (fp32, batch_size % TILE == 0, unbatched properties).
 */
#ifndef EXCENV_H
#define EXCENV_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 4: v3 + excenv_random_state, excenv_update_ref_to, excenv_observe (additions only; every v3 signature is unchanged)
 * 5: v4 + excenv_stream_pattern (addition only)
 * 6: v5 + excenv_launch_opts_t.flags (the former `reserved` field; 0 keeps the old meaning); excenv_sim_ahead reads row-major
 *    actions inside the lane-major trajectory kernel (no workspace needed for that combination); additions:
 *    excenv_sim_ahead_fuses_actions, excenv_last_launch, excenv_allgather
 * 7: the lane-major truncated-flag trajectory is [row][B][flag] (an environment's flags adjacent; was [row][flag][B]):
 *    excenv_traj_gym_t.truncated of excenv_sim_ahead and the `truncated` output of excenv_rew_trunc_term. Every signature
 *    is unchanged; the env-major layout [B][row][flag] — the reference's — is untouched */
#define EXCENV_ABI_VERSION 7

/* Environment ids. Field orders follow the reference dataclasses. */
typedef enum {
  EXCENV_PENDULUM = 0,           /* pendulum/pendulum_env.py:116-142   S=(theta,omega) A=(torque) P=(g,l,m) */
  EXCENV_MASS_SPRING_DAMPER = 1, /* mass_spring_damper_env.py:113-139  S=(deflection,velocity) A=(force) P=(d,k,m) */
  EXCENV_CART_POLE = 2,          /* cart_pole_env.py:126-157           S=(deflection,velocity,theta,omega) A=(force) P=(mu_p,mu_c,l,m_p,m_c,g) */
  EXCENV_ACROBOT = 3,            /* acrobot_env.py:135-169             S=(theta_1,theta_2,omega_1,omega_2) A=(torque) P=(g,l_1,l_2,m_1,m_2,l_c1,l_c2,I_1,I_2) */
  EXCENV_FLUID_TANK = 4,         /* fluid_tank_env.py:70-95            S=(height) A=(inflow) P=(base_area,orifice_area,c_d,g) */
  EXCENV_PMSM = 5,               /* pmsm/pmsm_env.py:269-305 (linear dq model) S=(u_d_buffer,u_q_buffer,epsilon,i_d,i_q,torque,omega_el)
                                    A=(u_d,u_q) P=(p,r_s,l_d,l_q,psi_p,u_dc,deadtime) O=(i_d,i_q,omega_el,torque,cos eps,sin eps,u_d_buffer,u_q_buffer) */
  EXCENV_NUM_ENVS = 6
} excenv_env_t;

/* Fixed-step explicit solvers (stand-ins for diffrax.Euler / diffrax.Tsit5 passed as
 * `solver=`; RK4 is a build-side extension). */
typedef enum { EXCENV_EULER = 0, EXCENV_RK4 = 1, EXCENV_TSIT5 = 2, EXCENV_NUM_SOLVERS = 3 } excenv_solver_t;

typedef enum { EXCENV_F32 = 0, EXCENV_F64 = 1 } excenv_dtype_t;

typedef enum { EXCENV_LAYOUT_ENV_MAJOR = 0, EXCENV_LAYOUT_LANE_MAJOR = 1, EXCENV_LAYOUT_TILED = 2 } excenv_layout_t;
#define EXCENV_TILE 1024 /* environments per tile of EXCENV_LAYOUT_TILED */

/* Trajectory semantics of excenv_sim_ahead.
 *   EXCENV_SEM_STEP  : the K steps are exactly K applications of excenv_step (post-processing —
 *                      angle wrap, tank clip, PMSM angle prediction — acts on the carried state).
 *                      This is the property the reference tests (tests/envs/test_core_functions.py:134-155).
 *   EXCENV_SEM_AHEAD : structure of the reference's _ode_solver_simulate_ahead (e.g.
 *                      pendulum_env.py:196-259, fluid_tank_env.py:156-216, pmsm_env.py:709-801): the raw
 *                      ODE state is carried un-wrapped / un-clipped, post-processing is applied to the SAVED
 *                      rows only, PMSM clips all actions with the predicted angle eps0 + k*tau*omega, and RK
 *                      stages with c_i == 1 read action k+1 (core_env.py:435-439). Step size is exactly
 *                      obs_stepsize and the action index is exactly floor(step / substeps) (diffrax's
 *                      accumulated-time rounding is deliberately not reproduced).
 */
typedef enum { EXCENV_SEM_STEP = 0, EXCENV_SEM_AHEAD = 1 } excenv_semantics_t;

#define EXCENV_MAX_STATE 8
#define EXCENV_MAX_ACTION 2
#define EXCENV_MAX_STATIC 9
#define EXCENV_MAX_CONTROL 8

/* A property leaf is either broadcast (per_env == NULL, `value` used) or batched
 * (per_env -> [B] array of the working dtype): reference core_env.py:253-277. */
typedef struct {
  double value;
  const void* per_env;
} excenv_param_t;

/* PMSM saturated model (pmsm_env.py:316-363, 487-507): the six look-up tables of the flux linkages / differential
 * inductances over the (i_d, i_q) grid, prepared as the reference's generate_interpolators_and_lut does (NaNs filled
 * by nearest neighbour, edges repeated once). All arrays are device arrays of the working dtype.
 *   grid_d [n_d], grid_q [n_q] : strictly increasing grid coordinates
 *   tables [n_d][n_q][8]       : (L_dd, L_dq, L_qd, L_qq, Psi_d, Psi_q, 0, 0) at each grid node */
typedef struct {
  int32_t n_d, n_q;
  const void* grid_d;
  const void* grid_q;
  const void* tables;
} excenv_pmsm_lut_t;

/* EnvProperties (core_env.py:245-251): static params in the field order listed at the env id,
 * min/max of physical_normalizations per state field and of action_normalizations per action.
 * pmsm_lut: NULL, or (EXCENV_PMSM only) the LUTs of the saturated model — selects nonlinear_ode /
 * currents_to_torque_saturated instead of the linear dq model (EnvProperties.saturated, pmsm_env.py:307-314). */
typedef struct {
  excenv_param_t static_params[EXCENV_MAX_STATIC];
  excenv_param_t state_min[EXCENV_MAX_STATE];
  excenv_param_t state_max[EXCENV_MAX_STATE];
  excenv_param_t action_min[EXCENV_MAX_ACTION];
  excenv_param_t action_max[EXCENV_MAX_ACTION];
  const excenv_pmsm_lut_t* pmsm_lut;
} excenv_props_t;

/* Reference-tracking columns of the observation (generate_observation appends the normalised
 * `state.reference.<name>` for every name in control_state, e.g. pendulum_env.py:322-328).
 * n_control == 0 => no extra columns. reference[j] is a [B] array for state field control_idx[j]. */
typedef struct {
  int32_t n_control;
  int32_t control_idx[EXCENV_MAX_CONTROL];
  const void* reference[EXCENV_MAX_CONTROL];
  /* excenv_gym_step only: NULL, or the [B] reference values the OBSERVATION columns show when they differ from the ones the
   * reward is computed against — GymWrapper.gym_step takes the observation before update_ref and the reward after it
   * (gym_wrapper.py:109-126), so in a step that redraws a reference the two differ. */
  const void* obs_reference[EXCENV_MAX_CONTROL];
} excenv_control_t;

/* Per-call launch options (no reference counterpart; NULL = all defaults). Everything that shapes a launch travels
 * with the call: the library keeps no mutable state besides the per-thread error string.
 *   envs_per_lane  : lane-major / tiled trajectories and the step path: 0 = auto (16-byte accesses when the batch is large
 *                    enough to fill the chip that way, else one env per lane), 1 / 2 / 4 = forced. Ignored (always 1) by
 *                    excenv_gym_step and by every call that needs the general instantiation (per-env property arrays,
 *                    gym trajectories, row-major buffers)
 *   env_major_mode : env-major (row-major) buffers — 0: a fused kernel when both layouts are env-major and substeps == 1
 *                    (the register-ring form for large batches of broadcast-property environments with 128-byte aligned
 *                    trajectory arrays, else the LDS-ring form); 1: never (workspace + transposes, or generic strides);
 *                    2: fused, LDS-ring form only; 3: fused, register-ring form whenever its preconditions hold
 *   lds_pad_bytes  : extra dynamic LDS per sim_ahead workgroup (caps resident workgroups per CU; occupancy experiments)
 *   flags          : bit set of EXCENV_OPT_* (0 = defaults; unknown bits are an error)
 *     EXCENV_OPT_NO_FUSED_ACTIONS : row-major actions with lane-major trajectories are transposed through the workspace
 *                                   (or read with generic strides) instead of being read by the trajectory kernel itself */
#define EXCENV_OPT_NO_FUSED_ACTIONS 1
typedef struct {
  int32_t envs_per_lane;
  int32_t env_major_mode;
  int32_t lds_pad_bytes;
  int32_t flags; /* ABI <= 5: `reserved`, had to be 0 */
} excenv_launch_opts_t;

/* Optional reward / terminated / truncated trajectories of excenv_sim_ahead — what
 * CoreEnvironment.vmap_generate_rew_trunc_term_ahead (core_env.py:490-531, 618-647) computes from the returned states,
 * produced by the same launch from the registers that hold each saved state. All three pointers or NULL struct.
 *   reward     : N rows (saved rows 1..N) of one value per env, working dtype
 *   terminated : N rows of one byte (0/1) per env
 *   truncated  : N+1 rows (saved rows 0..N) of excenv_truncated_width() bytes per env
 * in the trajectory layout of the call: env-major reward / terminated [B][row], truncated [B][row][flag]; lane-major reward /
 * terminated [row][B], truncated [row][B][flag] (the flags of an environment adjacent: a lane of the trajectory kernel writes
 * the flags of its environments with one or two stores per row). Tiled: unsupported. */
typedef struct {
  void* reward;
  uint8_t* terminated;
  uint8_t* truncated;
} excenv_traj_gym_t;

/* ---- introspection -------------------------------------------------------------------- */
int excenv_abi_version(void);
const char* excenv_last_error(void);
/* Name of the trajectory-kernel form the last excenv_sim_ahead[_ws] call of this thread enqueued ("" before the first):
 * "sim_ahead_kernel (V=1|V=2|V=4)", "sim_ahead_kernel (general)", "sim_ahead_kernel (row-major actions fused)",
 * "sim_ahead_emr_kernel", "sim_ahead_em_kernel[ (general)]", "transposition workspace + sim_ahead_kernel". Informational
 * (tests assert that the path they mean to check is the one that ran). */
const char* excenv_last_launch(void);
/* S = physical_state_dim, A = action_dim, O = observation width without control columns, P = #static params. */
int excenv_env_dims(int env, int32_t* S, int32_t* A, int32_t* O, int32_t* P);
/* Algorithmic HBM bytes per env-step: w*(S+A+S+O) for the step path,
 * w*(A+O[+S]) for sim_ahead. */
int64_t excenv_step_bytes(int env, int dtype);
int64_t excenv_sim_ahead_bytes(int env, int dtype, int with_state_traj);

/* ---- replaces CoreEnvironment.vmap_step (core_env.py:533-569) -------------------------
 * and, for PMSM, PMSM.step (pmsm_env.py:851-883).
 *   state_in  : S pointers to [B] arrays (physical_state fields, reference order)
 *   action    : [B][A] row-major normalised action
 *   state_out : S pointers to [B] arrays (may alias state_in element-for-element)
 *   obs       : [B][O + n_control] row-major
 */
int excenv_step(int env, int solver, int dtype, int64_t B,
                const excenv_props_t* props, const excenv_control_t* control, double tau,
                const void* const* state_in, const void* action,
                void* const* state_out, void* obs, const excenv_launch_opts_t* opts, void* stream);

/* ---- replaces GymWrapper.gym_step (gym_wrapper.py:88-130): vmap_step fused with the environment's
 * generate_reward / generate_terminated / generate_truncated (e.g. pendulum_env.py:297-309,381-390,
 * pmsm_env.py:972-1037) so the new state is not read a second time.
 *   reward     : [B] values of the working dtype (the reference's [B,1])
 *   terminated : [B] bytes (0/1)
 *   truncated  : [B][excenv_truncated_width(env, n_control)] bytes (0/1): |obs| > 1 per observation column; one flag
 *                for PMSM (|i_dq| > 1 in normalised units) and FluidTank (constant 0)
 * The reference-generator (update_ref / generate_new_ref, JAX Threefry) is not part of this entry point. */
int32_t excenv_truncated_width(int env, int32_t n_control);
int excenv_gym_step(int env, int solver, int dtype, int64_t B,
                    const excenv_props_t* props, const excenv_control_t* control, double tau,
                    const void* const* state_in, const void* action,
                    void* const* state_out, void* obs, void* reward, uint8_t* terminated, uint8_t* truncated,
                    const excenv_launch_opts_t* opts, void* stream);

/* ---- replaces CoreEnvironment.vmap_sim_ahead (core_env.py:571-616) --------------------
 * and PMSM.sim_ahead (pmsm_env.py:746-801). One persistent launch runs all N = K*substeps
 * solver steps of step size obs_stepsize; action k is held for `substeps` solver steps.
 *   actions    : K rows of A normalised actions per env, in `action_layout`
 *   obs_traj   : N+1 rows of (O + n_control) per env, in `traj_layout` (row 0 = init state)
 *   state_traj : NULL, or S pointers to [B][N+1] (env-major) / [N+1][B] (lane-major) arrays
 *   last_state : S pointers to [B] arrays (row N of the trajectory; may alias state_in)
 *   env_tau    : the environment's own tau; only PMSM reads it (its voltage-angle prediction uses
 *                self.tau, pmsm_env.py:599-604,719-722, while the solver steps by obs_stepsize).
 *                PMSM requires substeps == 1 (reference quirk, pmsm_env.py:787).
 */
int excenv_sim_ahead(int env, int solver, int dtype, int64_t B, int64_t K, int32_t substeps,
                     const excenv_props_t* props, const excenv_control_t* control, double obs_stepsize,
                     double env_tau, const void* const* state_in, const void* actions, int action_layout,
                     void* obs_traj, void* const* state_traj, int traj_layout,
                     void* const* last_state, int semantics, const excenv_traj_gym_t* gym,
                     const excenv_launch_opts_t* opts, void* stream);

/* Same as excenv_sim_ahead, with a caller-provided device workspace. When a layout is EXCENV_LAYOUT_ENV_MAJOR
 * (the reference's row-major arrays) and `workspace_bytes >= excenv_sim_ahead_workspace_bytes(...)`, the library
 * transposes the actions into the workspace, runs the coalesced lane-major kernel there and transposes the
 * trajectories back with an LDS-tiled kernel (3 launches + S small ones, all on `stream`); without a workspace the
 * generic-stride path of the same kernel is used (one lane per env, scattered words). Results are bit-identical. */
int64_t excenv_sim_ahead_workspace_bytes(int env, int dtype, int64_t B, int64_t K, int32_t substeps, int32_t n_control,
                                         int action_layout, int traj_layout, int with_state_traj);
int excenv_sim_ahead_ws(int env, int solver, int dtype, int64_t B, int64_t K, int32_t substeps,
                        const excenv_props_t* props, const excenv_control_t* control, double obs_stepsize,
                        double env_tau, const void* const* state_in, const void* actions, int action_layout,
                        void* obs_traj, void* const* state_traj, int traj_layout, void* const* last_state,
                        int semantics, const excenv_traj_gym_t* gym, void* workspace, int64_t workspace_bytes,
                        const excenv_launch_opts_t* opts, void* stream);
/* 1 when excenv_sim_ahead[_ws] with these arguments reads the row-major actions[B][K][A] (what the reference's
 * vmap_sim_ahead is handed, core_env.py:571-616) inside the lane-major trajectory kernel itself — 64-byte windows of every
 * environment's row through LDS, no transposition pass, no workspace — else 0 (then a workspace of
 * excenv_sim_ahead_workspace_bytes lets the library transpose them first). Applies to broadcast properties without gym
 * trajectories (control columns are fine when every control->reference[j] is given: they are filled by a second small launch
 * behind the lean kernel; n_control is ignored by this query), the batch sizes that run V = 16 / sizeof(dtype) environments per lane with B % (64 V) == 0,
 * K * A * sizeof(dtype) a multiple of 16 and 16-byte aligned actions; the state / output pointers must be 16-byte aligned as
 * for every vectorised launch. */
int excenv_sim_ahead_fuses_actions(int env, int solver, int dtype, int64_t B, int64_t K, const excenv_props_t* props,
                                   int32_t n_control, int with_gym, int action_layout, int traj_layout, const void* actions,
                                   const excenv_launch_opts_t* opts);
/* out[n][m] = in[m][n] for a row-major M x N matrix of the given dtype (the conversion kernel used above). */
int excenv_transpose(int dtype, int64_t M, int64_t N, const void* in, void* out, void* stream);

/* ---- replaces CoreEnvironment.vmap_generate_rew_trunc_term_ahead (core_env.py:490-531, 618-647) for a trajectory that
 * is already in memory (the fused form is the `gym` argument of excenv_sim_ahead): one thread per (env, row).
 *   state_traj : S pointers to [B x rows] arrays with element strides (state_env_stride, state_row_stride) — any layout
 *                excenv_sim_ahead writes (lane-major: (1, B); env-major: (rows, 1))
 *   control    : reference[j] is the base of the reference values of field control_idx[j]; ref_strides[2j], [2j+1] are its
 *                element strides (env, row) — (1, 0) for a reference that is constant along the trajectory; NULL = (1, 0)
 *   reward     : rows-1 values per env (rows 1..), terminated: rows-1 bytes, truncated: rows x excenv_truncated_width bytes,
 *                laid out as `out_layout` (EXCENV_LAYOUT_ENV_MAJOR [B][row][flag] or EXCENV_LAYOUT_LANE_MAJOR: reward / terminated
 *                [row][B], truncated [row][B][flag]) */
int excenv_rew_trunc_term(int env, int dtype, int64_t B, int64_t rows, const excenv_props_t* props,
                          const excenv_control_t* control, const int64_t* ref_strides, const void* const* state_traj,
                          int64_t state_env_stride, int64_t state_row_stride, void* reward, uint8_t* terminated,
                          uint8_t* truncated, int out_layout, void* stream);

/* ---- replaces CoreEnvironment.vmap_generate_state_from_observation (core_env.py:689-705; per env e.g.
 * pendulum_env.py:331-364, pmsm_env.py:921-970): obs [B][O + n_control] row-major -> denormalised physical state leaves
 * state_out[S][B] and, for each controlled field control_idx[j], its denormalised reference leaf reference_out[j][B]
 * (the other reference leaves are NaN by definition and are not written here). */
int excenv_state_from_observation(int env, int dtype, int64_t B, const excenv_props_t* props, int32_t n_control,
                                  const int32_t* control_idx, const void* obs, void* const* state_out,
                                  void* const* reference_out, void* stream);

/* ---- replaces GymWrapper.update_ref / generate_new_ref (gym_wrapper.py:170-192), one thread per environment: where
 * hold[i] == 0 draw a random initial state from the environment's key (init_state(rng), e.g. pendulum_env.py:270-276, PMSM
 * pmsm_env.py:402-456 incl. jax.random.ball), copy its controlled fields into reference[j][i], split the key for the new hold
 * time (jax.random.randint(sub, (1,), hold_steps_min, hold_steps_max) in the default int type: the int32 form for EXCENV_F32, the
 * int64 form JAX draws under jax_enable_x64 for EXCENV_F64 — float64 arrays exist in the reference only with x64 on) and keep the
 * other half as the new key; then hold[i] -= 1. All arrays are updated in place. keys: [B][2] uint32 key words stored in int64
 * (jax.random key data). The samplers restate JAX's published algorithms (threefry2x32 split / bits / uniform / randint / normal /
 * gamma / ball); the CPU oracle's restatement of the same functions is pinned on the Random123 known-answer vectors and on the
 * jax.random.split / normal values printed in JAX's documentation (tests/test_oracle_rng.py), and these kernels are compared
 * with that oracle (tests/test_gpu_rng_oracle.py). Unpinned: the x64 form of randint (no published value; checked against a
 * big-integer restatement). */
int excenv_update_ref(int env, int dtype, int64_t B, const excenv_props_t* props, int32_t n_control,
                      const int32_t* control_idx, void* const* reference, int64_t* keys, int64_t* hold,
                      int32_t hold_steps_min, int32_t hold_steps_max, void* stream);

/* ---- replaces CoreEnvironment.generate_observation vmapped over the batch (e.g. pendulum_env.py:311-329, PMSM
 * pmsm_env.py:898-919; used by vmap_reset, core_env.py:665-687): obs [B][O + n_control] row-major = the normalised physical
 * state in the environment's observation order followed by the normalised reference of each controlled field
 * (control->reference[j], NaN allowed). The same device function the step / trajectory kernels fuse. */
int excenv_observe(int env, int dtype, int64_t B, const excenv_props_t* props, const excenv_control_t* control,
                   const void* const* state, void* obs, void* stream);

/* Out-of-place form of excenv_update_ref (the functional contract of GymWrapper.update_ref, gym_wrapper.py:170-175: the
 * incoming state is not modified): reads reference_in / keys_in / hold_in, writes every environment's values — redrawn or
 * carried over — to reference_out / keys_out / hold_out. Outputs must not alias the inputs. */
int excenv_update_ref_to(int env, int dtype, int64_t B, const excenv_props_t* props, int32_t n_control,
                         const int32_t* control_idx, const void* const* reference_in, const int64_t* keys_in,
                         const int64_t* hold_in, void* const* reference_out, int64_t* keys_out, int64_t* hold_out,
                         int32_t hold_steps_min, int32_t hold_steps_max, void* stream);

/* ---- replaces CoreEnvironment.vmap_init_state(rng) (core_env.py:649-662) with one key per environment: the random branch
 * of each environment's init_state (e.g. pendulum_env.py:270-276, PMSM pmsm_env.py:402-456) — state_out[S][B] physical state
 * leaves, key_leaf [B][2] the keys that become State.PRNGKey. Same samplers as excenv_update_ref. */
int excenv_random_state(int env, int dtype, int64_t B, const excenv_props_t* props, const int64_t* keys,
                        void* const* state_out, int64_t* key_leaf, void* stream);

/* ---- the path's one collective (no reference counterpart: the reference is single-device): all ranks integrate
 * their own contiguous slice of the batch with no communication, and a consumer that needs the global batch on every rank
 * reassembles it with ONE all-gather — typically the last observation row of a chunk (obs_traj + N * (O + n_control) * B elements
 * in the lane-major layout: count_per_rank = (O + n_control) * B_local). A thin wrapper over RCCL's ncclAllGather for binders
 * that do not go through torch.distributed (the Python mirror does): `nccl_comm` is the caller's ncclComm_t, `recv` holds
 * world_size * count_per_rank elements in rank order, enqueued on `stream`. librccl.so is loaded on first use; without it the
 * call returns EXCENV_EUNSUPPORTED. Shards must be equal-sized (pad the last rank's slice otherwise). */
int excenv_allgather(void* nccl_comm, int dtype, const void* send, void* recv, int64_t count_per_rank, void* stream);

/* ---- calibration, no reference counterpart: the memory access shape of excenv_sim_ahead (lane-major buffers) without any
 * arithmetic. `rows` times, every workgroup reads one 4 KiB piece of each of n_read streams and writes one 4 KiB piece of each
 * of n_write streams (16 bytes per lane, 256 lanes); stream s covers bytes [0, row_bytes) of its row and advances by its own
 * row stride: for a trajectory call the read streams are the A action components (base actions + c*B*w, row stride A*B*w), the
 * write streams the O observation components (base obs + c*B*w, row stride O*B*w) and the S state leaves (row stride B*w), with
 * row_bytes = B*w and rows = K*substeps. What is written is meaningless: use it on buffers whose contents are dead. It tells,
 * in the same process and over the very buffers of a trajectory call, how fast HBM takes that traffic where the driver placed
 * those buffers (bench.py: roofline.same_run_pattern_gbs; the Python mirror uses it to reject slow placements of large
 * trajectory buffers before the first trajectory is written). n_read <= 4, n_write <= 32; bases and strides
 * 16-byte aligned; nontemporal != 0 selects the streaming stores the trajectory kernels use. */
int excenv_stream_pattern(int32_t n_read, const void* const* read_base, const int64_t* read_row_stride_bytes,
                          int32_t n_write, void* const* write_base, const int64_t* write_row_stride_bytes,
                          int64_t row_bytes, int64_t rows, int32_t nontemporal, void* stream);

/* ---- device-math probes (tests only): out[i] = f(in[i]) for the in-kernel fp32 routines -- */
int excenv_probe_math(int which /*0 sin,1 cos,2 wrap_angle*/, int dtype, int64_t n,
                      const void* in, void* out, void* stream);

/* out_fast[i] = the kernels' division by a loop-invariant denominator (devmath.hpp InvDiv) of num[i] by den[i];
 * out_ref[i] = num[i] / den[i] as the compiler expands it. Tests assert equal bits. */
int excenv_probe_div(int dtype, int64_t n, const void* num, const void* den, void* out_fast, void* out_ref, void* stream);

#define EXCENV_OK 0
#define EXCENV_EINVAL (-1)  /* bad enum / size / combination */
#define EXCENV_ENULL (-2)   /* required pointer is NULL */
#define EXCENV_EHIP (-3)    /* HIP runtime error at launch */
#define EXCENV_EUNSUPPORTED (-4)

#ifdef __cplusplus
}
#endif
#endif /* EXCENV_H */
