// The two hot kernels: one fused step launch (vmap_step) and one persistent K-step launch (vmap_sim_ahead).
// One lane per environment (V = 1) or V adjacent environments per lane with 16-byte vector accesses
// (lane-major trajectories only). State, parameters and RK stages stay in registers for the whole trajectory.
#pragma once
#include "rk.hpp"

namespace excenv {

#ifndef EXCENV_BLOCK
#define EXCENV_BLOCK 256
#endif
#ifndef EXCENV_NT_STORES
#define EXCENV_NT_STORES 1
#endif
#ifndef EXCENV_PINGPONG
#define EXCENV_PINGPONG 1  // bit 0: Euler, bit 1: RK4 / Tsit5 — K loop unrolled by two with ping-pong action registers
#endif                      // (2-step prefetch distance, 2x loop code). Measured: +3.5 % Euler, -5 % Tsit5.
constexpr int BLOCK = EXCENV_BLOCK;
// Deliberately broken builds for the self-test of the static guards (tools/isa_guards.py, tools/isa_guards_selftest.sh): bit 0 counts
// one store too many in the hand-written wait of the action windows, bit 1 drops the LDS wait in front of the row barrier, bit 2
// drops the wait state between the write of M0 and the LDS-direct load. Never set in a product build (static_assert in excenv_api.hip).
#ifndef EXCENV_FAULT
#define EXCENV_FAULT 0
#endif
// Property leaves in kernel-argument order: P statics, S mins, S maxs, A mins, A maxs.
template <typename T, class M> struct KProps {
  static constexpr int N = M::P + 2 * M::S + 2 * M::A;
  T scalar[N];
  const T* ptr[N];  // per-env [B] array or nullptr
  // PMSM saturated model: LUT grids / tables (nullptr otherwise)
  const T* lut_gd;
  const T* lut_gq;
  const T* lut_tab;
  int32_t lut_nd, lut_nq;
  int32_t lut_lds;  // 1: the launch reserved dynamic LDS for grids + tables (they are staged there once per workgroup)
};

template <typename T, class M> struct StepArgs {
  KProps<T, M> kp;
  int64_t B;
  const T* state_in[M::S];
  T* state_out[M::S];
  const T* action;  // [B][A]
  T* obs;           // [B][O + n_control]
  int32_t n_control;
  int32_t control_idx[EXCENV_MAX_CONTROL];
  const T* reference[EXCENV_MAX_CONTROL];
  const T* obs_reference[EXCENV_MAX_CONTROL];  // what the observation columns show (== reference unless the caller says otherwise)
  T dt, env_tau, adv_coef;
  // optional gym outputs (all three or none): reward [B], terminated [B] (0/1 bytes), truncated [B][TW] (0/1 bytes)
  T* reward;
  uint8_t* terminated;
  uint8_t* truncated;
};

template <typename T, class M> struct SimArgs {
  KProps<T, M> kp;
  int64_t B, K;
  int32_t substeps;
  int32_t n_control;
  int32_t row_sync;  // one-environment-per-lane instantiations: 1 = a workgroup barrier before the stores of every row, 2 = rows leave through LDS as 16-byte stores (launch.hpp)
  const T* state_in[M::S];
  T* last_state[M::S];
  const T* actions;
  int64_t a_sb, a_sk, a_sc;  // element strides of (env, action step, component)
  int64_t a_wg;              // element offset between consecutive workgroups' first envs
  T* obs;
  int64_t o_sb, o_sk, o_sc, o_wg;
  T* straj[M::S];  // straj[0] == nullptr: no state trajectory
  int64_t s_sb, s_sk, s_wg;
  int32_t control_idx[EXCENV_MAX_CONTROL];
  const T* reference[EXCENV_MAX_CONTROL];
  T dt, env_tau, adv_coef;
  T lin_stop;  // tau * (K - 1) folded in double (PMSM predicted angles, Ctx::lin_stop)
  // optional gym trajectories (all three or none; GENERAL instantiation): reward / terminated hold rows 1..N at step
  // index n-1 with element strides (g_sb, g_sk); truncated holds rows 0..N with strides (t_sb, t_sk, t_sc)
  T* reward;
  uint8_t* terminated;
  uint8_t* truncated;
  int64_t g_sb, g_sk, t_sb, t_sk, t_sc;
};

template <bool BATCHED, typename T, class M, bool FASTDIV = true>
__device__ __forceinline__ void load_ctx(Ctx<T, M>& c, const KProps<T, M>& kp, int64_t i, T dt, T env_tau, T adv_coef) {
  c.fastdiv = FASTDIV;
  auto get = [&](int j) -> T {
    if constexpr (BATCHED) {
      const T* p = kp.ptr[j];
      return p ? p[i] : kp.scalar[j];
    } else {
      return kp.scalar[j];
    }
  };
#pragma unroll
  for (int j = 0; j < M::P; ++j) c.P[j] = get(j);
#pragma unroll
  for (int j = 0; j < M::S; ++j) {
    c.smin[j] = get(M::P + j);
    c.smax[j] = get(M::P + M::S + j);
  }
#pragma unroll
  for (int j = 0; j < M::A; ++j) {
    c.amin[j] = get(M::P + 2 * M::S + j);
    c.amax[j] = get(M::P + 2 * M::S + M::A + j);
  }
  c.dt = dt;
  c.env_tau = env_tau;
  c.adv_coef = adv_coef;
  c.lut_gd = kp.lut_gd;
  c.lut_gq = kp.lut_gq;
  c.lut_tab = kp.lut_tab;
  c.lut_nd = kp.lut_nd;
  c.lut_nq = kp.lut_nq;
  c.lut_lds = 0;
  if constexpr (M::HAS_LUT) {  // wave-uniform: first node and (n - 1) / (last - first) of both grids
    c.lut_g0[0] = c.lut_g0[1] = c.lut_sc[0] = c.lut_sc[1] = T(0);
    if (kp.lut_tab != nullptr) {
      const T d0 = kp.lut_gd[0], dl = kp.lut_gd[kp.lut_nd - 1], q0 = kp.lut_gq[0], ql = kp.lut_gq[kp.lut_nq - 1];
      c.lut_g0[0] = d0;
      c.lut_g0[1] = q0;
      c.lut_sc[0] = T(kp.lut_nd - 1) / (dl - d0);
      c.lut_sc[1] = T(kp.lut_nq - 1) / (ql - q0);
    }
  }
  prep_ctx(c);
}

// PMSM saturated model: copy the grids and the node-interleaved tables into LDS once per workgroup (47 KB in fp32 —
// larger than the 32 KB vector L1, so per-lane gathers would otherwise be served by L2). Every thread of the
// workgroup must call this before any early exit.
template <class M, typename T>
__device__ __forceinline__ void stage_lut(Ctx<T, M>& c, const KProps<T, M>& kp) {
  if constexpr (M::HAS_LUT) {
    extern __shared__ __align__(16) unsigned char excenv_smem[];
    if (kp.lut_lds) {
      T* sm = reinterpret_cast<T*>(excenv_smem);
      const int ntab = kp.lut_nd * kp.lut_nq * 8;
      for (int j = threadIdx.x; j < ntab; j += blockDim.x) sm[j] = kp.lut_tab[j];
      for (int j = threadIdx.x; j < kp.lut_nd; j += blockDim.x) sm[ntab + j] = kp.lut_gd[j];
      for (int j = threadIdx.x; j < kp.lut_nq; j += blockDim.x) sm[ntab + kp.lut_nd + j] = kp.lut_gq[j];
      // refined reciprocals of the cell widths (InvDiv: the interpolation weight keeps the bits of its division)
      T* rd = sm + ntab + kp.lut_nd + kp.lut_nq;
      T* rq = rd + kp.lut_nd;
      for (int j = threadIdx.x; j < kp.lut_nd; j += blockDim.x) {
        InvDiv<T> w;
        w.init((j + 1 < kp.lut_nd) ? kp.lut_gd[j + 1] - kp.lut_gd[j] : T(1), true);
        rd[j] = w.y;
      }
      for (int j = threadIdx.x; j < kp.lut_nq; j += blockDim.x) {
        InvDiv<T> w;
        w.init((j + 1 < kp.lut_nq) ? kp.lut_gq[j + 1] - kp.lut_gq[j] : T(1), true);
        rq[j] = w.y;
      }
      __syncthreads();
      c.lut_lds = 1;
    }
  }
}

// ---- vector access helpers ---------------------------------------------------------------
template <typename T, int V> struct VecOf;
template <> struct VecOf<float, 1> { using type = float; };
template <> struct VecOf<float, 2> { using type = float2; };
template <> struct VecOf<float, 4> { using type = float4; };
template <> struct VecOf<double, 1> { using type = double; };
template <> struct VecOf<double, 2> { using type = double2; };

template <typename T, int V> __device__ __forceinline__ void load_v(const T* p, T (&out)[V]) {
  static_assert(V * sizeof(T) <= 16, "at most one 16-byte access per lane");
  if constexpr (V == 1) {
    out[0] = *p;
  } else {
    using VT = typename VecOf<T, V>::type;
    const VT v = *reinterpret_cast<const VT*>(p);
    const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
    for (int j = 0; j < V; ++j) out[j] = e[j];
  }
}
template <typename T, int V> __device__ __forceinline__ void store_v(T* p, const T (&in)[V]) {
  if constexpr (V == 1) {
    *p = in[0];
  } else {
    using VT = typename VecOf<T, V>::type;
    VT v;
    T* e = reinterpret_cast<T*>(&v);
#pragma unroll
    for (int j = 0; j < V; ++j) e[j] = in[j];
    *reinterpret_cast<VT*>(p) = v;
  }
}

// Trajectory rows are written once and never read back by the kernel: optional streaming (nt) stores.
template <typename T, int V> __device__ __forceinline__ void store_stream(T* p, const T (&in)[V]) {
#if EXCENV_NT_STORES
  if constexpr (V == 1) {
    *p = in[0];  // V == 1 also serves the env-major layout, whose scattered words must merge in L2: no nt
  } else {
    typedef T NVT __attribute__((ext_vector_type(V)));  // the builtin wants a native clang vector
    NVT v;
#pragma unroll
    for (int j = 0; j < V; ++j) v[j] = in[j];
    __builtin_nontemporal_store(v, reinterpret_cast<NVT*>(p));
  }
#else
  store_v<T, V>(p, in);
#endif
}

// Row-major row of N values per lane (obs[B][O], action[B][A]): widest power-of-two chunks up to 16 bytes.
template <typename T, int N> __device__ __forceinline__ void store_row(T* dst, const T (&v)[N]) {
  constexpr int MAXV = 16 / sizeof(T);
  constexpr int W = (N % MAXV == 0) ? MAXV : ((N % 2 == 0 && MAXV >= 2) ? 2 : 1);
#pragma unroll
  for (int j = 0; j < N; j += W) {
    T tmp[W];
#pragma unroll
    for (int w = 0; w < W; ++w) tmp[w] = v[j + w];
    store_v<T, W>(dst + j, tmp);
  }
}
template <typename T, int N> __device__ __forceinline__ void load_row(const T* src, T (&v)[N]) {
  constexpr int MAXV = 16 / sizeof(T);
  constexpr int W = (N % MAXV == 0) ? MAXV : ((N % 2 == 0 && MAXV >= 2) ? 2 : 1);
#pragma unroll
  for (int j = 0; j < N; j += W) {
    T tmp[W];
    load_v<T, W>(src + j, tmp);
#pragma unroll
    for (int w = 0; w < W; ++w) v[j + w] = tmp[w];
  }
}

// V one-byte flags of V adjacent environments as ONE store (lane-major flag trajectories; p is V-byte aligned)
template <int V> __device__ __forceinline__ void store_flags(uint8_t* p, const uint8_t (&f)[V]) {
  if constexpr (V == 4) {
    *reinterpret_cast<uint32_t*>(p) = (uint32_t)f[0] | ((uint32_t)f[1] << 8) | ((uint32_t)f[2] << 16) | ((uint32_t)f[3] << 24);
  } else if constexpr (V == 2) {
    *reinterpret_cast<uint16_t*>(p) = (uint16_t)((uint16_t)f[0] | ((uint16_t)f[1] << 8));
  } else {
    *p = f[0];
  }
}

// NB adjacent flag bytes per lane — the truncated flags of a lane's V environments, TW each, in the lane-major flag layout
// [row][B][TW] — already packed into dwords: the fewest stores (16 / 12 / 8 / 4 / 2 bytes). p is aligned to NB's largest power-of-two
// factor up to 4 (V * TW bytes per lane: 4 with four environments per lane, 2 with two and an odd TW), not more: the types say so,
// the global stores of gfx950 take it.
template <int AL> struct FlagWords;
template <> struct FlagWords<4> {
  typedef uint32_t U4 __attribute__((ext_vector_type(4), aligned(4)));
  typedef uint32_t U3 __attribute__((ext_vector_type(3), aligned(4)));
  typedef uint32_t U2 __attribute__((ext_vector_type(2), aligned(4)));
  typedef uint32_t U1 __attribute__((aligned(4)));
};
template <> struct FlagWords<2> {
  typedef uint32_t U4 __attribute__((ext_vector_type(4), aligned(2)));
  typedef uint32_t U3 __attribute__((ext_vector_type(3), aligned(2)));
  typedef uint32_t U2 __attribute__((ext_vector_type(2), aligned(2)));
  typedef uint32_t U1 __attribute__((aligned(2)));
};
template <int NB> __device__ __forceinline__ void store_flag_bytes(uint8_t* p, const uint32_t (&w)[(NB + 3) / 4]) {
  static_assert(NB % 2 == 0, "an even number of environments per lane");
  typedef FlagWords<(NB % 4 == 0) ? 4 : 2> FW;
  typedef typename FW::U4 U4;
  typedef typename FW::U3 U3;
  typedef typename FW::U2 U2;
  typedef typename FW::U1 U1;
  constexpr int NW = NB / 4;
  constexpr int Q = NW / 4 * 4;
#pragma unroll
  for (int j = 0; j < Q; j += 4) {
    U4 v;
    v[0] = w[j]; v[1] = w[j + 1]; v[2] = w[j + 2]; v[3] = w[j + 3];
    *reinterpret_cast<U4*>(p + 4 * j) = v;
  }
  if constexpr (NW - Q == 3) {
    U3 v;
    v[0] = w[Q]; v[1] = w[Q + 1]; v[2] = w[Q + 2];
    *reinterpret_cast<U3*>(p + 4 * Q) = v;
  } else if constexpr (NW - Q == 2) {
    U2 v;
    v[0] = w[Q]; v[1] = w[Q + 1];
    *reinterpret_cast<U2*>(p + 4 * Q) = v;
  } else if constexpr (NW - Q == 1) {
    *reinterpret_cast<U1*>(p + 4 * Q) = w[Q];
  }
  if constexpr (NB % 4 == 2) *reinterpret_cast<uint16_t*>(p + 4 * NW) = (uint16_t)w[NW];
}
template <int K> struct IntC { static constexpr int value = K; };
// f(IntC<n>) for the wave-uniform n in [K, KMAX] (a chain of scalar branches; n outside the range: nothing)
template <int K, int KMAX, class F> __device__ __forceinline__ void dispatch_count(int n, F&& f) {
  if (n == K) f(IntC<K>{});
  else if constexpr (K < KMAX) dispatch_count<K + 1, KMAX>(n, f);
}

// ---- gym outputs of one saved state (generate_reward / generate_terminated / generate_truncated) ----------------
// `ob` is the observation row of `st` (without control columns); `ref[j]` the physical reference of control column j.
// Stores through the three element pointers (reward, terminated: one element; truncated: TW flags, stride t_sc).
template <class M, typename T>
__device__ __forceinline__ void gym_outputs(const T (&st)[M::S], const T (&ob)[M::O], const Ctx<T, M>& c, int n_control,
                                            const int* idx, const T (&ref)[EXCENV_MAX_CONTROL], T* reward, uint8_t* terminated,
                                            uint8_t* truncated, int64_t t_sc) {
  const T rew = (reward != nullptr) ? env_reward<M, T>(st, c, n_control, idx, ref) : T(0);
  if (reward != nullptr) *reward = rew;
  if constexpr (M::IS_PMSM) {  // pmsm_env.py:972-983: |i_dq_norm| > 1, terminated == truncated
    const T nd = normalize(st[3], c.smin[3], c.smax[3]), nq = normalize(st[4], c.smin[4], c.smax[4]);
    const uint8_t t = sqrt_exceeds_one(nd * nd + nq * nq);  // == sqrt(.) > 1, bit for bit (devmath.hpp)
    if (truncated != nullptr) truncated[0] = t;
    if (terminated != nullptr) *terminated = t;
  } else if constexpr (M::ID == EXCENV_FLUID_TANK) {  // fluid_tank_env.py:325-333: constants
    if (truncated != nullptr) truncated[0] = 0;
    if (terminated != nullptr) *terminated = 0;
  } else {  // truncated = |obs| > 1 over every observation column, terminated = (reward == 0)
    if (truncated != nullptr) {
#pragma unroll
      for (int q = 0; q < M::O; ++q) truncated[q * t_sc] = xabs(ob[q]) > T(1);
#pragma unroll
      for (int j = 0; j < EXCENV_MAX_CONTROL; ++j) {
        if (j < n_control) {
          T x, lo, hi;
          pick_field<M, T>(st, c, idx[j], x, lo, hi);
          truncated[(M::O + j) * t_sc] = xabs(normalize(ref[j], lo, hi)) > T(1);
        }
      }
    }
    if (terminated != nullptr) *terminated = rew == T(0);
  }
}

// ---- vmap_step: one fused launch (reference core_env.py:533-569) ---------------------------
// V adjacent envs per lane (V > 1 only in the non-GENERAL instantiations): [B] state arrays move as 16-byte vectors, the
// row-major action / obs rows of the V envs are one contiguous run of V*A / V*O words. GENERAL (V == 1): per-env property
// arrays, reference-tracking observation columns and the optional fused gym outputs.
template <class M, typename T, int SOLVER, bool GENERAL, int V>
__global__ void __launch_bounds__(BLOCK) step_kernel(const StepArgs<T, M> ka) {
  constexpr int S = M::S, A = M::A, O = M::O;
  static_assert(!(GENERAL && V > 1), "vectorised lanes share one uniform property set");
  // Observation rows wider than 16 bytes (PMSM: 32 B in fp32, 64 B in fp64; the 4-column environments in fp64): one row per
  // lane would leave as 16-byte stores 32 / 64 bytes apart — every store instruction of the wave half / quarter dense. Those
  // rows pass through an LDS staging area of the wave instead and leave as 64 CONSECUTIVE 16-byte pieces per instruction
  // (1 KiB), like the trajectory kernels' stores. Lanes past the batch end stay alive on this path: they carry pieces of
  // other lanes' rows.
  constexpr int VW = 16 / (int)sizeof(T);
  constexpr bool DENSE = !GENERAL && V == 1 && (O % VW) == 0 && (O / VW) > 1;
  const int64_t blk0 = (int64_t)blockIdx.x * (BLOCK * V);
  const unsigned lane_env = threadIdx.x * V;
  const int64_t i = blk0 + lane_env;
  Ctx<T, M> c;
  load_ctx<GENERAL, T, M, false>(c, ka.kp, (i < ka.B) ? i : 0, ka.dt, ka.env_tau, ka.adv_coef);  // one step: plain division
  stage_lut<M, T>(c, ka.kp);
  const bool live = i < ka.B;
  if (!DENSE && !live) return;
  const int64_t il = live ? i : ka.B - 1;  // DENSE: a lane past the end re-reads the last environment and stores nothing of its own
  T st[V][S], a[V * A], ob[V * O];
#pragma unroll
  for (int j = 0; j < S; ++j) {
    T tmp[V];
    load_v<T, V>(ka.state_in[j] + il, tmp);
#pragma unroll
    for (int v = 0; v < V; ++v) st[v][j] = tmp[v];
  }
  load_row<T, V * A>(ka.action + il * A, a);
#pragma unroll
  for (int v = 0; v < V; ++v) {
    T av[A], ov[O];
#pragma unroll
    for (int q = 0; q < A; ++q) av[q] = a[v * A + q];
    env_step<M, SOLVER>(st[v], av, c);
    M::observe(st[v], c, ov);
#pragma unroll
    for (int q = 0; q < O; ++q) ob[v * O + q] = ov[q];
  }
  if (live) {
#pragma unroll
    for (int j = 0; j < S; ++j) {
      T tmp[V];
#pragma unroll
      for (int v = 0; v < V; ++v) tmp[v] = st[v][j];
      store_v<T, V>(ka.state_out[j] + i, tmp);
    }
  }
  if constexpr (DENSE) {
    constexpr int PR = O / VW;  // 16-byte pieces per row
    __shared__ __align__(16) T stage[BLOCK * O];
    const unsigned wave = threadIdx.x / 64u, lane = threadIdx.x % 64u;
    T* const ws = stage + wave * 64u * O;  // this wave's 64 rows; only this wave touches them: LDS operations of one wave execute
#pragma unroll                             // in issue order, so a compiler fence is all the synchronisation needed
    for (int q = 0; q < O; q += VW) {
      T v[VW];
#pragma unroll
      for (int h = 0; h < VW; ++h) v[h] = ob[q + h];
      store_v<T, VW>(ws + lane * O + q, v);
    }
    asm volatile("" ::: "memory");
    const int64_t row0 = blk0 + wave * 64u;  // first environment of the wave
    T* const wobs = ka.obs + row0 * O;
#pragma unroll
    for (int k = 0; k < PR; ++k) {
      const unsigned p = lane + 64u * k;  // piece index == memory order
      T v[VW];
      load_v<T, VW>(ws + p * VW, v);
      if (row0 + (int64_t)(p / PR) < ka.B) store_v<T, VW>(wobs + p * VW, v);
    }
  } else if constexpr (!GENERAL) {
    store_row<T, V * O>(ka.obs + blk0 * O + lane_env * O, ob);
  } else {
    T rref[EXCENV_MAX_CONTROL];
#pragma unroll
    for (int j = 0; j < EXCENV_MAX_CONTROL; ++j) rref[j] = (j < ka.n_control) ? ka.reference[j][i] : T(0);
    if (ka.n_control == 0) {
      store_row<T, O>(ka.obs + i * O, ob);
    } else {
      T* row = ka.obs + i * (O + ka.n_control);
#pragma unroll
      for (int j = 0; j < O; ++j) row[j] = ob[j];
#pragma unroll
      for (int j = 0; j < EXCENV_MAX_CONTROL; ++j) {
        if (j < ka.n_control) {
          T x, lo, hi;
          pick_field<M, T>(st[0], c, ka.control_idx[j], x, lo, hi);
          row[O + j] = normalize(ka.obs_reference[j][i], lo, hi);
        }
      }
    }
    if (ka.reward != nullptr) {  // GymWrapper.gym_step (gym_wrapper.py:117-126), fused: no second pass over the state
      const int64_t tw = (M::IS_PMSM || M::ID == EXCENV_FLUID_TANK) ? 1 : O + ka.n_control;
      gym_outputs<M, T>(st[0], ob, c, ka.n_control, ka.control_idx, rref, ka.reward + i, ka.terminated + i,
                        ka.truncated + i * tw, 1);
    }
  }
}

// ---- vmap_sim_ahead: one persistent launch for all N = K*substeps solver steps ------------------
// (reference core_env.py:571-616 + each env's _ode_solver_simulate_ahead; PMSM.sim_ahead pmsm_env.py:746-801)
//
// GENERAL instantiation (V == 1): per-env property arrays, reference-tracking observation columns and the optional
// reward / terminated / truncated trajectories (core_env.py:490-531). The other instantiations assume broadcast
// properties and no control columns (the host routes accordingly).
//
// Pipeline: the action rows ping-pong between two register sets (component-major, so a 16-byte load lands in place);
// the row of solver step n+1 is requested before row n is saved and step n is computed, and is first read one whole step
// later (Euler) or in the last RK stage of step n (c_i == 1 stages see action k+1) — its s_waitcnt therefore sits after a
// compute phase and never has to drain the trajectory stores issued in between (vmcnt counts loads and stores in order).
//
// STATES: the state trajectories are written (1), not written (0), or decided by ka.straj[0] at run time (GENERAL only: -1 with
// the gym outputs' code, -2 without it).
// It is a compile-time property on the vectorised instantiations because s_waitcnt vmcnt counts loads and stores in issue
// order: with the state stores behind a run-time branch the compiler must assume the shorter path, and the wait for the
// prefetched action row then also waits for the first seven stores of the row just written.
// LUT_LDS (look-up models): the launch staged the tables in LDS (true) or the kernel gathers from global memory (false) —
// a compile-time property so that each instantiation carries ONE copy of the look-up code (the saturated-PMSM loops are
// the largest in the library and run out of the 64 KB instruction cache otherwise).
//
// AEM (row-major actions — the reference's plain actions[B][K][A] tensor — read by the lane-major kernel itself, no transposition
// pass; lean instantiations with V * sizeof(T) == 16 only). Time is the contiguous axis of that array, so what a lane needs per
// step (A values of each of ITS V environments) lies K * A elements apart from its neighbour's. Fetching it 16 bytes at a time
// costs one 64-byte fabric read per piece (measured: TCC_EA_RDREQ == number of pieces, 7.4 ms for the headline launch); the L2
// does not keep a line until the walk returns to it. So an environment's row is fetched in WINDOWS of AEM_NP pieces (64 bytes)
// by AEM_NP ADJACENT LANES of one LDS-direct load (global_load_lds_dwordx4: lane t of instruction i loads piece t % NP of
// environment-slot i * 64 / NP + t / NP and the hardware puts it at M0 + 16 t — the window of an environment is contiguous in
// LDS as in memory, one 64-byte request per environment and window). Every wave owns V * NP blocks of 1 KiB (+ 16 bytes each:
// bank skew for the readers) and fetches only the rows of its own lanes' environments: no barrier, every wave on its own.
// Single-buffered: a slot's 16-byte piece (SP rows) is read into registers once per piece, and the window is re-filled right after
// the read of its LAST piece (round 5; rounds 4 - 5 read a row per step and re-filled after the last row), so the fill has SP rows to
// land; the counted s_waitcnt in front of the first piece read behind a fill leaves the last saved row's trajectory stores in flight.
template <typename T, int V> constexpr bool aem_shape_ok() { return V * (int)sizeof(T) == 16; }
#ifndef EXCENV_AEM_NP
#define EXCENV_AEM_NP 4  // 16-byte pieces per window (64 bytes; V * NP KiB of LDS per wave)
#endif
constexpr int AEM_BLOCK_BYTES = 1024 + 16;  // one LDS-direct load instruction's 1 KiB + the bank skew
// Pieces per window. 64-byte windows everywhere (same-session sweep over nine workloads, tools/r4_aem_sweep.sh: 32-byte windows
// cost 3 ... 30 % more: two fabric requests per 64 bytes) except acrobot, whose registers already cap it at two workgroups per
// CU and which gains 6 % from the smaller LDS footprint.
template <class M> constexpr int aem_np() { return M::ID == EXCENV_ACROBOT ? 2 : EXCENV_AEM_NP; }
template <class M, typename T, int V> constexpr size_t aem_lds_bytes() { return (size_t)(BLOCK / 64) * V * aem_np<M>() * AEM_BLOCK_BYTES; }

// LGYM (round 4; lean, V * sizeof(T) == 16, not the look-up model): the reward / terminated / truncated trajectories of
// core_env.py:490-531 written by the wide kernel itself. PMSM (pmsm_env.py:972-1037): three references per environment, ~40
// instructions, one flag byte (|i_dq| > 1) — one 16-byte reward store and two packed 4-byte flag stores per row of the lane's four
// environments. The other models (e.g. pendulum_env.py:297-309, 381-390): control_state is a subset of the S state fields, what
// depends on the references alone is computed once per trajectory, per row one sin / cos per controlled ANGLE and environment, and
// O + n_control packed flag stores. The general instantiation (one environment per lane, byte stores, the reward loop unrolled over
// eight possible controls) took 7.5 ms (PMSM) / 4.5 ms (pendulum) for launches the lean kernels do in 4.9 / 1.6.
//
// NT (round 4): threads per workgroup. The Euler kernels of the two- and one-leaf models (pendulum, mass-spring-damper, tank) do so
// little arithmetic per row that their waves mostly wait for memory; sixteen waves that store a row TOGETHER (one barrier per row)
// write 16 KiB runs per stream where four unsynchronised waves wrote 1 KiB runs whenever each got there, and the memory side
// rewards that: pendulum Euler fp32 (C2) 3.98 -> 3.56 ms, MSD Euler 1.53 -> 1.43, tank 0.87 -> 0.81 in a same-buffers A/B
// (tools/ab_same_buffers.py, profiles/r04_pattern_sweep.md). With more arithmetic per row (RK4 / Tsit5, cart-pole, acrobot, PMSM)
// lockstep takes away the overlap of one wave's arithmetic with another's stores and the same change LOSES 2 ... 9 %: NT == BLOCK
// there, no barrier. sim_threads<M, T>() (launch.hpp) is the rule.
// Register caps (the "amdgpu-waves-per-eu" lower bound: 512 / n registers per lane). The compiler does not weigh a wave per SIMD
// against a few registers: when the math primitives were restructured in round 5 these instantiations went from 205 ... 254 to
// 259 ... 294 registers — one wave per SIMD instead of two (acrobot's gym trajectories: 4.9 -> 6.0 ms). Everything else keeps the
// compiler's choice (the fp64 RK kernels of the four-leaf models and the look-up model need more than 256).
// The Euler kernels of cart-pole and acrobot with gym outputs (192 registers with the packed flags): neither HBM nor the vector
// units are saturated at two waves per SIMD (0.60 … 0.62 of the roof); capped at 168 registers a third wave is resident, at the
// price of 48 … 128 spilled bytes: cart-pole 4.72 -> 4.56 ms, acrobot 4.81 -> 4.68 (B = 2^22, K = 100, two runs each).
template <class M, typename T, bool GENERAL, bool AEM, bool LGYM, int STATES, int SOLVER = -1> constexpr int sim_min_waves() {
  if (LGYM && (M::ID == EXCENV_ACROBOT || M::ID == EXCENV_CART_POLE) && sizeof(T) == 4 && SOLVER == EXCENV_EULER) return 3;
  if (LGYM && M::ID == EXCENV_ACROBOT && sizeof(T) == 4) return 2;
  if (GENERAL && M::ID == EXCENV_PENDULUM && sizeof(T) == 8) return 2;
  if (AEM && M::IS_PMSM && !M::HAS_LUT && sizeof(T) == 8 && STATES == 0) return 2;
  return 1;
}
#define EXCENV_SIM_KERNEL_ATTR __attribute__((amdgpu_waves_per_eu(sim_min_waves<M, T, GENERAL, AEM, LGYM, STATES, SOLVER>())))
template <class M, typename T, int SOLVER, bool AHEAD, bool GENERAL, int V, int STATES, bool LUT_LDS = false, bool AEM = false, bool LGYM = false,
          int NT = BLOCK>
__global__ void __launch_bounds__(NT) EXCENV_SIM_KERNEL_ATTR sim_ahead_kernel(const SimArgs<T, M> ka) {
  constexpr int S = M::S, A = M::A, O = M::O;
  extern __shared__ __align__(16) unsigned char excenv_smem[];
  static_assert(NT == BLOCK || (!GENERAL && !AEM && !M::HAS_LUT), "wide workgroups: lean instantiations only (plain or with gym outputs)");
  constexpr bool ROW_BARRIER = NT > BLOCK;  // wide workgroups: the sixteen waves store every row together
  // GENERAL stays at one environment per lane. Round 4 tried two, each with its own property set (a second Ctx in registers:
  // every leaf may differ per environment, so none can stay in SGPRs — 195 registers, two waves per SIMD): 5.91 ... 6.27 ms for one,
  // 6.02 ... 6.06 for two (tools/general_path_cost.py, two sessions) — no gain, removed. What did help is compiling the gym
  // outputs' code out where none are asked for (STATES == -2): 0.569 -> 0.60 of the roof, the lean one-environment form's level.
  static_assert(!(GENERAL && V > 1), "per-environment property sets: one environment per lane");
  static_assert(GENERAL == (STATES < 0), "STATES -1 / -2 (general, with / without the gym outputs' code) and 0 / 1 (lean)");
  static_assert(!LGYM || (!GENERAL && !M::HAS_LUT && aem_shape_ok<T, V>()), "lean gym outputs: widest lean form, no look-up model");
  constexpr int NC = GENERAL ? V : 1;  // property sets per lane
  constexpr bool GYM = GENERAL && STATES == -1;
  static_assert(!AEM || (!GENERAL && !M::HAS_LUT && aem_shape_ok<T, V>() && (16 / (int)sizeof(T)) % A == 0),
                "row-major actions are fused into the widest lean instantiation only");
  const int64_t blk0 = (int64_t)blockIdx.x * (NT * V);  // first env of this workgroup
  const unsigned lane_env = threadIdx.x * V;
  const int64_t i0 = blk0 + lane_env;
  Ctx<T, M> cs[NC];
#pragma unroll
  for (int v = 0; v < NC; ++v) {
    load_ctx<GENERAL>(cs[v], ka.kp, (i0 + v < ka.B) ? i0 + v : 0, ka.dt, ka.env_tau, ka.adv_coef);
    cs[v].lin_stop = ka.lin_stop;
    cs[v].lin_div = T(ka.K - 1);
    cs[v].lin_last = ka.K - 1;
  }
  Ctx<T, M>& c = cs[0];  // what is the same for every environment of the lane (dead time, look-up tables: V == 1 there)
#define EXCENV_CX(v) cs[GENERAL ? (v) : 0]
  stage_lut<M, T>(c, ka.kp);
  if constexpr (M::HAS_LUT) c.lut_lds = LUT_LDS ? 1 : 0;  // == ka.kp.lut_lds (launch_sim_v picks the instantiation by it)
  // host guarantees B % V == 0; AEM: B % (64 V) == 0 — a wave is whole or absent (its lanes also fetch for each other)
  if (i0 >= ka.B) return;

  T st[V][S];
#pragma unroll
  for (int j = 0; j < S; ++j) {
    T tmp[V];
    load_v<T, V>(ka.state_in[j] + blk0 + lane_env, tmp);
#pragma unroll
    for (int v = 0; v < V; ++v) st[v][j] = tmp[v];
  }
  AheadAux<T> aux[V];
  if constexpr (AHEAD && M::IS_PMSM) {
#pragma unroll
    for (int v = 0; v < V; ++v) {
      aux[v].eps0 = st[v][2];
      aux[v].prev_clip[0] = st[v][0];
      aux[v].prev_clip[1] = st[v][1];
    }
  }
  // LGYM, the other models (pendulum_env.py:297-309, 381-390 and the like): control_state is a subset of the S state fields, so at
  // most NCM = S controls. Everything that depends on a reference only — sin / cos of a controlled angle's reference or the
  // normalised reference of any other field, and the (constant) truncated flag of its observation column — is computed once per
  // trajectory with the functions the general instantiation calls per row: same values, same bits.
  constexpr int NCM = (LGYM && !M::IS_PMSM) ? S : 1;
  T gp_a[V][NCM], gp_b[V][NCM];
  uint8_t gp_f[V][NCM];
  if constexpr (LGYM && !M::IS_PMSM) {
#pragma unroll
    for (int j = 0; j < NCM; ++j) {
#pragma unroll
      for (int v = 0; v < V; ++v) {
        gp_a[v][j] = gp_b[v][j] = T(0);
        gp_f[v][j] = 0;
      }
      if (j < ka.n_control) {
        const int f = ka.control_idx[j];
        T lo = c.smin[0], hi = c.smax[0];
        bool ang = false;
#pragma unroll
        for (int q = 0; q < S; ++q) {
          lo = (f == q) ? c.smin[q] : lo;
          hi = (f == q) ? c.smax[q] : hi;
          ang = ang || (is_angle_field<M>(q) && f == q);
        }
        T tmp[V];
        load_v<T, V>(ka.reference[j] + blk0 + lane_env, tmp);
#pragma unroll
        for (int v = 0; v < V; ++v) {
          const T nr = normalize(tmp[v], lo, hi);
          gp_f[v][j] = xabs(nr) > T(1);
          if (ang) sincos_t(tmp[v], gp_a[v][j], gp_b[v][j]);
          else gp_a[v][j] = nr;
        }
      }
    }
  }
  // ... and the control columns' share of a row's packed truncated flags (lane-major flag layout [row][B][TW]: the V * TW bytes
  // of a lane's environments are adjacent; byte v * TW + O + j is control column j of environment v), once per trajectory
  constexpr int GP_NW = (LGYM && !M::IS_PMSM) ? (V * (O + NCM) + 3) / 4 : 1;
  uint32_t gp_w[GP_NW];
  if constexpr (LGYM && !M::IS_PMSM && M::ID != EXCENV_FLUID_TANK) {
#pragma unroll
    for (int i = 0; i < GP_NW; ++i) gp_w[i] = 0u;
    dispatch_count<0, NCM>(ka.n_control, [&](auto tag) {
      constexpr int NC = decltype(tag)::value, TWc = O + NC;
#pragma unroll
      for (int v = 0; v < V; ++v) {
#pragma unroll
        for (int j = 0; j < NC; ++j) {
          const int b = v * TWc + O + j;
          gp_w[b >> 2] |= (uint32_t)gp_f[v][j] << ((b & 3) * 8);
        }
      }
    });
  }
  // LGYM, PMSM: the references of the controlled fields among i_d (3), i_q (4), torque (5), per environment of the lane
  T g_id[V], g_iq[V], g_tq[V];
  bool has_id = false, has_iq = false, has_tq = false;
  if constexpr (LGYM && M::IS_PMSM) {
#pragma unroll
    for (int v = 0; v < V; ++v) g_id[v] = g_iq[v] = g_tq[v] = T(0);
#pragma unroll
    for (int j = 0; j < EXCENV_MAX_CONTROL; ++j) {
      if (j < ka.n_control) {
        const int f = ka.control_idx[j];
        T tmp[V];
        load_v<T, V>(ka.reference[j] + blk0 + lane_env, tmp);
        if (f == 3) has_id = true;
        if (f == 4) has_iq = true;
        if (f == 5) has_tq = true;
#pragma unroll
        for (int v = 0; v < V; ++v) {
          g_id[v] = (f == 3) ? tmp[v] : g_id[v];
          g_iq[v] = (f == 4) ? tmp[v] : g_iq[v];
          g_tq[v] = (f == 5) ? tmp[v] : g_tq[v];
        }
      }
    }
  }
  const bool deadtime_on = (M::IS_PMSM) ? (c.P[M::P - 1] > T(0)) : false;
  // look-up models: the table values at each environment's current operating point, found once per solver step and used
  // for the torque of the saved row and for the first stage of the step that starts there
  T memo[V][6];
  if constexpr (M::HAS_LUT && !AHEAD) {
#pragma unroll
    for (int v = 0; v < V; ++v) M::lookup(st[v][3], st[v][4], c, memo[v]);
  }

  // reference-tracking columns: constant along the trajectory, loaded and normalised once (static register indices)
  T rref[NC][EXCENV_MAX_CONTROL], cref[NC][EXCENV_MAX_CONTROL];
  if constexpr (GENERAL) {
#pragma unroll
    for (int v = 0; v < NC; ++v) {
#pragma unroll
      for (int j = 0; j < EXCENV_MAX_CONTROL; ++j) {
        rref[v][j] = T(0);
        cref[v][j] = T(0);
        if (j < ka.n_control) {
          const int f = ka.control_idx[j];
          T lo = cs[v].smin[0], hi = cs[v].smax[0];
#pragma unroll
          for (int q = 1; q < S; ++q) {
            lo = (f == q) ? cs[v].smin[q] : lo;
            hi = (f == q) ? cs[v].smax[q] : hi;
          }
          rref[v][j] = ka.reference[j][i0 + v];
          cref[v][j] = normalize(rref[v][j], lo, hi);
        }
      }
    }
  }

  const int64_t N = ka.K * ka.substeps;
  // V > 1 implies env stride 1; for V == 1 the host has checked that 256 * stride * sizeof(T) < 2^31
  const T* a_blk = ka.actions + (int64_t)blockIdx.x * ka.a_wg;
  T* o_blk = ka.obs + (int64_t)blockIdx.x * ka.o_wg;
  const int64_t s_blk = (int64_t)blockIdx.x * ka.s_wg;
  const unsigned a_lane = (V == 1) ? threadIdx.x * (unsigned)ka.a_sb : lane_env;
  const unsigned o_lane = (V == 1) ? threadIdx.x * (unsigned)ka.o_sb : lane_env;
  const unsigned s_lane = (V == 1) ? threadIdx.x * (unsigned)ka.s_sb : lane_env;
  const bool with_states = (STATES < 0) ? (ka.straj[0] != nullptr) : (STATES != 0);

  // ---- save row n: observation, (control columns), state leaves, (gym outputs); returns the saved state in sv ----
  auto save_row = [&](int64_t n, T (&sv)[V][S]) {
#pragma unroll
    for (int v = 0; v < V; ++v) {
#pragma unroll
      for (int j = 0; j < S; ++j) sv[v][j] = st[v][j];
      if constexpr (AHEAD) {
        if constexpr (M::HAS_LUT) M::post_q(sv[v], c, memo[v]);
        else M::post(sv[v], EXCENV_CX(v));
        if constexpr (M::IS_PMSM) {  // pmsm_env.py:785-791
          if (deadtime_on) {
            sv[v][0] = aux[v].prev_clip[0];  // row 0: still the initial buffer (prev_clip starts as it)
            sv[v][1] = aux[v].prev_clip[1];
          } else {
            sv[v][0] = T(0);
            sv[v][1] = T(0);
          }
        }
      }
    }
    T ob[V][O];
#pragma unroll
    for (int v = 0; v < V; ++v) M::observe(sv[v], EXCENV_CX(v), ob[v]);
    T* orow = o_blk + n * ka.o_sk;
    // the waves of a workgroup store a row together: always in the wide-workgroup form; with one environment per lane (4-byte
    // stores, 256-byte runs per wave and stream) where the host asks for it — PMSM with per-environment properties at B = 2^22:
    // 6.29 -> 5.50 ms (0.567 -> 0.648 of the roof), the lean V = 1 form 6.34 -> 5.61
    bool direct = true;
    if constexpr (ROW_BARRIER) {
      __builtin_amdgcn_s_barrier();
    } else if constexpr (V == 1 && !M::HAS_LUT && !AEM) {
      // row_sync == 2 (lane-major arrays, whole workgroups, 16-byte aligned): the row goes through LDS — every lane leaves its
      // values as [stream][lane] words, one barrier, then the waves share the streams and store 16 bytes per lane: 1 KiB runs per
      // instruction instead of 256-byte ones, a quarter of the store instructions. Two buffers alternate, so the barrier of row
      // n + 1 is also the one that frees row n's buffer.
      if (ka.row_sync == 2) {
        constexpr int VE = 16 / (int)sizeof(T), CH = NT / (64 * VE), NW = NT / 64;
        const int OWr = O + (GENERAL ? ka.n_control : 0);
        const int NS = OWr + (with_states ? S : 0);
        T* buf = reinterpret_cast<T*>(excenv_smem) + (unsigned)(n & 1) * (unsigned)(NS * NT);
#pragma unroll
        for (int q = 0; q < O; ++q) buf[q * NT + threadIdx.x] = ob[0][q];
        if constexpr (GENERAL) {
#pragma unroll
          for (int j = 0; j < EXCENV_MAX_CONTROL; ++j)
            if (j < ka.n_control) buf[(O + j) * NT + threadIdx.x] = cref[0][j];
        }
        if (with_states) {
#pragma unroll
          for (int j = 0; j < S; ++j) buf[(OWr + j) * NT + threadIdx.x] = sv[0][j];
        }
        // this wave's LDS writes must have landed before it signals: gfx950 backs barriers off instead of waiting implicitly, and the
        // compiler adds no wait in front of the raw builtin (the disassembly showed ds_write ...; s_barrier). lgkmcnt only — a
        // __syncthreads() would also drain the trajectory stores still in flight (vmcnt), which is what this path must not do
#if !(EXCENV_FAULT & 2)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
        __builtin_amdgcn_s_barrier();
        const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        const unsigned ln = threadIdx.x & 63u;
        for (int u = wv; u < NS * CH; u += NW) {
          const int q = u / CH;
          const unsigned e = (unsigned)(u % CH) * (64u * VE) + ln * VE;
          T tmp[VE];
          load_v<T, VE>(buf + q * NT + e, tmp);
          T* dst = (q < OWr) ? orow + q * ka.o_sc : ka.straj[q - OWr] + s_blk + n * ka.s_sk;
          store_stream<T, VE>(dst + e, tmp);
        }
        direct = false;
      } else if (ka.row_sync) {
        __builtin_amdgcn_s_barrier();
      }
    } else if constexpr (V == 1) {
      if (ka.row_sync) __builtin_amdgcn_s_barrier();
    }
    if (direct) {
#pragma unroll
      for (int q = 0; q < O; ++q) {
        T tmp[V];
#pragma unroll
        for (int v = 0; v < V; ++v) tmp[v] = ob[v][q];
        store_stream<T, V>(orow + q * ka.o_sc + o_lane, tmp);
      }
      if constexpr (GENERAL) {
#pragma unroll
        for (int j = 0; j < EXCENV_MAX_CONTROL; ++j) {
          if (j < ka.n_control) {
            T tmp[V];
#pragma unroll
            for (int v = 0; v < V; ++v) tmp[v] = cref[v][j];
            store_v<T, V>(orow + (O + j) * ka.o_sc + o_lane, tmp);
          }
        }
      }
      if (with_states) {
#pragma unroll
        for (int j = 0; j < S; ++j) {
          T tmp[V];
#pragma unroll
          for (int v = 0; v < V; ++v) tmp[v] = sv[v][j];
          store_stream<T, V>(ka.straj[j] + s_blk + n * ka.s_sk + s_lane, tmp);
        }
      }
    }
    if constexpr (LGYM && !M::IS_PMSM) {  // generate_reward / generate_truncated / generate_terminated of the other models, V wide
      const int64_t e0 = blk0 + lane_env;
      T rew[V];
#pragma unroll
      for (int v = 0; v < V; ++v) rew[v] = T(0);
#pragma unroll
      for (int j = 0; j < NCM; ++j) {
        if (j < ka.n_control) {
          const int f = ka.control_idx[j];
          bool ang = false;
#pragma unroll
          for (int q = 0; q < S; ++q) ang = ang || (is_angle_field<M>(q) && f == q);
#pragma unroll
          for (int v = 0; v < V; ++v) {
            T x, lo, hi;
            pick_field<M, T>(sv[v], c, f, x, lo, hi);
            if (ang) {
              T sx, cx;
              sincos_t(x, sx, cx);
              const T ds = sx - gp_a[v][j], dc = cx - gp_b[v][j];
              rew[v] = rew[v] + -(ds * ds + dc * dc);
            } else {
              const T d = normalize(x, lo, hi) - gp_a[v][j];
              rew[v] = rew[v] + -(d * d);
            }
          }
        }
      }
      uint8_t fl[V];
      if constexpr (M::ID == EXCENV_FLUID_TANK) {  // fluid_tank_env.py:325-333: constants (TW = 1)
#pragma unroll
        for (int v = 0; v < V; ++v) fl[v] = 0;
        store_flags<V>(ka.truncated + n * ka.t_sk + e0, fl);
      } else {
        // |obs| > 1 per observation column + the control columns' constant flags: the V * TW bytes of the lane's environments are
        // adjacent in the row — ONE 16-byte store for pendulum [theta] instead of four 4-byte ones (round 5; the store
        // instructions of a gym launch were 2.0 ... 2.5 x the plain launch's for 1.2 ... 1.4 x its bytes)
        dispatch_count<0, NCM>(ka.n_control, [&](auto tag) {
          constexpr int NC = decltype(tag)::value, TWc = O + NC, NB = V * TWc, NW = (NB + 3) / 4;
          uint32_t w[NW];
#pragma unroll
          for (int i = 0; i < NW; ++i) w[i] = gp_w[i];
#pragma unroll
          for (int v = 0; v < V; ++v) {
#pragma unroll
            for (int q = 0; q < O; ++q) {
              const int b = v * TWc + q;
              w[b >> 2] |= (xabs(ob[v][q]) > T(1)) ? (1u << ((b & 3) * 8)) : 0u;
            }
          }
          store_flag_bytes<NB>(ka.truncated + n * ka.t_sk + e0 * TWc, w);
        });
#pragma unroll
        for (int v = 0; v < V; ++v) fl[v] = rew[v] == T(0);  // generate_terminated: reward == 0
      }
      if (n > 0) {
        store_stream<T, V>(ka.reward + (n - 1) * ka.g_sk + e0, rew);
        store_flags<V>(ka.terminated + (n - 1) * ka.g_sk + e0, fl);
      }
    }
    if constexpr (LGYM && M::IS_PMSM) {  // the same outputs for the V environments of a lane: truncated row n, reward / terminated row n - 1
      T rew[V];
      uint8_t fl[V];
#pragma unroll
      for (int v = 0; v < V; ++v) {
        rew[v] = pmsm_reward<M, T>(sv[v], c, has_id, g_id[v], has_iq, g_iq[v], has_tq, g_tq[v]);
        const T nd = normalize(sv[v][3], c.smin[3], c.smax[3]), nq = normalize(sv[v][4], c.smin[4], c.smax[4]);
        fl[v] = sqrt_exceeds_one(nd * nd + nq * nq);  // pmsm_env.py:972-983: sqrt(.) > 1 (devmath.hpp), terminated == truncated
      }
      const int64_t e0 = blk0 + lane_env;
      store_flags<V>(ka.truncated + n * ka.t_sk + e0, fl);
      if (n > 0) {
        store_stream<T, V>(ka.reward + (n - 1) * ka.g_sk + e0, rew);
        store_flags<V>(ka.terminated + (n - 1) * ka.g_sk + e0, fl);
      }
    }
    if constexpr (GYM) {  // core_env.py:490-531: truncated on every row, reward / terminated on rows 1..N (V == 1 here)
      if (ka.truncated != nullptr) {
        const int64_t e = i0;
        uint8_t* tr = ka.truncated + e * ka.t_sb + n * ka.t_sk;
        const bool tail = n > 0;
        T* rw = tail ? ka.reward + e * ka.g_sb + (n - 1) * ka.g_sk : nullptr;
        uint8_t* te = tail ? ka.terminated + e * ka.g_sb + (n - 1) * ka.g_sk : nullptr;
        gym_outputs<M, T>(sv[0], ob[0], c, ka.n_control, ka.control_idx, rref[0], rw, te, tr, ka.t_sc);
      }
    }
  };
  auto publish_last = [&](const T (&sv)[V][S]) {
#pragma unroll
    for (int j = 0; j < S; ++j) {
      T tmp[V];
#pragma unroll
      for (int v = 0; v < V; ++v) tmp[v] = sv[v][j];
      store_v<T, V>(ka.last_state[j] + blk0 + lane_env, tmp);
    }
  };
  T sv[V][S];
  if (N == 0) {  // no action row exists (ka.actions may be NULL)
    save_row(0, sv);
    publish_last(sv);
    return;
  }

  T a0[A][V], a1[A][V];
  // (k, sub): action row and sub-step of solver step n; (kn, subn): those of step n + 1
  int64_t k = 0, kn = 0;
  int32_t sub = 0, subn = 0;
  // ---- AEM: the wave's action windows in LDS (see the comment above the kernel) ----
  constexpr int VW = 16 / (int)sizeof(T);  // elements per 16-byte piece
  constexpr int SP = AEM ? VW / A : 1;     // action rows per piece
  constexpr int NP = aem_np<M>();          // pieces per window
  constexpr int EPI = 64 / NP;             // environments (reader lanes) per load instruction
  static_assert(64 % NP == 0, "a load instruction covers whole windows");
  const unsigned wave = threadIdx.x / 64u, lane64 = threadIdx.x % 64u;
  const unsigned wave_off = AEM ? __builtin_amdgcn_readfirstlane(wave * (unsigned)(V * NP * AEM_BLOCK_BYTES)) : 0u;  // this wave's blocks
  const unsigned wave_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)excenv_smem + wave_off;  // as an LDS address
  // reader side: lane l's environment v sits in block (v, l / EPI) at window offset (l % EPI) * NP * 16
  const unsigned rd_lane = (lane64 / EPI) * AEM_BLOCK_BYTES + (lane64 % EPI) * (NP * 16u);
  // loader side: lane t serves reader lane slot t / NP of the instruction's EPI, piece t % NP
  const unsigned ld_piece = lane64 % NP;
  const uint64_t ld_lane_off = AEM ? ((uint64_t)(wave * 64u + lane64 / NP) * V) * (uint64_t)ka.a_sb : 0;  // elements, before (i, v)
  // pieces per environment row (host: K * A % VW == 0, and K * A < 2^23: the window bookkeeping below is unsigned 32-bit scalar
  // arithmetic — as int64_t every division by a power of two was a 64-bit shift with a sign fix-up, per slot and step)
  const int32_t n_pieces = AEM ? (int32_t)((ka.K * A) / VW) : 0;
  constexpr int NSTORE = O + ((STATES != 0) ? S : 0);   // trajectory stores per saved row: issued between a fill and its first read
  // Sector-aligned windows (round 5). A window is NP pieces = 64 bytes and one fabric request — if it does not straddle two
  // 64-byte sectors of memory. Rows of K * A * sizeof(T) bytes start on 16-byte boundaries only (PMSM, K = 100: 800 bytes, every
  // second environment starts in the middle of a sector), so windows counted from the row's first byte straddled for half of the
  // environments: 6.8e7 fabric reads where 5.2e7 would do, FETCH_SIZE 1.29 x the action bytes (round 4). Now the windows of an
  // environment are the SECTORS its row touches: with ph = (first piece of the row) mod NP, row piece j sits at position
  // (j + ph) % NP of window (j + ph) / NP, the first window holds NP - ph pieces (the lanes in front of it re-fetch the row's first
  // piece, never read), every further one is one aligned sector. ph depends on the environment only through its slot v of the lane
  // (the lanes' environments are V apart and V * pieces-per-row is a multiple of NP — else ph = 0 for everybody: the round-4
  // scheme), so each slot keeps its own wave-uniform window count and refills when ITS window ends.
  const bool aem_aligned = AEM && ((n_pieces * V) % NP) == 0;
  unsigned aem_ph[V];
  int32_t w_hi[V];  // highest window requested so far, per slot (wave-uniform)
#pragma unroll
  for (int v = 0; v < V; ++v) {
    aem_ph[v] = aem_aligned ? (unsigned)((((uintptr_t)a_blk >> 4) + (uint64_t)v * (uint64_t)(uint32_t)n_pieces) % NP) : 0u;
    w_hi[v] = -1;
  }
  auto dma_window = [&](int v, int32_t w) __attribute__((always_inline)) {  // v: compile-time constant at every call
    if constexpr (AEM) {
      int32_t pc = w * NP + (int32_t)ld_piece - (int32_t)aem_ph[v];
      pc = pc < 0 ? 0 : (pc < n_pieces ? pc : n_pieces - 1);  // in front of the row / behind it: a piece of the row again (never read)
      const T* lane_src = a_blk + ld_lane_off + (int64_t)pc * VW;
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        // Inline assembly, not __builtin_amdgcn_global_load_lds: the compiler treats an LDS-direct load as a FLAT access and puts
        // `s_waitcnt vmcnt(0)` in front of every later LDS read — that would drain the trajectory stores once per step. Hidden
        // from it, the only wait is the counted one in load_action below. M0 = the LDS byte address of the block: declared as
        // clobbered, and the s_nop is the wait state gfx9-family parts need between an SALU write of M0 and an LDS-direct load
        // (the compiler's hazard recognizer emits the same s_nop behind the builtin; it does not look inside an asm string).
        // tests/test_isa_guards.py checks both in the disassembly of the built library.
        const T* src = lane_src + (uint64_t)(i * EPI * V + v) * (uint64_t)ka.a_sb;
        // (M0 is a reserved register: clang warns that it "may not be preserved"; listing it is what makes the compiler's own M0
        // initialisations — s_set_gpr_idx, its LDS-direct loads — see this statement as a redefinition)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
#if EXCENV_FAULT & 4
        asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(wave_lds + (unsigned)(v * NP + i) * AEM_BLOCK_BYTES) : "memory", "m0");
#else
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(wave_lds + (unsigned)(v * NP + i) * AEM_BLOCK_BYTES)
                     : "memory", "m0");
#endif
#pragma clang diagnostic pop
      }
    }
  };
  int32_t pc_j = -1;                       // the piece held in pc_reg (wave-uniform)
  T pc_reg[AEM ? V : 1][AEM ? VW : 1];     // a slot's current 16-byte piece: SP rows of A values
  bool fill_pending = false;               // a window was requested and no counted wait has run since
  auto load_action = [&](int64_t krow, T (&dst)[A][V]) __attribute__((always_inline)) {
    if constexpr (AEM) {
      // Once per PIECE (SP rows), not per row (round 5, second half): the slot's 16-byte piece is read from its LDS window into
      // registers (one ds_read_b128), the rows are picked out of the registers, and a window whose LAST piece has just been read is
      // re-requested right away — its fill has SP rows instead of one to land, no LDS read of the wave sits behind a fill in flight
      // for SP rows, and the bookkeeping (which window, which position, last piece or not: ~110 scalar instructions per wave-step
      // when it ran per row and slot) runs once per piece. Measured: neutral to -2 % against the row-wise form on every workload
      // (same buffers) — neither the bookkeeping, nor reads queued behind a fill, nor the fill's slack is what the small models lose
      // with row-major actions (DESIGN.md §4.1b has the list of what was ruled out). Kept for what it removes.
      const uint32_t kr = (uint32_t)krow;              // 0 <= krow < K < 2^23
      const int32_t j = (int32_t)(kr / (uint32_t)SP);  // the row's piece of its environment's row
      const unsigned rs = kr % (uint32_t)SP;           // the row inside that piece
      const bool newp = j != pc_j;
      // first piece after a fill was requested (at least one row, i.e. one saved row's stores, earlier): everything but the
      // trajectory stores issued since must be back (vmcnt retires in issue order: with at least NSTORE vector-memory instructions
      // behind the fill, vmcnt(NSTORE) waits for the fill and for nothing it need not — fills of other slots issued behind it only
      // make the wait stricter; FEWER than NSTORE behind it and the wait would prove nothing — tools/isa_guards.py counts them on
      // every path of the built code). expcnt(6) never blocks here (no exports) and marks the hand-written waits for that tool.
      // Wave-uniform.
      if (newp && fill_pending) asm volatile("s_waitcnt vmcnt(%0) expcnt(6)" ::"n"((NSTORE + (EXCENV_FAULT & 1)) < 63 ? (NSTORE + (EXCENV_FAULT & 1)) : 63) : "memory");
      if (newp) {
        fill_pending = false;
        pc_j = j;
        uint32_t last_mask = 0u;  // slots whose piece is the last of a window with a successor not yet requested
        int32_t w[V];
#pragma unroll
        for (int v = 0; v < V; ++v) {
          const uint32_t q = (uint32_t)j + aem_ph[v];
          w[v] = (int32_t)(q / (uint32_t)NP);
          const unsigned pos = q % (uint32_t)NP;
          load_v<T, VW>(reinterpret_cast<const T*>(excenv_smem + wave_off + (unsigned)(v * NP) * AEM_BLOCK_BYTES + rd_lane + pos * 16u), pc_reg[v]);
          // (w_hi: once per window, whatever the clamped tail of the trajectory repeats)
          if (pos == NP - 1 && w[v] + 1 > w_hi[v] && (w[v] + 1) * NP - (int32_t)aem_ph[v] < n_pieces) last_mask |= 1u << v;
        }
        if (last_mask != 0u) {  // those windows' LDS is dead once the reads above have returned -> request their successors into it
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
          for (int v = 0; v < V; ++v) {
            if ((last_mask >> v) & 1u) {
              dma_window(v, w[v] + 1);
              w_hi[v] = w[v] + 1;
            }
          }
          fill_pending = true;
        }
      }
      // the row out of the registers (rs is wave-uniform: SP - 1 selects per value)
#pragma unroll
      for (int v = 0; v < V; ++v) {
#pragma unroll
        for (int q = 0; q < A; ++q) {
          T val = pc_reg[v][q];
#pragma unroll
          for (int r = 1; r < SP; ++r) {
            const T alt = pc_reg[v][r * A + q];  // (a value, not an lvalue: a ternary between two lvalues becomes a pointer select -> scratch)
            val = (rs == (unsigned)r) ? alt : val;
          }
          dst[q][v] = val;
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < A; ++q) load_v<T, V>(a_blk + krow * ka.a_sk + q * ka.a_sc + a_lane, dst[q]);
    }
  };
  auto advance = [&](const T (&cur)[A][V], const T (&nxt)[A][V], int64_t k, int64_t k1) __attribute__((always_inline)) {
#pragma unroll
    for (int v = 0; v < V; ++v) {
      T ac[A], an[A];
#pragma unroll
      for (int q = 0; q < A; ++q) {
        ac[q] = cur[q][v];
        an[q] = nxt[q][v];
      }
      if constexpr (AHEAD) {
        env_advance_raw<M, SOLVER>(st[v], ac, an, k, k1, EXCENV_CX(v), aux[v], M::HAS_LUT ? &memo[v] : nullptr);
      } else {
        env_step<M, SOLVER>(st[v], ac, EXCENV_CX(v), M::HAS_LUT ? &memo[v] : nullptr);
      }
    }
  };
  auto next_index = [&]() {
    kn = k;
    subn = sub + 1;
    if (subn == ka.substeps) { subn = 0; kn = k + 1; }
  };
  const int64_t klast = ka.K - 1;
  if constexpr (AEM) {  // the first window of every environment
#pragma unroll
    for (int v = 0; v < V; ++v) {
      dma_window(v, 0);
      w_hi[v] = 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // once per trajectory (the initial state has arrived as well)
  }
  load_action(0, a0);
  if constexpr (AEM) {
    // A slot whose first window holds a single piece has re-requested it already, and the loop's first load_action follows without a
    // saved row in between: nothing would stand behind that fill for the counted wait to count (found by tools/isa_guards.py on the
    // piece-wise form; the row-wise form of rounds 4 - 5 had the same hole for one-row pieces — PMSM fp64 — and an action pointer that
    // is 16- but not 64-byte aligned). Once per trajectory: wait for it outright.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    fill_pending = false;
  }
  // look-up models keep the single-step loop: twice the (large) look-up code does not fit the instruction cache
  constexpr bool PINGPONG = !M::HAS_LUT && ((SOLVER == EXCENV_EULER) ? (EXCENV_PINGPONG & 1) : (EXCENV_PINGPONG & 2)) != 0;
  if constexpr (PINGPONG) {
    for (int64_t n = 0;; n += 2) {
      // even step: a0 holds action row k; the row of step n + 1 goes to a1 (clamped: always a valid row, so the load is
      // unconditional and its result needs no merge with an undefined value)
      next_index();
      int64_t k1 = (kn < klast) ? kn : klast;
      load_action(k1, a1);
      save_row(n, sv);
      if (n == N) break;
      advance(a0, a1, k, k1);
      k = kn;
      sub = subn;
      // odd step: roles swapped
      next_index();
      k1 = (kn < klast) ? kn : klast;
      load_action(k1, a0);
      save_row(n + 1, sv);
      if (n + 1 == N) break;
      advance(a1, a0, k, k1);
      k = kn;
      sub = subn;
    }
  } else {
    for (int64_t n = 0;; ++n) {
      // the row of step n + 1 is requested before row n is saved (clamped: always a valid row, so the load is unconditional);
      // it is first needed by the register rotation after the compute phase
      next_index();
      const int64_t k1 = (kn < klast) ? kn : klast;
      load_action(k1, a1);
      save_row(n, sv);
      if (n == N) break;
      advance(a0, a1, k, k1);
#pragma unroll
      for (int q = 0; q < A; ++q)
#pragma unroll
        for (int v = 0; v < V; ++v) a0[q][v] = a1[q][v];
      k = kn;
      sub = subn;
    }
  }
  publish_last(sv);
#undef EXCENV_CX
}

// ---- reference-tracking observation columns of a trajectory (control_state; e.g. pendulum_env.py:311-329) --------------
// They are constant along the trajectory: the normalised reference of each controlled field, repeated on every row. With
// broadcast properties and no gym outputs the trajectory itself is produced by the lean (vectorised) sim_ahead_kernel, which
// leaves these columns of the observation buffer untouched, and this kernel fills them: one 16-byte store per (4 environments,
// row, column) in the lane-major / tiled layouts. Same values as the GENERAL kernel writes (plain normalize()).
template <typename T, class M> struct ControlFillArgs {
  KProps<T, M> kp;
  int64_t B, rows;
  int32_t n_control, tiled;
  int32_t control_idx[EXCENV_MAX_CONTROL];
  const T* reference[EXCENV_MAX_CONTROL];
  T* obs;
  int64_t o_sk, o_sc, tile_pitch;  // element strides: row, column; tiled: elements per tile
};

template <class M, typename T, int V> __global__ void __launch_bounds__(BLOCK) control_fill_kernel(const ControlFillArgs<T, M> ka) {
  const int64_t b0 = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) * V;
  if (b0 >= ka.B) return;
  T* base = ka.obs + (ka.tiled ? (b0 / EXCENV_TILE) * ka.tile_pitch + (b0 % EXCENV_TILE) : b0);
#pragma unroll
  for (int j = 0; j < EXCENV_MAX_CONTROL; ++j) {
    if (j < ka.n_control) {
      const int f = ka.control_idx[j];
      const T lo = ka.kp.scalar[M::P + f], hi = ka.kp.scalar[M::P + M::S + f];
      T r[V], c[V];
      load_v<T, V>(ka.reference[j] + b0, r);
#pragma unroll
      for (int v = 0; v < V; ++v) c[v] = normalize(r[v], lo, hi);
      T* col = base + (M::O + j) * ka.o_sc;
      for (int64_t n = blockIdx.y; n < ka.rows; n += gridDim.y) store_stream<T, V>(col + n * ka.o_sk, c);
    }
  }
}

// ---- generate_rew_trunc_term_ahead on a stored trajectory (core_env.py:490-531, 618-647) -----------------------------
// One thread per (env, row) of the state trajectories a previous vmap_sim_ahead returned (any of its layouts: the host
// passes element strides and says which of the two indices is the contiguous one). reward / terminated are written for
// rows 1.. at index row-1, truncated for every row. References may vary along the trajectory (stride 0 = constant).
template <typename T, class M> struct TrajGymArgs {
  KProps<T, M> kp;
  int64_t B, rows;
  int32_t n_control, fast_is_env;
  int32_t control_idx[EXCENV_MAX_CONTROL];
  const T* reference[EXCENV_MAX_CONTROL];
  int64_t r_sb[EXCENV_MAX_CONTROL], r_sk[EXCENV_MAX_CONTROL];
  const T* straj[M::S];
  int64_t s_sb, s_sk;
  T* reward;
  uint8_t* terminated;
  uint8_t* truncated;
  int64_t g_sb, g_sk, t_sb, t_sk, t_sc;
};

template <class M, typename T> __global__ void __launch_bounds__(BLOCK) traj_gym_kernel(const TrajGymArgs<T, M> ka) {
  constexpr int S = M::S, O = M::O;
  const int64_t nfast = ka.fast_is_env ? ka.B : ka.rows;
  const int64_t nb_fast = (nfast + BLOCK - 1) / BLOCK;
  const int64_t slow = (int64_t)blockIdx.x / nb_fast;
  const int64_t fast = ((int64_t)blockIdx.x - slow * nb_fast) * BLOCK + threadIdx.x;
  if (fast >= nfast) return;
  const int64_t b = ka.fast_is_env ? fast : slow, n = ka.fast_is_env ? slow : fast;
  Ctx<T, M> c;
  load_ctx<true, T, M, false>(c, ka.kp, b, T(0), T(0), T(0));
  T st[S], ob[O], rref[EXCENV_MAX_CONTROL];
#pragma unroll
  for (int j = 0; j < S; ++j) st[j] = ka.straj[j][b * ka.s_sb + n * ka.s_sk];
#pragma unroll
  for (int j = 0; j < EXCENV_MAX_CONTROL; ++j)
    rref[j] = (j < ka.n_control) ? ka.reference[j][b * ka.r_sb[j] + n * ka.r_sk[j]] : T(0);
  M::observe(st, c, ob);
  const bool tail = n > 0;
  gym_outputs<M, T>(st, ob, c, ka.n_control, ka.control_idx, rref, tail ? ka.reward + b * ka.g_sb + (n - 1) * ka.g_sk : nullptr,
                    tail ? ka.terminated + b * ka.g_sb + (n - 1) * ka.g_sk : nullptr,
                    ka.truncated + b * ka.t_sb + n * ka.t_sk, ka.t_sc);
}

// ---- generate_state_from_observation (e.g. pendulum_env.py:331-364; PMSM pmsm_env.py:921-970) -------------------------
// obs [B][O + n_control] row-major -> S physical-state leaves [B] (denormalised) and the reference leaves of the
// controlled fields. PMSM recovers eps from atan2(sin, cos) / pi before denormalising.
template <typename T, class M> struct FromObsArgs {
  KProps<T, M> kp;
  int64_t B;
  int32_t n_control;
  int32_t control_idx[EXCENV_MAX_CONTROL];
  const T* obs;
  T* state_out[M::S];
  T* reference_out[EXCENV_MAX_CONTROL];
};

__device__ __forceinline__ float xatan2(float y, float x) { return ::atan2f(y, x); }
__device__ __forceinline__ double xatan2(double y, double x) { return ::atan2(y, x); }

template <class M, typename T> __global__ void __launch_bounds__(BLOCK) from_obs_kernel(const FromObsArgs<T, M> ka) {
  constexpr int S = M::S, O = M::O;
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= ka.B) return;
  Ctx<T, M> c;
  load_ctx<true, T, M, false>(c, ka.kp, i, T(0), T(0), T(0));
  const T* row = ka.obs + i * (O + ka.n_control);
  T nrm[S];
  if constexpr (M::IS_PMSM) {  // obs = [i_d, i_q, omega_el, torque, cos eps, sin eps, u_d_buffer, u_q_buffer]
    nrm[0] = row[6];
    nrm[1] = row[7];
    nrm[2] = xatan2(row[5], row[4]) / K<T>::pi;
    nrm[3] = row[0];
    nrm[4] = row[1];
    nrm[5] = row[3];
    nrm[6] = row[2];
  } else {
#pragma unroll
    for (int j = 0; j < S; ++j) nrm[j] = row[j];
  }
#pragma unroll
  for (int j = 0; j < S; ++j) ka.state_out[j][i] = denormalize(nrm[j], c.smin[j], c.smax[j]);
#pragma unroll
  for (int j = 0; j < EXCENV_MAX_CONTROL; ++j) {
    if (j < ka.n_control) {
      const int f = ka.control_idx[j];
      T lo = c.smin[0], hi = c.smax[0];
#pragma unroll
      for (int q = 1; q < S; ++q) {
        lo = (f == q) ? c.smin[q] : lo;
        hi = (f == q) ? c.smax[q] : hi;
      }
      ka.reference_out[j][i] = denormalize(row[O + j], lo, hi);
    }
  }
}

// ---- generate_observation on a batch of states (e.g. pendulum_env.py:311-329, PMSM pmsm_env.py:898-919) ----------------------
// obs[i] = M::observe(state i) followed by the normalised reference of every controlled field: the same device function the
// step and trajectory kernels fuse, as a launch of its own (vmap_reset, user code that builds states by hand).
template <typename T, class M> struct ObserveArgs {
  KProps<T, M> kp;
  int64_t B;
  int32_t n_control;
  int32_t control_idx[EXCENV_MAX_CONTROL];
  const T* state[M::S];
  const T* reference[EXCENV_MAX_CONTROL];
  T* obs;
};

template <class M, typename T> __global__ void __launch_bounds__(BLOCK) observe_kernel(const ObserveArgs<T, M> ka) {
  constexpr int S = M::S, O = M::O;
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= ka.B) return;
  Ctx<T, M> c;
  load_ctx<true, T, M, false>(c, ka.kp, i, T(0), T(0), T(0));
  T st[S], ob[O];
#pragma unroll
  for (int j = 0; j < S; ++j) st[j] = ka.state[j][i];
  M::observe(st, c, ob);
  T* row = ka.obs + i * (O + ka.n_control);
#pragma unroll
  for (int j = 0; j < O; ++j) row[j] = ob[j];
#pragma unroll
  for (int j = 0; j < EXCENV_MAX_CONTROL; ++j) {
    if (j < ka.n_control) {
      const int f = ka.control_idx[j];
      T lo = c.smin[0], hi = c.smax[0];
#pragma unroll
      for (int q = 1; q < S; ++q) {
        lo = (f == q) ? c.smin[q] : lo;
        hi = (f == q) ? c.smax[q] : hi;
      }
      row[O + j] = normalize(ka.reference[j][i], lo, hi);
    }
  }
}

// ---- probes for the in-kernel math (tests) ---------------------------------------------------
template <typename T> __global__ void probe_kernel(int which, int64_t n, const T* in, T* out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const T x = in[i];
  T r;
  if (which == 0) r = sin_t(x);
  else if (which == 1) r = cos_t(x);
  else r = wrap_angle(x);
  out[i] = r;
}

// out_fast[i] = InvDiv(den[i]).div(num[i]), out_ref[i] = num[i] / den[i] (the compiler's IEEE sequence) — must be equal bits
template <typename T> __global__ void probe_div_kernel(int64_t n, const T* num, const T* den, T* out_fast, T* out_ref) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  InvDiv<T> d;
  d.init(den[i], true);
  out_fast[i] = d.div(num[i], true);
  out_ref[i] = num[i] / den[i];
}

}  // namespace excenv
