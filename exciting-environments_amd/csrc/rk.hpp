// Fixed-step explicit Runge-Kutta steppers, fully unrolled at compile time so stage values live in VGPRs.
// Euler follows diffrax.Euler.step: y1 = y0 + f(y0)*dt. RK4 / Tsit5: k_j = f(y_j,u_j)*dt,
// y_i = y0 + sum_j a_ij k_j, y1 = y0 + sum_j b_j k_j with the sums accumulated left to right by explicit fma,
// zero coefficients skipped — the same definition as oracle/oracle_body.inc rk_step.
#pragma once
#include "models.hpp"

namespace excenv {

template <int SOLVER> struct Tableau;

template <> struct Tableau<EXCENV_RK4> {
  static constexpr int NS = 4;
  __host__ __device__ static constexpr double a(int s, int q) {
    constexpr double A[4][4] = {{0, 0, 0, 0}, {0.5, 0, 0, 0}, {0.0, 0.5, 0, 0}, {0.0, 0.0, 1.0, 0}};
    return A[s][q];
  }
  __host__ __device__ static constexpr double b(int q) {
    constexpr double B[4] = {1.0 / 6.0, 1.0 / 3.0, 1.0 / 3.0, 1.0 / 6.0};
    return B[q];
  }
  __host__ __device__ static constexpr bool c_is_one(int s) { return s == 3; }
};

// Tsitouras 5(4), first six stages (b7 = 0; the FSAL stage only feeds the unused error estimate).
template <> struct Tableau<EXCENV_TSIT5> {
  static constexpr int NS = 6;
  __host__ __device__ static constexpr double a(int s, int q) {
    constexpr double A[6][6] = {
        {0, 0, 0, 0, 0, 0},
        {0.161, 0, 0, 0, 0, 0},
        {-0.008480655492356989, 0.335480655492357, 0, 0, 0, 0},
        {2.8971530571054935, -6.359448489975075, 4.3622954328695815, 0, 0, 0},
        {5.325864828439257, -11.74888356406283, 7.4955393428898365, -0.09249506636175525, 0, 0},
        {5.86145544294642, -12.92096931784711, 8.159367898576159, -0.071584973281401, -0.028269050394068383, 0}};
    return A[s][q];
  }
  __host__ __device__ static constexpr double b(int q) {
    constexpr double B[6] = {0.09646076681806523, 0.01, 0.4798896504144996,
                             1.379008574103742,   -3.290069515436081, 2.324710524099774};
    return B[q];
  }
  __host__ __device__ static constexpr bool c_is_one(int s) { return s == 5; }
};

// u: action held over the step; u1_of(u1): fills the action seen by stages with c_i == 1 (== u on the step path). It is a
// callable evaluated AT that stage: on the sim_ahead path it reads the prefetched next action row, and the s_waitcnt for that
// load sits in front of the first instruction that touches those registers — evaluated up front, every solver step would
// start by waiting for a load issued a few hundred instructions earlier.
// q0 (look-up models only): the interpolated table values at the step's starting point when the caller already has them.
template <class M, int SOLVER, typename T, class U1>
__device__ __forceinline__ void rk_step(T (&y)[M::NY], const T (&u)[M::A], U1&& u1_of, const Ctx<T, M>& c,
                                        const T (&st)[M::S], const T (*q0)[6] = nullptr) {
  constexpr int NY = M::NY;
  T dy[NY];
  auto f_first = [&](const T (&yy)[NY], T (&d)[NY]) __attribute__((always_inline)) {
    if constexpr (M::HAS_LUT) {
      if (q0 != nullptr) {
        M::f_q(yy, u, c, st, d, *q0);
        return;
      }
    }
    M::f(yy, u, c, st, d);
  };
  if constexpr (SOLVER == EXCENV_EULER) {
    f_first(y, dy);
#pragma unroll
    for (int j = 0; j < NY; ++j) y[j] = y[j] + dy[j] * c.dt;
  } else {
    using TB = Tableau<SOLVER>;
    constexpr int NS = TB::NS;
    T k[NS][NY], yi[NY];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
      for (int j = 0; j < NY; ++j) {
        T acc = T(0);
#pragma unroll
        for (int q = 0; q < s; ++q) {
          const T a = T(TB::a(s, q));
          if (a != T(0)) acc = xfma(a, k[q][j], acc);
        }
        yi[j] = (s == 0) ? y[j] : y[j] + acc;
      }
      if (TB::c_is_one(s)) {
        T u1[M::A];
        u1_of(u1);
        M::f(yi, u1, c, st, dy);
      } else if (s == 0) {
        f_first(yi, dy);
      } else {
        M::f(yi, u, c, st, dy);
      }
#pragma unroll
      for (int j = 0; j < NY; ++j) k[s][j] = dy[j] * c.dt;
    }
#pragma unroll
    for (int j = 0; j < NY; ++j) {
      T acc = T(0);
#pragma unroll
      for (int q = 0; q < NS; ++q) {
        const T b = T(TB::b(q));
        if (b != T(0)) acc = xfma(b, k[q][j], acc);
      }
      y[j] = y[j] + acc;
    }
  }
}

// One reference `step` on the carried state (CoreEnvironment.step core_env.py:393-425; PMSM.step pmsm_env.py:851-883).
// memo (look-up models, trajectory kernels): in = the table values at the state's operating point, out = those at the new one.
template <class M, int SOLVER, typename T>
__device__ __forceinline__ void env_step(T (&st)[M::S], const T (&a)[M::A], const Ctx<T, M>& c, T (*memo)[6] = nullptr) {
  T u[M::A];
  if constexpr (M::IS_PMSM) {
    T uc[2];
    M::constraint(a, st[2], st[6], c, uc);
    if (c.P[6] > T(0)) {  // deadtime: apply the buffered voltage, buffer the new one
      u[0] = st[0]; u[1] = st[1];
      st[0] = uc[0]; st[1] = uc[1];
    } else {
      u[0] = uc[0]; u[1] = uc[1];
    }
  } else {
    u[0] = denormalize(a[0], c.amin[0], c.amax[0]);
  }
  T y[M::NY];
  M::get_y(st, y);
  rk_step<M, SOLVER>(y, u, [&](T (&u1)[M::A]) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < M::A; ++q) u1[q] = u[q];
  }, c, st, memo);
  M::set_y(st, y);
  if constexpr (M::HAS_LUT) {
    if (memo != nullptr) {
      M::post_q(st, c, *memo);
      return;
    }
  }
  M::post(st, c);
}

// Extra carried values for the reference's sim_ahead structure (only PMSM needs any).
template <typename T> struct AheadAux {
  T eps0, prev_clip[2];  // prev_clip starts as the initial voltage buffer, so row 0 of the trajectory shows that buffer
};

// jnp.linspace(0, tau * (K - 1), K)[k] as JAX evaluates it (jax.numpy.linspace: start * (1 - s) + stop * s with s = k / (K - 1)
// for k < K - 1, the end point itself for k == K - 1; start == 0 contributes +0). Wave-uniform: k and K are.
template <typename T, class M, typename I> __device__ __forceinline__ T ahead_time(I k, const Ctx<T, M>& c) {
  return (k == c.lin_last) ? c.lin_stop : c.lin_stop * (T(k) / c.lin_div);
}

// One solver step of the raw ODE state, reference _ode_solver_simulate_ahead structure (SEM_AHEAD):
// no wrap / clip of the carried state; PMSM clips with the predicted angle eps0 + linspace(0, tau*(K-1), K)[k] * omega
// (pmsm_env.py:719-722) and applies actions_dead[k] (:766-777).
// I: the integer type of the action row indices (int64_t, or int in kernels whose host side bounds K: a 32-bit index converts to
// T in one instruction and compares in the scalar unit; same values, same bits).
template <class M, int SOLVER, typename T, typename I>
__device__ __forceinline__ void env_advance_raw(T (&st)[M::S], const T (&a)[M::A], const T (&a1)[M::A], I k,
                                                I k1, const Ctx<T, M>& c, AheadAux<T>& aux,
                                                const T (*q0)[6] = nullptr) {
  T u[M::A];
  T uc[2] = {T(0), T(0)};
  bool dead = false;
  if constexpr (M::IS_PMSM) {
    M::constraint(a, aux.eps0 + ahead_time(k, c) * st[6], st[6], c, uc);
    dead = c.P[6] > T(0);
    u[0] = dead ? aux.prev_clip[0] : uc[0];
    u[1] = dead ? aux.prev_clip[1] : uc[1];
    aux.prev_clip[0] = uc[0];
    aux.prev_clip[1] = uc[1];
  } else {
    u[0] = denormalize(a[0], c.amin[0], c.amax[0]);
  }
  // the action of stages with c_i == 1: row k1 (== k inside an action's sub-steps), read only when that stage is reached
  auto u1_of = [&](T (&u1)[M::A]) __attribute__((always_inline)) {
    if constexpr (M::IS_PMSM) {
      if (dead) {
        u1[0] = (k1 == k) ? u[0] : uc[0];
        u1[1] = (k1 == k) ? u[1] : uc[1];
      } else {
        M::constraint(a1, aux.eps0 + ahead_time(k1, c) * st[6], st[6], c, u1);
      }
    } else {
      u1[0] = denormalize(a1[0], c.amin[0], c.amax[0]);
    }
  };
  T y[M::NY];
  M::get_y(st, y);
  rk_step<M, SOLVER>(y, u, u1_of, c, st, q0);
  M::set_y(st, y);
}

}  // namespace excenv
