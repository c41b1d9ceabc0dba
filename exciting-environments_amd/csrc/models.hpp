// The six vector fields and their step pre/post-processing, one lane per environment instance.
// Each expression keeps the reference's operation order (citations into /root/reference/exciting_environments).
#pragma once
#include "../../include/excenv.h"
#include "devmath.hpp"

namespace excenv {

// Per-lane (or wave-uniform, when no property is batched) environment properties.
template <typename T, class M> struct Ctx {
  T P[M::P];
  T smin[M::S], smax[M::S];
  T amin[M::A], amax[M::A];
  T dt;        // solver step (obs_stepsize; == tau on the step path)
  T env_tau;   // self.tau (PMSM angle prediction)
  T adv_coef;  // PMSM: (deadtime + 0.5) * tau, folded in double on the host (pmsm_env.py:599-604)
  // PMSM sim_ahead: jnp.linspace(0, tau * (K - 1), K) of the predicted angles (pmsm_env.py:719-722) — its end point
  // tau * (K - 1) (folded in double on the host like the Python expression), K - 1 as a float and as an index
  T lin_stop, lin_div;
  int64_t lin_last;
  // PMSM saturated model only: LUT grids and the node-interleaved tables [n_d][n_q][8]
  const T* lut_gd;
  const T* lut_gq;
  const T* lut_tab;
  int lut_nd, lut_nq;
  int lut_lds;  // tables + grids staged in dynamic LDS (layout: tables, grid_d, grid_q, recip_d, recip_q)
  T lut_g0[2], lut_sc[2];  // per grid (d, q): first node and (n - 1) / (last - first) — the arithmetic cell guess of find()
  // loop-invariant denominators (devmath.hpp InvDiv): (smax - smin) of every state field (normalisation) and the model's own
  InvDiv<T> nrm[M::S];
  InvDiv<T> den[M::ND > 0 ? M::ND : 1];
  bool fastdiv;  // compile-time constant per kernel (set by load_ctx): trajectory kernels true, step kernel false
};

// utils.py:13-17 with the state field's precomputed (smax - smin): same operation order, same bits as normalize()
template <typename T, class M> __device__ __forceinline__ T normalize_field(const Ctx<T, M>& c, int j, T x) {
  return c.nrm[j].div(T(2) * (x - c.smin[j]), c.fastdiv) - T(1);
}

// N state fields normalised at once (one validity test for all their divisions): out[j] = normalize(x[j]; field F[j])
template <int N, typename T, class M>
__device__ __forceinline__ void normalize_fields(const Ctx<T, M>& c, const int (&F)[N], const T (&x)[N], T (&out)[N]) {
  const InvDiv<T>* d[N];
  T num[N];
#pragma unroll
  for (int j = 0; j < N; ++j) {
    d[j] = &c.nrm[F[j]];
    num[j] = T(2) * (x[j] - c.smin[F[j]]);
  }
  div_all<N, T>(d, num, out, c.fastdiv);
#pragma unroll
  for (int j = 0; j < N; ++j) out[j] = out[j] - T(1);
}

template <typename T, class M> __device__ __forceinline__ void prep_ctx(Ctx<T, M>& c) {
#pragma unroll
  for (int j = 0; j < M::S; ++j) c.nrm[j].init(c.smax[j] - c.smin[j], c.fastdiv);
  M::prep(c);
}

// ---- Pendulum: pendulum_env.py:144-150,188 ; P = (g,l,m) --------------------------------
template <typename T> struct Pendulum {
  static constexpr int ID = EXCENV_PENDULUM, S = 2, A = 1, O = 2, P = 3, NY = 2, ND = 1;
  static constexpr bool IS_PMSM = false;
  static constexpr bool HAS_LUT = false;
  using C = Ctx<T, Pendulum>;
  __device__ static __forceinline__ void prep(C& c) { c.den[0].init(c.P[2] * (c.P[1] * c.P[1]), c.fastdiv); }  // m * (l * l)
  __device__ static __forceinline__ void get_y(const T (&st)[S], T (&y)[NY]) { y[0] = st[0]; y[1] = st[1]; }
  __device__ static __forceinline__ void set_y(T (&st)[S], const T (&y)[NY]) { st[0] = y[0]; st[1] = y[1]; }
  __device__ static __forceinline__ void f(const T (&y)[NY], const T (&u)[A], const C& c, const T (&)[S], T (&dy)[NY]) {
    const T g = c.P[0], l = c.P[1], m = c.P[2];
    dy[1] = c.den[0].div(u[0] + l * m * g * sin_t(y[0]), c.fastdiv);  // / (m * (l * l))
    dy[0] = y[1];
  }
  __device__ static __forceinline__ void post(T (&st)[S], const C&) { st[0] = wrap_angle(st[0]); }
  __device__ static __forceinline__ void observe(const T (&st)[S], const C& c, T (&ob)[O]) {
    const int F[2] = {0, 1};
    normalize_fields<2>(c, F, st, ob);
  }
};

// ---- MassSpringDamper: mass_spring_damper_env.py:142-148 ; P = (d,k,m) ---------------------
template <typename T> struct MassSpringDamper {
  static constexpr int ID = EXCENV_MASS_SPRING_DAMPER, S = 2, A = 1, O = 2, P = 3, NY = 2, ND = 1;
  static constexpr bool IS_PMSM = false;
  static constexpr bool HAS_LUT = false;
  using C = Ctx<T, MassSpringDamper>;
  __device__ static __forceinline__ void prep(C& c) { c.den[0].init(c.P[2], c.fastdiv); }  // m
  __device__ static __forceinline__ void get_y(const T (&st)[S], T (&y)[NY]) { y[0] = st[0]; y[1] = st[1]; }
  __device__ static __forceinline__ void set_y(T (&st)[S], const T (&y)[NY]) { st[0] = y[0]; st[1] = y[1]; }
  __device__ static __forceinline__ void f(const T (&y)[NY], const T (&u)[A], const C& c, const T (&)[S], T (&dy)[NY]) {
    const T d = c.P[0], k = c.P[1];
    dy[1] = c.den[0].div(u[0] - d * y[1] - k * y[0], c.fastdiv);  // / m
    dy[0] = y[1];
  }
  __device__ static __forceinline__ void post(T (&)[S], const C&) {}
  __device__ static __forceinline__ void observe(const T (&st)[S], const C& c, T (&ob)[O]) {
    const int F[2] = {0, 1};
    normalize_fields<2>(c, F, st, ob);
  }
};

// ---- CartPole: cart_pole_env.py:159-180,229 ; P = (mu_p,mu_c,l,m_p,m_c,g) -----------------
template <typename T> struct CartPole {
  static constexpr int ID = EXCENV_CART_POLE, S = 4, A = 1, O = 4, P = 6, NY = 4, ND = 2;
  static constexpr bool IS_PMSM = false;
  static constexpr bool HAS_LUT = false;
  using C = Ctx<T, CartPole>;
  __device__ static __forceinline__ void prep(C& c) {
    c.den[0].init(c.P[4] + c.P[3], c.fastdiv);  // m_c + m_p
    c.den[1].init(c.P[3] * c.P[2], c.fastdiv);  // m_p * l
  }
  __device__ static __forceinline__ void get_y(const T (&st)[S], T (&y)[NY]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) y[j] = st[j];
  }
  __device__ static __forceinline__ void set_y(T (&st)[S], const T (&y)[NY]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) st[j] = y[j];
  }
  __device__ static __forceinline__ void f(const T (&y)[NY], const T (&u)[A], const C& c, const T (&)[S], T (&dy)[NY]) {
    const T mu_p = c.P[0], mu_c = c.P[1], l = c.P[2], m_p = c.P[3], g = c.P[5];
    const T velocity = y[1], theta = y[2], omega = y[3];
    T s, co;
    sincos_t(theta, s, co);
    const auto& by_mass = c.den[0];  // / (m_c + m_p)
    const T d_omega =
        (g * s + co * by_mass.div(-u[0] - m_p * l * (omega * omega) * s + mu_c * sign_of(velocity), c.fastdiv) -
         c.den[1].div(mu_p * omega, c.fastdiv)) /
        (l * (T(4.0 / 3.0) - by_mass.div(m_p * (co * co), c.fastdiv)));
    const T d_velocity = by_mass.div(u[0] + m_p * l * ((omega * omega) * s - d_omega * co) - mu_c * sign_of(velocity), c.fastdiv);
    dy[0] = velocity;
    dy[1] = d_velocity;
    dy[2] = omega;
    dy[3] = d_omega;
  }
  __device__ static __forceinline__ void post(T (&st)[S], const C&) { st[2] = wrap_angle(st[2]); }
  __device__ static __forceinline__ void observe(const T (&st)[S], const C& c, T (&ob)[O]) {
    const int F[4] = {0, 1, 2, 3};
    normalize_fields<4>(c, F, st, ob);
  }
};

// EXCENV_ACROBOT_ANGLE_SUM = 1 (build option, fp32 only, OFF by default): cos(theta_1 + pi/2) and cos(theta_1 + theta_2 + pi/2)
// through -sin(theta_1) and -(sin theta_1 cos theta_2 + cos theta_1 sin theta_2) — one more sincos instead of two cos_t. Measured in
// round 5 (same-session A/B, B = 2^22): acrobot Tsit5 8.97 -> 8.50 ms (0.211 -> 0.2225 of the HBM roof, +5.4 %), acrobot Euler flat.
// Against the literal forms: identical for the first 10 rows of the reference's fixture, 1.2e-7 (full scale) at row 64, 3.6e-4 at
// row 10 000 (both forms sit 1.5e-3 ... 1.9e-3 from the fp64 fixture there: the double pendulum's own amplification). It stays off
// because the reference's expression rounds theta + pi/2 FIRST: at unwrapped angles of 1e4 ... 3e5 (sim_ahead integrates the raw
// angle) that rounding is up to 0.015 rad, the identity does not reproduce it, and the oracle comparison at such angles
// (test_unwrapped_angles_far_outside_the_principal_range) leaves its tolerance (2.1e-4 against 2e-4). Parity before 5 %.
#ifndef EXCENV_ACROBOT_ANGLE_SUM
#define EXCENV_ACROBOT_ANGLE_SUM 0
#endif
// ---- Acrobot: acrobot_env.py:171-197,247-248 ; P = (g,l_1,l_2,m_1,m_2,l_c1,l_c2,I_1,I_2) ----
template <typename T> struct Acrobot {
  static constexpr int ID = EXCENV_ACROBOT, S = 4, A = 1, O = 4, P = 9, NY = 4, ND = 0;
  static constexpr bool IS_PMSM = false;
  static constexpr bool HAS_LUT = false;
  using C = Ctx<T, Acrobot>;
  __device__ static __forceinline__ void prep(C&) {}  // every denominator of the vector field depends on theta_2
  __device__ static __forceinline__ void get_y(const T (&st)[S], T (&y)[NY]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) y[j] = st[j];
  }
  __device__ static __forceinline__ void set_y(T (&st)[S], const T (&y)[NY]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) st[j] = y[j];
  }
  __device__ static __forceinline__ void f(const T (&y)[NY], const T (&u)[A], const C& c, const T (&)[S], T (&dy)[NY]) {
    const T g = c.P[0], l_1 = c.P[1], m_1 = c.P[3], m_2 = c.P[4], l_c1 = c.P[5], l_c2 = c.P[6], I_1 = c.P[7], I_2 = c.P[8];
    const T theta_1 = y[0], theta_2 = y[1], omega_1 = y[2], omega_2 = y[3];
    T s2, c2;
    sincos_t(theta_2, s2, c2);
    const T d_11 = m_1 * (l_c1 * l_c1) + m_2 * (l_1 * l_1 + l_c2 * l_c2 + T(2) * l_1 * l_c2 * c2) + I_1 + I_2;
    const T d_12 = m_2 * (l_c2 * l_c2 + l_1 * l_c2 * c2) + I_2;
    const T d_22 = m_2 * (l_c2 * l_c2) + I_2;
    const T h_1 = -m_2 * l_1 * l_c2 * s2 * (omega_2 * omega_2) - T(2) * m_2 * l_1 * l_c2 * s2 * omega_1 * omega_2;
    const T h_2 = m_2 * l_1 * l_c2 * s2 * (omega_1 * omega_1);
    T cA, cB;
    if constexpr (EXCENV_ACROBOT_ANGLE_SUM != 0 && sizeof(T) == 4) {
      // cos(x + pi/2) = -sin x and cos(x + y + pi/2) = -(sin x cos y + cos x sin y): ONE more sincos instead of two cos_t, whose
      // argument sums each round first (the reference's expression, acrobot_env.py:182-183, kept literally in fp64 and in the
      // oracle). Deviates from the literal form by the rounding of theta + pi/2 (~1e-7 in the argument): DESIGN.md §6.2b.
      T s1, c1;
      sincos_t(theta_1, s1, c1);
      cA = -s1;
      cB = -(s1 * c2 + c1 * s2);
    } else {
      cA = cos_t(theta_1 + K<T>::half_pi);
      cB = cos_t(theta_1 + theta_2 + K<T>::half_pi);
    }
    const T phi_1 = (m_1 * l_c1 + m_2 * l_1) * g * cA + m_2 * l_c2 * g * cB;
    const T phi_2 = m_2 * l_c2 * g * cB;
    const T d_omega_1 = T(1) / (d_12 - d_22 / d_12 * d_11) * (u[0] + d_22 / d_12 * (h_1 + phi_1) - h_2 - phi_2);
    const T d_omega_2 = (-d_11 * d_omega_1 - h_1 - phi_1) / d_12;
    dy[0] = omega_1;
    dy[1] = omega_2;
    dy[2] = d_omega_1;
    dy[3] = d_omega_2;
  }
  __device__ static __forceinline__ void post(T (&st)[S], const C&) {
    st[0] = wrap_angle(st[0]);
    st[1] = wrap_angle(st[1]);
  }
  __device__ static __forceinline__ void observe(const T (&st)[S], const C& c, T (&ob)[O]) {
    const int F[4] = {0, 1, 2, 3};
    normalize_fields<4>(c, F, st, ob);
  }
};

// ---- FluidTank: fluid_tank_env.py:97-106,146 ; P = (base_area, orifice_area, c_d, g) ---------
template <typename T> struct FluidTank {
  static constexpr int ID = EXCENV_FLUID_TANK, S = 1, A = 1, O = 1, P = 4, NY = 1, ND = 1;
  static constexpr bool IS_PMSM = false;
  static constexpr bool HAS_LUT = false;
  using C = Ctx<T, FluidTank>;
  __device__ static __forceinline__ void prep(C& c) { c.den[0].init(c.P[0], c.fastdiv); }  // base_area
  __device__ static __forceinline__ void get_y(const T (&st)[S], T (&y)[NY]) { y[0] = st[0]; }
  __device__ static __forceinline__ void set_y(T (&st)[S], const T (&y)[NY]) { st[0] = y[0]; }
  __device__ static __forceinline__ void f(const T (&y)[NY], const T (&u)[A], const C& c, const T (&)[S], T (&dy)[NY]) {
    const T base_area = c.P[0], orifice_area = c.P[1], c_d = c.P[2], g = c.P[3];
    const T h = max_nan(y[0], T(0));
    dy[0] = c.den[0].div(u[0], c.fastdiv) - c_d * orifice_area / base_area * xsqrt(T(2) * g * h);
  }
  __device__ static __forceinline__ void post(T (&st)[S], const C&) { st[0] = max_nan(st[0], T(0)); }
  __device__ static __forceinline__ void observe(const T (&st)[S], const C& c, T (&ob)[O]) {
    ob[0] = normalize_field(c, 0, st[0]);
  }
};

// ---- PMSM (linear dq-frame model): pmsm_env.py:509-523 ; P = (p,r_s,l_d,l_q,psi_p,u_dc,deadtime)
//      state = (u_d_buffer,u_q_buffer,epsilon,i_d,i_q,torque,omega_el) ; y = (i_d,i_q,eps)
template <typename T> struct Pmsm {
  static constexpr int ID = EXCENV_PMSM, S = 7, A = 2, O = 8, P = 7, NY = 3, ND = 2;
  static constexpr bool IS_PMSM = true;
  static constexpr bool HAS_LUT = false;
  using C = Ctx<T, Pmsm>;
  __device__ static __forceinline__ void prep(C& c) {
    c.den[0].init(c.P[2], c.fastdiv);  // l_d
    c.den[1].init(c.P[3], c.fastdiv);  // l_q
  }
  __device__ static __forceinline__ void get_y(const T (&st)[S], T (&y)[NY]) { y[0] = st[3]; y[1] = st[4]; y[2] = st[2]; }
  __device__ static __forceinline__ void set_y(T (&st)[S], const T (&y)[NY]) { st[3] = y[0]; st[4] = y[1]; st[2] = y[2]; }
  __device__ static __forceinline__ void f(const T (&y)[NY], const T (&u)[A], const C& c, const T (&st)[S], T (&dy)[NY]) {
    const T r_s = c.P[1], l_d = c.P[2], l_q = c.P[3], psi_p = c.P[4];
    const T omega_el = st[6];
    const InvDiv<T>* d[2] = {&c.den[0], &c.den[1]};  // / l_d, / l_q
    const T num[2] = {u[0] + omega_el * l_q * y[1] - r_s * y[0], u[1] - omega_el * (l_d * y[0] + psi_p) - r_s * y[1]};
    T q[2];
    div_all<2, T>(d, num, q, c.fastdiv);
    dy[0] = q[0];
    dy[1] = q[1];
    dy[2] = omega_el;
  }
  // pmsm_env.py:365-375
  __device__ static __forceinline__ T torque(T i_d, T i_q, const C& c) {
    const T p = c.P[0], l_d = c.P[2], l_q = c.P[3], psi_p = c.P[4];
    return T(1.5) * p * (psi_p + (l_d - l_q) * i_d) * i_q;
  }
  // pmsm_env.py:571-578 (wrap eps, derive torque from the new currents)
  __device__ static __forceinline__ void post(T (&st)[S], const C& c) {
    st[2] = wrap_angle(st[2]);
    st[5] = torque(st[3], st[4], c);
  }
  // pmsm_env.py:898-919: [i_d, i_q, omega_el, torque, cos eps, sin eps, u_d_buffer, u_q_buffer]
  template <class CC> __device__ static __forceinline__ void observe(const T (&st)[S], const CC& c, T (&ob)[O]) {
    T sn, cs;
    sincos_t(st[2], sn, cs);
    const int F[6] = {3, 4, 6, 5, 0, 1};
    const T x[6] = {st[3], st[4], st[6], st[5], st[0], st[1]};
    T n[6];
    normalize_fields<6>(c, F, x, n);
    ob[0] = n[0];
    ob[1] = n[1];
    ob[2] = n[2];
    ob[3] = n[3];
    ob[4] = cs;
    ob[5] = sn;
    ob[6] = n[4];
    ob[7] = n[5];
  }

  // pmsm_env.py:92-102 apply_hex_constraint. The sector bits idx_k = [sin(angle(c) - 2*pi*k/3) >= 0] are taken
  // algebraically (sin(phi - t) * |c| = beta*cos t - alpha*sin t): no atan2 / sin. The clip is continuous across
  // sector seams, so a different pick exactly on a seam changes the result by rounding only. ROTATION_MAP is
  // complex64 in the reference (:37-43): sqrt(3)/2 is float32-rounded in every working dtype.
  __device__ static __forceinline__ void hex_clip(T& alpha, T& beta) {
    const T re = alpha, im = beta;
    const T t = T(1.7320508075688772) * re;
    const bool i0 = im >= T(0);
    const bool i1 = (-im - t) >= T(0);
    const bool i2 = (t - im) >= T(0);
    const T q = T(0.8660254037844386f);
    // ROTATION_MAP[code], code = 4 i0 + 2 i1 + i2 (pmsm_env.py:37-43): (1, 0) for codes 0, 3', 4, 7; (1/2, q) for 5; (1/2, -q) for 6;
    // (-1/2, -q) for 2; (-1/2, q) for 1; (-1, 0) for 3. As arithmetic on the three bits (exact: only 0, 1/2, 1 and q occur), round 5 —
    // the table walk was eight integer compares and five selects per call, each select behind an `s_nop` (VALU-written SGPR):
    //   im(rot) = q * (i2 - i1);  |re(rot)| = 1 - |i2 - i1| / 2;  re(rot) < 0  iff  not i0 and (i1 or i2)
    const T b0 = i0 ? T(1) : T(0), b1 = i1 ? T(1) : T(0), b2 = i2 ? T(1) : T(0);
    const T d = b2 - b1;
    const T ri = q * d;
    const T neg = (T(1) - b0) * ((b1 > b2) ? b1 : b2);
    const T rr = (T(1) - T(0.5f) * xabs(d)) * (T(1) - T(2) * neg);
    T tr = re * rr - im * ri;
    T ti = re * ri + im * rr;
    const T lim_re = T(2.0 / 3.0);
    const T lim_im = T(2.0 / 3.0) * xsqrt(T(3));
    tr = min_nan(max_nan(tr, -lim_re), lim_re);
    ti = min_nan(max_nan(ti, T(0)), lim_im);
    alpha = tr * rr - ti * (-ri);
    beta = tr * (-ri) + ti * rr;
  }

  // pmsm_env.py:594-616 constraint_denormalization with the angle `eps` as given
  template <class CC>
  __device__ static __forceinline__ void constraint(const T (&a)[A], T eps, T omega_el, const CC& c, T (&uc)[2]) {
    const T u_dc = c.P[5];
    const T half_dc = u_dc / T(2);
    const T u_d = denormalize(a[0], c.amin[0], c.amax[0]);
    const T u_q = denormalize(a[1], c.amin[1], c.amax[1]);
    const T sc = T(1) / half_dc;
    const T n_d = u_d * sc, n_q = u_q * sc;
    T adv = eps + c.adv_coef * omega_el;  // step_eps(eps, deadtime+0.5, tau, omega_el) (:82-89)
    adv = pymod_two_pi(adv);
    adv = adv + ((adv > K<T>::pi) ? T(-2) * K<T>::pi : T(0));
    T sn, cs;
    sincos_t(adv, sn, cs);
    const T sm = -sn;  // T(-adv) = [[cos, sin(-adv)], [-sin(-adv), cos]]
    T al = cs * n_d + sm * n_q;
    T be = (-sm) * n_d + cs * n_q;
    hex_clip(al, be);
    const T o_d = cs * al + sn * be;  // T(adv)
    const T o_q = (-sn) * al + cs * be;
    uc[0] = o_d * half_dc;
    uc[1] = o_q * half_dc;
  }
};

// ---- PMSM saturated model: nonlinear_ode (pmsm_env.py:487-507), currents_to_torque_saturated (:377-381) -----------
// Everything else (action path, dead time, observation) is the linear model's.
template <typename T> struct PmsmSat {
  static constexpr int ID = EXCENV_PMSM, S = 7, A = 2, O = 8, P = 7, NY = 3, ND = 0;
  static constexpr bool IS_PMSM = true;
  static constexpr bool HAS_LUT = true;
  using C = Ctx<T, PmsmSat>;
  __device__ static __forceinline__ void prep(C&) {}
  using L = Pmsm<T>;
  __device__ static __forceinline__ void get_y(const T (&st)[S], T (&y)[NY]) { y[0] = st[3]; y[1] = st[4]; y[2] = st[2]; }
  __device__ static __forceinline__ void set_y(T (&st)[S], const T (&y)[NY]) { st[3] = y[0]; st[4] = y[1]; st[2] = y[2]; }

  // jax.scipy.interpolate.RegularGridInterpolator._find_indices: i = clip(searchsorted(g, x) - 1, 0, n - 2),
  // t = (x - g[i]) / (g[i+1] - g[i])  (not clipped: linear extrapolation; the padded edge makes it constant).
  // GP is a pointer to the grid in global memory or (address_space(3)) in LDS.
  // g0 / sc: the grid's first node and (n - 1) / (last - first), trajectory invariants kept in the Ctx. rc: refined
  // reciprocals of the cell widths (devmath.hpp InvDiv; staged next to the grids in LDS) or a null pointer -> plain division;
  // either way t has the bits of (x - lo) / (hi - lo).
  template <class GP>
  __device__ static __forceinline__ void find(GP g, int n, T x, T g0, T sc, int& i, T& lo, T& hi) {
    // The reference's grids are np.linspace: guess the cell arithmetically, then walk to the cell the stored grid values
    // define (same result as searchsorted(g, x, side="left") - 1 clipped to [0, n-2]). On a uniform grid the guess is off by
    // at most one, so each walk runs 0 or 1 times; on any other monotonic grid the walks still end in the right cell.
    T gf = (x - g0) * sc;
    gf = (gf > T(0)) ? gf : T(0);  // also maps NaN to 0
    gf = (gf < T(n - 2)) ? gf : T(n - 2);
    int i0 = (int)gf;
    lo = g[i0];
    hi = g[i0 + 1];
    while (!(lo < x) && i0 > 0 && x == x) {  // x <= g[i0]: the cell is further down
      --i0;
      hi = lo;
      lo = g[i0];
    }
    while (hi < x && i0 < n - 2) {  // x > g[i0 + 1]: further up
      ++i0;
      lo = hi;
      hi = g[i0 + 1];
    }
    i = i0;
  }
  // six bilinear look-ups sharing one cell: _evaluate_linear's corner order (i,j), (i,j+1), (i+1,j), (i+1,j+1)
  // rd / rq: refined reciprocals of the cell widths (devmath.hpp InvDiv; staged next to the grids in LDS) when `fast`;
  // either way the weights have the bits of (x - lo) / (hi - lo).
  template <class GP>
  __device__ static __forceinline__ void lookup_at(GP gd, GP gq, GP tab, GP rd, GP rq, bool fast, const C& c, T i_d, T i_q, T (&q)[6]) {
    const int nd = c.lut_nd, nq = c.lut_nq;
    int ix, iy;
    T lox, hix, loy, hiy, tx, ty;
    find(gd, nd, i_d, c.lut_g0[0], c.lut_sc[0], ix, lox, hix);
    find(gq, nq, i_q, c.lut_g0[1], c.lut_sc[1], iy, loy, hiy);
    if (fast) {
      InvDiv<T> wx, wy;
      wx.b = hix - lox;
      wx.y = rd[ix];
      wy.b = hiy - loy;
      wy.y = rq[iy];
      const InvDiv<T>* w[2] = {&wx, &wy};
      const T num[2] = {i_d - lox, i_q - loy};
      T t[2];
      div_all<2, T>(w, num, t, true);
      tx = t[0];
      ty = t[1];
    } else {
      tx = (i_d - lox) / (hix - lox);
      ty = (i_q - loy) / (hiy - loy);
    }
    const T w00 = (T(1) - tx) * (T(1) - ty), w01 = (T(1) - tx) * ty, w10 = tx * (T(1) - ty), w11 = tx * ty;
    GP n00 = tab + (ix * nq + iy) * 8;
    GP n10 = n00 + nq * 8;
#pragma unroll
    for (int k = 0; k < 6; ++k) q[k] = T(0) + n00[k] * w00 + n00[8 + k] * w01 + n10[k] * w10 + n10[8 + k] * w11;
  }
  __device__ static __forceinline__ void lookup(T i_d, T i_q, const C& c, T (&q)[6]) {
    if (c.lut_lds) {  // tables staged in LDS by stage_lut(): statically LDS-typed pointers -> ds_read gathers
      typedef const __attribute__((address_space(3))) T* LP;
      extern __shared__ __align__(16) unsigned char excenv_smem[];
      LP sm = (LP)excenv_smem;
      const int ntab = c.lut_nd * c.lut_nq * 8;
      LP gd = sm + ntab, gq = gd + c.lut_nd, rd = gq + c.lut_nq, rq = rd + c.lut_nd;
      lookup_at<LP>(gd, gq, sm, rd, rq, c.fastdiv, c, i_d, i_q, q);
    } else {
      lookup_at<const T*>(c.lut_gd, c.lut_gq, c.lut_tab, nullptr, nullptr, false, c, i_d, i_q, q);
    }
  }
  // the vector field with the six interpolated values of its point already at hand (the trajectory kernels look a point up
  // once and use it for the torque of the saved row and for the first stage of the step that starts there)
  __device__ static __forceinline__ void f_q(const T (&y)[NY], const T (&u)[A], const C& c, const T (&st)[S], T (&dy)[NY],
                                             const T (&q)[6]) {
    const T r_s = c.P[1], omega_el = st[6];
    const T L_dd = q[0], L_dq = q[1], L_qd = q[2], L_qq = q[3], Psi_d = q[4], Psi_q = q[5];
    // 2x2 inverse in closed form (the reference calls jnp.linalg.inv: LU; parity of this model is unpinned anyway)
    const T det = L_dd * L_qq - L_dq * L_qd;
    InvDiv<T> by_det;  // four divisions by the same determinant: one reciprocal refinement (same bits as `/`, devmath.hpp)
    by_det.init(det, c.fastdiv);
    const InvDiv<T>* dd[4] = {&by_det, &by_det, &by_det, &by_det};
    const T num[4] = {L_qq, -L_dq, -L_qd, L_dd};
    T a[4];
    div_all<4, T>(dd, num, a, c.fastdiv);
    const T a00 = a[0], a01 = a[1], a10 = a[2], a11 = a[3];
    const T j0 = -Psi_q, j1 = Psi_d;  // J_k @ psi_dq, J_k = [[0,-1],[1,0]]
    const T d1_0 = (-a00 * r_s) * y[0] + (-a01 * r_s) * y[1];
    const T d1_1 = (-a10 * r_s) * y[0] + (-a11 * r_s) * y[1];
    const T d2_0 = a00 * u[0] + a01 * u[1];
    const T d2_1 = a10 * u[0] + a11 * u[1];
    const T d3_0 = ((-a00) * j0 + (-a01) * j1) * omega_el;
    const T d3_1 = ((-a10) * j0 + (-a11) * j1) * omega_el;
    dy[0] = d1_0 + d2_0 + d3_0;
    dy[1] = d1_1 + d2_1 + d3_1;
    dy[2] = omega_el;
  }
  __device__ static __forceinline__ void f(const T (&y)[NY], const T (&u)[A], const C& c, const T (&st)[S], T (&dy)[NY]) {
    T q[6];
    lookup(y[0], y[1], c, q);
    f_q(y, u, c, st, dy, q);
  }
  __device__ static __forceinline__ T torque_q(T i_d, T i_q, const C& c, const T (&q)[6]) {
    return T(1.5) * c.P[0] * (q[4] * i_q - q[5] * i_d);
  }
  __device__ static __forceinline__ T torque(T i_d, T i_q, const C& c) {
    T q[6];
    lookup(i_d, i_q, c, q);
    return torque_q(i_d, i_q, c, q);
  }
  __device__ static __forceinline__ void post(T (&st)[S], const C& c) {
    st[2] = wrap_angle(st[2]);
    st[5] = torque(st[3], st[4], c);
  }
  // post() that also hands the look-up of the new operating point to the caller
  __device__ static __forceinline__ void post_q(T (&st)[S], const C& c, T (&q)[6]) {
    st[2] = wrap_angle(st[2]);
    lookup(st[3], st[4], c, q);
    st[5] = torque_q(st[3], st[4], c, q);
  }
  __device__ static __forceinline__ void observe(const T (&st)[S], const C& c, T (&ob)[O]) { L::observe(st, c, ob); }
  __device__ static __forceinline__ void constraint(const T (&a)[A], T eps, T omega_el, const C& c, T (&uc)[2]) {
    L::constraint(a, eps, omega_el, c, uc);
  }
};

// generate_observation without a branch: the values of M::observe wherever `bad` stays false (same operations, same order — the
// fast paths of sincos_t and InvDiv::div), garbage where it turns true; the caller redoes such a block with M::observe.
// Available (returns true at compile time) for the fast-division trajectory kernels, PMSM in fp32 only (fp64 sin / cos are the
// device library's).
template <class M, typename T> constexpr bool observe_defer_ok() { return !(M::IS_PMSM && sizeof(T) == 8) && !M::HAS_LUT; }
template <class M, typename T>
__device__ __forceinline__ void observe_defer(const T (&st)[M::S], const Ctx<T, M>& c, T (&ob)[M::O], bool& bad) {
  if constexpr (M::IS_PMSM) {  // Pmsm::observe
    if constexpr (sizeof(T) == 4) {
      T sn, cs;
      sincos_defer(st[2], sn, cs, bad);
      const InvDiv<T>* d[6] = {&c.nrm[3], &c.nrm[4], &c.nrm[6], &c.nrm[5], &c.nrm[0], &c.nrm[1]};
      const T num[6] = {T(2) * (st[3] - c.smin[3]), T(2) * (st[4] - c.smin[4]), T(2) * (st[6] - c.smin[6]),
                        T(2) * (st[5] - c.smin[5]), T(2) * (st[0] - c.smin[0]), T(2) * (st[1] - c.smin[1])};
      T n[6];
      div_all_defer<6, T>(d, num, n, bad);
      ob[0] = n[0] - T(1);
      ob[1] = n[1] - T(1);
      ob[2] = n[2] - T(1);
      ob[3] = n[3] - T(1);
      ob[4] = cs;
      ob[5] = sn;
      ob[6] = n[4] - T(1);
      ob[7] = n[5] - T(1);
    }
  } else {  // the other five: every state field normalised, in order (normalize_fields / normalize_field)
    static_assert(M::IS_PMSM || M::O == M::S, "observation = the normalised state");
    const InvDiv<T>* d[M::S];
    T num[M::S], n[M::S];
#pragma unroll
    for (int j = 0; j < M::S; ++j) {
      d[j] = &c.nrm[j];
      num[j] = T(2) * (st[j] - c.smin[j]);
    }
    div_all_defer<M::S, T>(d, num, n, bad);
#pragma unroll
    for (int j = 0; j < M::S; ++j) ob[j] = n[j] - T(1);
  }
}

// ---- reward / truncated / terminated (reference generate_reward / generate_truncated / generate_terminated) -----
// What GymWrapper.gym_step evaluates after every vmap_step (gym_wrapper.py:117-126). `ref[j]` is the physical
// reference value of state field idx[j] (control_state order).
template <class M> __host__ __device__ constexpr bool is_angle_field(int f) {
  return (M::ID == EXCENV_PENDULUM && f == 0) || (M::ID == EXCENV_CART_POLE && f == 2) ||
         (M::ID == EXCENV_ACROBOT && (f == 0 || f == 1));
}

template <class M, typename T>
__device__ __forceinline__ void pick_field(const T (&st)[M::S], const Ctx<T, M>& c, int f, T& x, T& lo, T& hi) {
  x = st[0]; lo = c.smin[0]; hi = c.smax[0];
#pragma unroll
  for (int q = 1; q < M::S; ++q) {
    x = (f == q) ? st[q] : x;
    lo = (f == q) ? c.smin[q] : lo;
    hi = (f == q) ? c.smax[q] : hi;
  }
}

// e.g. pendulum_env.py:297-309, mass_spring_damper_env.py:296-302, pmsm_env.py:985-1037. The control loop is unrolled
// over EXCENV_MAX_CONTROL with a uniform guard so that `ref` (registers) is only indexed statically.
// PMSM reward (pmsm_env.py:985-1037) from the references of the controlled fields among i_d, i_q, torque: one function for the
// general instantiation (which finds them by scanning control_idx) and the lean gym instantiation (which holds them per
// environment of a lane) — same operations, same order, same bits.
template <class M, typename T>
__device__ __forceinline__ T pmsm_reward(const T (&st)[M::S], const Ctx<T, M>& c, bool has_id, T r_id, bool has_iq, T r_iq,
                                         bool has_tq, T r_tq) {
  T reward = T(0);
  const T i_d = normalize(st[3], c.smin[3], c.smax[3]);
  const T i_q = normalize(st[4], c.smin[4], c.smax[4]);
  if (has_id && has_iq) {  // current_reward_func (pmsm_env.py:1010-1012), gamma = 0.85
    const T dd = i_d - normalize(r_id, c.smin[3], c.smax[3]);
    const T dq = i_q - normalize(r_iq, c.smin[4], c.smax[4]);
    const T mse = T(0.5) * (dd * dd) + T(0.5) * (dq * dq);
    reward = reward + T(-1) * (mse * T(1 - 0.85));
  }
  if (has_tq) {  // torque_reward_func(i_d, i_q, torque, torque_ref, 1, 0.85) (pmsm_env.py:1014-1037)
    const T tq = normalize(st[5], c.smin[5], c.smax[5]);
    const T tr = normalize(r_tq, c.smin[5], c.smax[5]);
    const T i_s = xsqrt(i_d * i_d + i_q * i_q);
    const T i_n = T(1), i_d_plus = T(0.2) * i_n, tol = T(0.01);
    T rew = T(0);
    rew = (i_s > T(1)) ? T(-1) * xabs(i_s) : rew;
    rew = ((i_s < T(1)) && (i_s > i_n)) ? T(0.5) * (T(1) - (i_s - i_n) / (T(1) - i_n)) - T(1) : rew;
    rew = ((i_s < i_n) && (i_d > i_d_plus)) ? T(-0.5) * ((i_d - i_d_plus) / (i_n - i_d_plus)) : rew;
    const T ad = xabs(tq - tr);
    rew = ((i_s < i_n) && (i_d < i_d_plus) && (ad > tol)) ? T(0.5) * (T(1) - xabs((tr - tq) / T(2))) : rew;
    rew = ((i_s < i_n) && (i_d < i_d_plus) && (ad < tol)) ? T(1) - T(0.5) * i_s : rew;
    reward = reward + rew * T(1 - 0.85);
  }
  return reward;
}

template <class M, typename T>
__device__ __forceinline__ T env_reward(const T (&st)[M::S], const Ctx<T, M>& c, int n_control, const int* idx,
                                        const T (&ref)[EXCENV_MAX_CONTROL]) {
  T reward = T(0);
  if constexpr (M::IS_PMSM) {
    // control_state membership: "i_d" (3), "i_q" (4), "torque" (5)
    T r_id = T(0), r_iq = T(0), r_tq = T(0);
    bool has_id = false, has_iq = false, has_tq = false;
#pragma unroll
    for (int j = 0; j < EXCENV_MAX_CONTROL; ++j) {
      if (j < n_control) {
        const int f = idx[j];
        if (f == 3) { has_id = true; r_id = ref[j]; }
        if (f == 4) { has_iq = true; r_iq = ref[j]; }
        if (f == 5) { has_tq = true; r_tq = ref[j]; }
      }
    }
    reward = pmsm_reward<M, T>(st, c, has_id, r_id, has_iq, r_iq, has_tq, r_tq);
  } else {
#pragma unroll
    for (int j = 0; j < EXCENV_MAX_CONTROL; ++j) {
      if (j < n_control) {
        const int f = idx[j];
        T x, lo, hi;
        pick_field<M, T>(st, c, f, x, lo, hi);
        const T r = ref[j];
        bool ang = false;
#pragma unroll
        for (int q = 0; q < M::S; ++q) ang = ang || (is_angle_field<M>(q) && f == q);
        if (ang) {
          T sx, cx, sr, cr;
          sincos_t(x, sx, cx);
          sincos_t(r, sr, cr);
          const T ds = sx - sr, dc = cx - cr;
          reward = reward + -(ds * ds + dc * dc);
        } else {
          const T d = normalize(x, lo, hi) - normalize(r, lo, hi);
          reward = reward + -(d * d);
        }
      }
    }
  }
  return reward;
}

}  // namespace excenv
