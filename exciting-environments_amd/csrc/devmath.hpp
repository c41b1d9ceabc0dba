// Device math for the batched ODE kernels (gfx950). Built with -ffp-contract=off: every + - * / and
// sqrt below is a single IEEE-rounded operation in the order written, FMAs appear only where written
// explicitly, so the trig-free environments are bit-identical to the CPU oracle.
#pragma once
#include <hip/hip_runtime.h>

namespace excenv {

template <typename T> struct K;  // constants, rounded from the Python doubles the reference uses
template <> struct K<float> {
  static constexpr float pi = 3.14159265358979323846f;
  static constexpr float two_pi = 6.28318530717958647692f;
  static constexpr float half_pi = 1.57079632679489661923f;
  static constexpr float inv_two_pi = 0.15915494309189533577f;
};
template <> struct K<double> {
  static constexpr double pi = 3.14159265358979323846;
  static constexpr double two_pi = 6.28318530717958647692;
  static constexpr double half_pi = 1.57079632679489661923;
  static constexpr double inv_two_pi = 0.15915494309189533577;
};

__device__ __forceinline__ __attribute__((nodebug)) float xfma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ __attribute__((nodebug)) double xfma(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ __attribute__((nodebug)) float xtrunc(float a) { return __builtin_truncf(a); }
__device__ __forceinline__ __attribute__((nodebug)) double xtrunc(double a) { return __builtin_trunc(a); }
__device__ __forceinline__ __attribute__((nodebug)) float xabs(float a) { return __builtin_fabsf(a); }
__device__ __forceinline__ __attribute__((nodebug)) double xabs(double a) { return __builtin_fabs(a); }
__device__ __forceinline__ __attribute__((nodebug)) float xsqrt(float a) { return __builtin_sqrtf(a); }  // correctly rounded (hipcc default)
__device__ __forceinline__ __attribute__((nodebug)) double xsqrt(double a) { return __builtin_sqrt(a); }
// sqrt(s) > 1 without the square root (the PMSM flags |i_dq| > 1, pmsm_env.py:972-983, per environment and saved row). With a
// correctly rounded square root the two are the same predicate for EVERY s: sqrt(s) rounds above 1 iff sqrt(s) > 1 + ulp/2 (the tie
// goes to the even neighbour, 1), i.e. s > 1 + ulp + ulp^2/4, i.e. s >= 1 + 2 ulp — as a value of the type: s > nextafter(1). NaN: false
// both ways; inf: true both ways. Checked exhaustively over the 2^21 values around 1 in both precisions (tests/test_oracle_golden.py).
__device__ __forceinline__ bool sqrt_exceeds_one(float s) { return s > 1.00000011920928955078125f; }   // 1 + 2^-23
__device__ __forceinline__ bool sqrt_exceeds_one(double s) { return s > 1.0000000000000002220446049250313; }  // 1 + 2^-52
// (the library's bit-serial remainder loop: out of line, see sincos_lib below — taken only for |x| / 2 pi >= 2^22, NaN, inf)
__device__ __attribute__((noinline)) float xfmod_slow(float a, float b) { return fmodf(a, b); }
__device__ __attribute__((noinline)) double xfmod_slow(double a, double b) { return fmod(a, b); }

// jnp.maximum / jnp.minimum against a bound: NaN in x propagates (jnp.clip semantics).
template <typename T> __device__ __forceinline__ __attribute__((nodebug)) T max_nan(T x, T lo) { return (x < lo) ? lo : x; }
template <typename T> __device__ __forceinline__ __attribute__((nodebug)) T min_nan(T x, T hi) { return (x > hi) ? hi : x; }
// jnp.sign
template <typename T> __device__ __forceinline__ T sign_of(T x) { return (x > T(0)) ? T(1) : ((x < T(0)) ? T(-1) : x); }

// utils.py:13-17 — exactly this operation order
template <typename T> __device__ __forceinline__ T normalize(T x, T lo, T hi) { return T(2) * (x - lo) / (hi - lo) - T(1); }
template <typename T> __device__ __forceinline__ T denormalize(T x, T lo, T hi) { return (x + T(1)) / T(2) * (hi - lo) + lo; }

// ---- division by a loop-invariant denominator, bit-identical to `a / b` -----------------------------------------------
// hipcc expands an IEEE-rounded `a / b` into v_div_scale x2, v_rcp, a Newton refinement of the reciprocal of the (scaled)
// denominator, q = a*y, one (fp64) or two (fp32) residual corrections folded into v_div_fmas, and v_div_fixup: 11-12
// instructions, 44 % of the VALU work of MassSpringDamper Tsit5 fp64. Everything that depends only on b is loop-invariant
// on this path ((hi - lo) of a normalisation, m, l_d, l_q, ...). InvDiv keeps that part — the SAME refinement sequence the
// compiler emits (AMDGPU LowerFDIV32 / LowerFDIV64) — and repeats per division only the same multiply and residual FMAs on
// the unscaled operands. v_div_scale leaves its operands untouched unless an exponent is extreme, so for moderate b and a
// moderate quotient the result has the bits of the compiler's sequence; everything else (zero, inf, NaN, denormal, huge or
// tiny operands, or a denominator outside the moderate range: y is NaN then) fails the range test on the quotient and takes
// the plain `a / b`. Checked bit for bit on the GPU over random and edge-case operands (test_invariant_division_*).
// The range test, round 5. It used to be two float compares per quotient (v_cmp into SGPR pairs, s_or, and an `s_nop` per
// compare -> select hazard): for PMSM Tsit5 257 scalar instructions + 149 s_nops per wave-step, more than the divisions' own
// arithmetic (tools/isa_tally.py). Now every quotient contributes its EXPONENT's distance from the low end of the window to an
// unsigned running maximum (v_and, v_sub, v_max_u32: no scalar register involved; an exponent below the window wraps to a huge
// value, NaN / inf carry the largest exponent) and a group of divisions is tested with ONE compare. The window in exponents,
// [2^-60, 2^60) for fp32 and [2^-400, 2^400) for fp64, lies inside the old value window, so wherever the fast path is taken it is
// taken under the same guarantee (the quotient has the bits of the compiler's division); zero still goes to the plain division.
template <typename T> struct InvDivLimits;
template <> struct InvDivLimits<float> {
  static constexpr float b_lo = 0x1p-40f, b_hi = 0x1p40f;
  static constexpr uint32_t e_mask = 0x7F800000u, e_lo = (127u - 60u) << 23, e_span = (120u << 23) - 1u;  // exponents -60 .. +59
};
template <> struct InvDivLimits<double> {
  static constexpr double b_lo = 0x1p-300, b_hi = 0x1p300;
  static constexpr uint32_t e_mask = 0x7FF00000u, e_lo = (1023u - 400u) << 20, e_span = (800u << 20) - 1u;  // exponents -400 .. +399, high word
};
__device__ __forceinline__ __attribute__((nodebug)) uint32_t exponent_word(float q) { return __float_as_uint(q); }
__device__ __forceinline__ __attribute__((nodebug)) uint32_t exponent_word(double q) { return (uint32_t)__double2hiint(q); }

template <typename T> struct InvDiv {
  T b, y;  // denominator; refined reciprocal (NaN when b is outside the moderate range -> every division takes `a / b`)
  __device__ __forceinline__ void init(T b_, bool fast) {
    b = b_;
    y = T(0);
    if (!fast) return;
    T r;
    if constexpr (sizeof(T) == 4) {
      const T y0 = __builtin_amdgcn_rcpf(b);
      const T e0 = xfma(-b, y0, T(1));
      r = xfma(e0, y0, y0);
    } else {
      const T y0 = __builtin_amdgcn_rcp(b);
      const T e0 = xfma(-b, y0, T(1));
      const T y1 = xfma(y0, e0, y0);
      const T e1 = xfma(-b, y1, T(1));
      r = xfma(y1, e1, y1);
    }
    const T ab = xabs(b);
    y = (ab >= InvDivLimits<T>::b_lo && ab <= InvDivLimits<T>::b_hi) ? r : __builtin_nan("");
  }
  // The fast quotient alone; its exponent's distance from the window goes into the group's running maximum `acc` (start at 0,
  // test with out_of_window()): lets a caller issue several independent divisions as straight-line code and test once.
  __device__ __forceinline__ T fastq(T a, uint32_t& acc) const {
    T q = a * y;
    if constexpr (sizeof(T) == 4) {
      const T r0 = xfma(-b, q, a);
      q = xfma(r0, y, q);
    }
    const T r1 = xfma(-b, q, a);
    q = xfma(r1, y, q);
    const uint32_t d = (exponent_word(q) & InvDivLimits<T>::e_mask) - InvDivLimits<T>::e_lo;
    acc = d > acc ? d : acc;
    return q;
  }
  static __device__ __forceinline__ bool out_of_window(uint32_t acc) { return acc > InvDivLimits<T>::e_span; }
  // `fast` is a per-kernel compile-time constant carried in the Ctx (false on the one-step-per-launch path, where the
  // reciprocal would be set up and used once: plain division is cheaper there)
  __device__ __forceinline__ T div(T a, bool fast) const {
    if (!fast) return a / b;
    uint32_t acc = 0;
    T q = fastq(a, acc);
    const bool slow = out_of_window(acc);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(slow) != 0, 0)) {  // wave-uniform: skipped entirely in the common case
      const T exact = a / b;
      q = slow ? exact : q;
    }
    return q;
  }
};

// N independent divisions num[j] / d[j]->b with ONE wave-uniform test: out[j] has the bits of the plain division.
template <int N, typename T>
__device__ __forceinline__ void div_all(const InvDiv<T>* const (&d)[N], const T (&num)[N], T (&out)[N], bool fast) {
  if (!fast) {
#pragma unroll
    for (int j = 0; j < N; ++j) out[j] = num[j] / d[j]->b;
    return;
  }
  uint32_t acc = 0;
#pragma unroll
  for (int j = 0; j < N; ++j) out[j] = d[j]->fastq(num[j], acc);
  if (__builtin_expect(__builtin_amdgcn_ballot_w64(InvDiv<T>::out_of_window(acc)) != 0, 0)) {
#pragma unroll
    for (int j = 0; j < N; ++j) out[j] = num[j] / d[j]->b;  // every lane: same bits as the fast path where that was valid
  }
}

// the same without the test: `bad` collects "some quotient needs the plain division" for the caller's one test per block
template <int N, typename T>
__device__ __forceinline__ void div_all_defer(const InvDiv<T>* const (&d)[N], const T (&num)[N], T (&out)[N], bool& bad) {
  uint32_t acc = 0;
#pragma unroll
  for (int j = 0; j < N; ++j) out[j] = d[j]->fastq(num[j], acc);
  bad = bad || InvDiv<T>::out_of_window(acc);
}

// Exact C fmod(|x|, Y) for the compile-time divisor Y = 2*pi, without the library's bit-serial loop:
// the truncated quotient is estimated with a reciprocal multiply (off by at most one for |q| < 2^22),
// the remainder |x| - q*Y is then a single exactly-representable fma, re-derived if the estimate was off.
// Falls back to the library for huge quotients. Then the jnp.remainder sign fix (result takes the
// divisor's sign): lax.rem + select(add) — reference semantics of `%` in e.g. pendulum_env.py:188.
// Round 5: straight-line. The three-way fix-up of the estimate used to be nested per-lane branches (s_and_saveexec / s_xor /
// s_cbranch_execz per level: ~10 scalar instructions per call next to 8 vector ones, in kernels whose waves are bound by how
// many instructions of ANY kind they issue); now the adjusted quotient is selected and the remainder re-derived unconditionally
// (fma(-q, Y, |x|) with the unadjusted q gives the first remainder's bits again), and the one remaining guard — a quotient too
// large for the estimate, NaN, inf — is wave-uniform. Same values as before, bit for bit (test_device_pymod_*).
template <typename T> __device__ __forceinline__ T pymod_two_pi(T x) {
  const T y = K<T>::two_pi;
  const T ax = xabs(x);
  const T q = xtrunc(ax * K<T>::inv_two_pi);
  const bool big = !(q < T(4194304.0));  // also NaN / inf
  const T r0 = xfma(-q, y, ax);
  const T qa = (r0 < T(0)) ? q - T(1) : ((r0 >= y) ? q + T(1) : q);
  T r = xfma(-qa, y, ax);
  if (__builtin_expect(__builtin_amdgcn_ballot_w64(big) != 0, 0)) {
    const T lib = xfmod_slow(ax, y);
    r = big ? lib : r;
  }
  r = (x < T(0)) ? -r : r;       // fmod carries the dividend's sign
  r = (r < T(0)) ? r + y : r;    // Python-style: shift negatives by the (positive) divisor
  return r;
}

// ((theta + pi) % (2*pi)) - pi
template <typename T> __device__ __forceinline__ T wrap_angle(T th) { return pymod_two_pi(th + K<T>::pi) - K<T>::pi; }

// ---- sin / cos ---------------------------------------------------------------------------
// fp64: device library. fp32: a 3-constant Cody-Waite reduction by pi/2 (pi/2 split so that n * 1.5703125 is exact for
// n < 2^16) and the Cephes single-precision minimax kernels, ~25 VALU ops for the pair. |x| <= pi (the wrapped angles of
// the step path): <= 1.5 ulp. Up to |x| <= 65536 (raw sim_ahead angles are not wrapped between steps and grow with the
// trajectory) the ABSOLUTE error stays <= 9.4e-8 (measured over 5e6 points per decade, incl. the neighbours of every
// multiple of pi/2), i.e. <= 1.5 ulp of a value near 1 and a few ulp of the argument near the zeros. Beyond that the
// device library's full-range routine (Payne-Hanek) takes over.
__device__ __forceinline__ void sincos_t(double x, double& s, double& c) { ::sincos(x, &s, &c); }

__device__ __forceinline__ void sincos_fast(float x, float& s, float& c);
// The device library's full-range routines (Payne-Hanek reduction: ~300 instructions for the pair) as a REAL function: once per
// code object instead of once per call site. Inlined, every sincos_t of a trajectory loop carried its own copy — the lean gym
// loops of cart-pole / acrobot (20 call sites per half loop) were 80 KB, 130 KB after the guards became wave-uniform; with the
// call they are 44 KB, register counts unchanged, no scratch (round 5, tools/loop_code_size.py). Never executed for |x| <= 65536.
#ifndef EXCENV_SLOW_PATHS_NOINLINE
#define EXCENV_SLOW_PATHS_NOINLINE 1
#endif
#if EXCENV_SLOW_PATHS_NOINLINE
#define EXCENV_SLOW_FN __device__ __attribute__((noinline))
#else
#define EXCENV_SLOW_FN __device__ __forceinline__
#endif
EXCENV_SLOW_FN float2 sincos_lib(float x) { return make_float2(::sinf(x), ::cosf(x)); }
__device__ __forceinline__ void sincos_t(float x, float& s, float& c) {
  sincos_fast(x, s, c);
  const bool big = !(xabs(x) <= 65536.0f);  // also NaN / inf
  if (__builtin_expect(__builtin_amdgcn_ballot_w64(big) != 0, 0)) {  // wave-uniform guard (a per-lane one costs an exec-mask dance per call)
    const float2 l = sincos_lib(x);
    s = big ? l.x : s;
    c = big ? l.y : c;
  }
}
// The fast path alone (valid for |x| <= 65536) + its guard OR-ed into `bad`: a caller that evaluates several independent rows as
// ONE straight-line block (the flush of kernels_emr.hpp) tests `bad` once per block and redoes the block with the guarded forms —
// the guards of sincos_t / InvDiv::div end basic blocks, and the chains of independent rows cannot interleave across them.
__device__ __forceinline__ void sincos_defer(float x, float& s, float& c, bool& bad) {
  bad = bad || !(xabs(x) <= 65536.0f);
  sincos_fast(x, s, c);
}
__device__ __forceinline__ void sincos_fast(float x, float& s, float& c) {
  float n = __builtin_rintf(x * 0.63661977236758134308f);  // x * 2/pi
  float r = xfma(-n, 1.5703125f, x);                        // pi/2 split in three (Cephes DP1..3 doubled)
  r = xfma(-n, 4.837512969970703125e-4f, r);
  r = xfma(-n, 7.54978995489188216e-8f, r);
  int q = (int)n;
  float z = r * r;
  // sin(r) on |r| <= pi/4
  float ps = xfma(-1.9515295891e-4f, z, 8.3321608736e-3f);
  ps = xfma(ps, z, -1.6666654611e-1f);
  float sr = xfma(ps * z, r, r);
  // cos(r) on |r| <= pi/4
  float pc = xfma(2.443315711809948e-5f, z, -1.388731625493765e-3f);
  pc = xfma(pc, z, 4.166664568298827e-2f);
  float cr = xfma(pc * z, z, xfma(-0.5f, z, 1.0f));
  float s0 = (q & 1) ? cr : sr;
  float c0 = (q & 1) ? sr : cr;
  s = (q & 2) ? -s0 : s0;
  c = ((q + 1) & 2) ? -c0 : c0;
}

__device__ __forceinline__ float sin_t(float x) { float s, c; sincos_t(x, s, c); return s; }
__device__ __forceinline__ float cos_t(float x) { float s, c; sincos_t(x, s, c); return c; }
__device__ __forceinline__ double sin_t(double x) { return ::sin(x); }
__device__ __forceinline__ double cos_t(double x) { return ::cos(x); }

}  // namespace excenv
