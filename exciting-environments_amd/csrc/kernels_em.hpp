// Fused env-major (reference row-major) trajectory kernel: actions [B][K][A] in, observations [B][N+1][OW] and state
// leaves [B][N+1] out, with NO transposition pass. Time is the contiguous axis of these arrays while the parallelism
// runs across environments, so each wave (64 environments, one lane each) stages TK solver steps in LDS and then
// writes every environment's TK-step run contiguously: 256-byte runs for PMSM observations (16 B per lane), TK-word
// runs for the state leaves. One wave per workgroup, so the two barriers per tile are wave-local.
#pragma once
#include "kernels.hpp"

namespace excenv {

constexpr int EM_TK = 8;      // solver steps staged per tile
constexpr int EM_LANES = 64;  // one wave per workgroup

template <typename T> __host__ __device__ constexpr size_t em_lds_elems(int A, int OW, int S, bool with_states) {
  return (size_t)EM_LANES * ((EM_TK + 1) * A + 1) + (size_t)EM_LANES * (EM_TK * OW + 1) +
         (with_states ? (size_t)S * EM_LANES * (EM_TK + 1) : 0);
}

template <class M, typename T, int SOLVER, bool AHEAD, bool BATCHED>
__global__ void __launch_bounds__(EM_LANES) sim_ahead_em_kernel(const SimArgs<T, M> ka) {
  constexpr int S = M::S, A = M::A, O = M::O, TK = EM_TK;
  extern __shared__ __align__(16) unsigned char excenv_em_smem[];
  const int OW = O + ka.n_control;
  const bool with_states = ka.straj[0] != nullptr;
  const int LDA = (TK + 1) * A + 1, LDO = TK * OW + 1, LDS_ = TK + 1;  // odd leading dimensions: conflict-free columns
  T* tact = reinterpret_cast<T*>(excenv_em_smem);
  T* tobs = tact + EM_LANES * LDA;
  T* tst = tobs + EM_LANES * LDO;

  const int lane = threadIdx.x;
  const int64_t b0 = (int64_t)blockIdx.x * EM_LANES;
  const int64_t i0 = b0 + lane;
  const bool active = i0 < ka.B;
  const int nenv = (int)((ka.B - b0 < EM_LANES) ? (ka.B - b0) : EM_LANES);  // envs of this workgroup
  Ctx<T, M> c;
  load_ctx<BATCHED>(c, ka.kp, active ? i0 : 0, ka.dt, ka.env_tau, ka.adv_coef);

  T st[S];
#pragma unroll
  for (int j = 0; j < S; ++j) st[j] = active ? ka.state_in[j][i0] : T(0);
  AheadAux<T> aux;
  if constexpr (AHEAD && M::IS_PMSM) {
    aux.eps0 = st[2];
    aux.buf0[0] = aux.prev_clip[0] = st[0];
    aux.buf0[1] = aux.prev_clip[1] = st[1];
  }
  const bool deadtime_on = M::IS_PMSM ? (c.P[M::P - 1] > T(0)) : false;
  T cref[EXCENV_MAX_CONTROL];  // normalised reference columns (constant along the trajectory); static indices only
#pragma unroll
  for (int j = 0; j < EXCENV_MAX_CONTROL; ++j) {
    cref[j] = T(0);
    if (j < ka.n_control) {
      const int f = ka.control_idx[j];
      T lo = c.smin[0], hi = c.smax[0];
#pragma unroll
      for (int q = 1; q < S; ++q) {
        lo = (f == q) ? c.smin[q] : lo;
        hi = (f == q) ? c.smax[q] : hi;
      }
      if (active) cref[j] = normalize(ka.reference[j][i0], lo, hi);
    }
  }

  const int64_t N = ka.K;  // substeps == 1 on this path (host)
  for (int64_t n0 = 0; n0 <= N; n0 += TK) {
    const int cnt = (int)((N + 1 - n0 < TK) ? (N + 1 - n0) : TK);               // rows n0 .. n0+cnt-1
    int na = (int)((ka.K - n0 < TK + 1) ? (ka.K - n0) : (TK + 1));               // actions n0 .. n0+na-1 (one ahead)
    na = na < 0 ? 0 : na;
    // ---- action tile: every env's na*A words are contiguous in [B][K][A] ----
    {
      const int per = na * A;
      const T* src = ka.actions + (b0 * ka.K + n0) * A;
      for (int idx = lane; idx < nenv * per; idx += EM_LANES) {
        const int e = idx / per, j = idx - e * per;
        tact[e * LDA + j] = src[(int64_t)e * ka.K * A + j];
      }
    }
    __syncthreads();
    // ---- TK solver steps, rows staged in LDS ----
    T sv[S];
    for (int t = 0; t < cnt; ++t) {
      const int64_t n = n0 + t;
#pragma unroll
      for (int j = 0; j < S; ++j) sv[j] = st[j];
      if constexpr (AHEAD) {
        M::post(sv, c);
        if constexpr (M::IS_PMSM) {
          if (deadtime_on) {
            sv[0] = (n == 0) ? aux.buf0[0] : aux.prev_clip[0];
            sv[1] = (n == 0) ? aux.buf0[1] : aux.prev_clip[1];
          } else {
            sv[0] = T(0);
            sv[1] = T(0);
          }
        }
      }
      T ob[O];
      M::observe(sv, c, ob);
#pragma unroll
      for (int q = 0; q < O; ++q) tobs[lane * LDO + t * OW + q] = ob[q];
#pragma unroll
      for (int j = 0; j < EXCENV_MAX_CONTROL; ++j)
        if (j < ka.n_control) tobs[lane * LDO + t * OW + O + j] = cref[j];
      if (with_states) {
#pragma unroll
        for (int j = 0; j < S; ++j) tst[(j * EM_LANES + lane) * LDS_ + t] = sv[j];
      }
      if (n < N) {
        T a_cur[A], a_nxt[A];
        const int t1 = (t + 1 < na) ? t + 1 : na - 1;
#pragma unroll
        for (int q = 0; q < A; ++q) {
          a_cur[q] = tact[lane * LDA + t * A + q];
          a_nxt[q] = tact[lane * LDA + t1 * A + q];
        }
        if constexpr (AHEAD) {
          const int64_t k1 = (n + 1 < ka.K) ? n + 1 : ka.K - 1;
          env_advance_raw<M, SOLVER>(st, a_cur, a_nxt, n, k1, c, aux);
        } else {
          env_step<M, SOLVER>(st, a_cur, c);
        }
      }
    }
    __syncthreads();
    // ---- flush: per-env contiguous runs ----
    {
      const int per = cnt * OW;
      T* dst = ka.obs + (b0 * (N + 1) + n0) * OW;
      constexpr int VW = 16 / (int)sizeof(T);  // elements per 16-byte piece
      if ((OW % VW) == 0) {  // every env row and tile start is then 16-byte aligned
        const int perv = per / VW;
        for (int idx = lane; idx < nenv * perv; idx += EM_LANES) {
          const int e = idx / perv, j = (idx - e * perv) * VW;
          T v[VW];
#pragma unroll
          for (int q = 0; q < VW; ++q) v[q] = tobs[e * LDO + j + q];
          store_v<T, VW>(dst + (int64_t)e * (N + 1) * OW + j, v);
        }
      } else {
        for (int idx = lane; idx < nenv * per; idx += EM_LANES) {
          const int e = idx / per, j = idx - e * per;
          dst[(int64_t)e * (N + 1) * OW + j] = tobs[e * LDO + j];
        }
      }
      if (with_states) {
#pragma unroll
        for (int s = 0; s < S; ++s) {
          T* sd = ka.straj[s] + b0 * (N + 1) + n0;
          for (int idx = lane; idx < nenv * cnt; idx += EM_LANES) {
            const int e = idx / cnt, t = idx - e * cnt;
            sd[(int64_t)e * (N + 1) + t] = tst[(s * EM_LANES + e) * LDS_ + t];
          }
        }
      }
    }
    if (n0 + cnt > N) {  // the tile that holds row N: publish last_state from registers
      if (active) {
#pragma unroll
        for (int j = 0; j < S; ++j) ka.last_state[j][i0] = sv[j];
      }
    }
    // the barrier after the next action-tile load also orders this flush before the next compute phase
  }
}

}  // namespace excenv
