// Fused env-major (reference row-major) trajectory kernel: actions [B][K][A] in, observations [B][N+1][OW] and state
// leaves [B][N+1] out, with NO transposition pass. Time is the contiguous axis of these arrays while the parallelism
// runs across environments, so each wave (64 environments, one lane each) stages TK solver steps in LDS and then
// writes every environment's TK-step run contiguously: 256-byte runs for PMSM observations (16 B per lane), TK-word
// runs for the state leaves. One wave per workgroup, so the two barriers per tile are wave-local.
#pragma once
#include "kernels.hpp"

namespace excenv {

#ifndef EXCENV_EM_TK
#define EXCENV_EM_TK 8
#endif
constexpr int EM_TK = EXCENV_EM_TK;  // solver steps staged per tile
constexpr int EM_LANES = 64;  // one wave per workgroup

// Leading dimension of the observation tile: rows of TK*OW words + a pad that keeps 16-byte row alignment when OW is a
// multiple of the 16-byte vector width (then ds_write_b128 / ds_read_b128 are conflict-free: consecutive lanes / rows sit
// 4 banks apart modulo 32), else one word (odd stride for the scalar accesses).
template <typename T> __host__ __device__ constexpr int em_ldo(int OW) {
  constexpr int VW = 16 / (int)sizeof(T);
  return (OW % VW) == 0 ? EM_TK * OW + VW : ((EM_TK * OW) | 1);
}
template <typename T> __host__ __device__ constexpr size_t em_lds_elems(int A, int OW, int S, bool with_states) {
  return (size_t)EM_LANES * ((EM_TK + 1) * A + 1) + (size_t)EM_LANES * em_ldo<T>(OW) +
         (with_states ? (size_t)S * EM_LANES * (EM_TK + 1) : 0);
}

// Walks idx = lane, lane + 64, ... over a [nenv][per] index space without per-iteration divisions:
// (e, j) = (idx / per, idx % per) advanced by (64 / per, 64 % per) with carry.
struct EmWalk {
  int e, j, qe, qj, per;
  __device__ __forceinline__ EmWalk(int lane, int per_) : per(per_) {
    e = lane / per_;
    j = lane - e * per_;
    qe = EM_LANES / per_;
    qj = EM_LANES - qe * per_;
  }
  __device__ __forceinline__ void next() {
    e += qe;
    j += qj;
    if (j >= per) { j -= per; ++e; }
  }
};

template <class M, typename T, int SOLVER, bool AHEAD, bool BATCHED>
__global__ void __launch_bounds__(EM_LANES) sim_ahead_em_kernel(const SimArgs<T, M> ka) {
  constexpr int S = M::S, A = M::A, O = M::O, TK = EM_TK;
  extern __shared__ __align__(16) unsigned char excenv_em_smem[];
  const int OW = O + ka.n_control;
  const bool with_states = ka.straj[0] != nullptr;
  constexpr int VW = 16 / (int)sizeof(T);  // elements per 16-byte piece
  static_assert((TK & (TK - 1)) == 0 && EM_LANES % TK == 0, "EXCENV_EM_TK must be a power of two <= 64");
  const bool vec_rows = (OW % VW) == 0;  // every env row, tile start and LDS row is then 16-byte aligned
  const int LDA = (TK + 1) * A + 1, LDO = em_ldo<T>(OW), LDS_ = TK + 1;  // odd leading dimensions: conflict-free columns
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
  // Action tiles are prefetched one tile ahead into registers (NA_IT words per lane, all loads in flight together) and
  // parked in LDS after the compute phase, so their HBM latency hides behind TK solver steps.
  constexpr int NA_IT = (TK + 1) * A;
  T areg[NA_IT];
  auto na_of = [&](int64_t n0) -> int {
    int64_t na = ka.K - n0;
    na = na < (TK + 1) ? na : (TK + 1);
    return (int)(na < 0 ? 0 : na);
  };
  auto fetch_actions = [&](int64_t n0) {  // global -> registers, no wait
    const int per = na_of(n0) * A;
    if (per == 0) return;
    const T* src = ka.actions + (b0 * ka.K + n0) * A;
    EmWalk w(lane, per);
#pragma unroll
    for (int it = 0; it < NA_IT; ++it) {
      if (w.e < nenv) areg[it] = src[(int64_t)w.e * ka.K * A + w.j];
      w.next();
    }
  };
  auto park_actions = [&](int64_t n0) {  // registers -> LDS
    const int per = na_of(n0) * A;
    if (per == 0) return;
    EmWalk w(lane, per);
#pragma unroll
    for (int it = 0; it < NA_IT; ++it) {
      if (w.e < nenv) tact[w.e * LDA + w.j] = areg[it];
      w.next();
    }
  };
  fetch_actions(0);
  park_actions(0);
  for (int64_t n0 = 0; n0 <= N; n0 += TK) {
    const int cnt = (int)((N + 1 - n0 < TK) ? (N + 1 - n0) : TK);  // rows n0 .. n0+cnt-1
    const int na = na_of(n0);                                      // actions n0 .. n0+na-1 (one ahead)
    __syncthreads();                                               // action tile parked; previous flush has read its rows
    fetch_actions(n0 + TK);                                        // next tile's loads fly during this tile's steps
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
      if constexpr (O % VW == 0) {
        if (vec_rows) {  // 16-byte LDS stores (row base and t * OW are multiples of VW)
#pragma unroll
          for (int q = 0; q < O; q += VW) {
            T v[VW];
#pragma unroll
            for (int h = 0; h < VW; ++h) v[h] = ob[q + h];
            store_v<T, VW>(&tobs[lane * LDO + t * OW + q], v);
          }
        } else {
#pragma unroll
          for (int q = 0; q < O; ++q) tobs[lane * LDO + t * OW + q] = ob[q];
        }
      } else {
#pragma unroll
        for (int q = 0; q < O; ++q) tobs[lane * LDO + t * OW + q] = ob[q];
      }
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
    // The next action tile is parked BEFORE the flush: its loads were issued a whole tile ago, so the s_waitcnt in front of
    // these LDS writes is already satisfied — and it must not come after the flush, where the in-order vmcnt would make it
    // wait for every trajectory store of this tile (a full memory drain per tile). This tile's steps are done with tact
    // (barrier above) and the flush below reads only tobs / tst; the barrier at the loop top publishes the new tile.
    park_actions(n0 + TK);
    // ---- flush: per-env contiguous runs ----
    // Full tiles take the batched paths: all LDS reads of a batch are issued before the first global store, so the wave
    // pays one LDS round trip per batch instead of one per word (the trip counts are static, nothing is loop-carried).
    {
      const int per = cnt * OW;
      T* dst = ka.obs + (b0 * (N + 1) + n0) * OW;
      if (vec_rows && cnt == TK) {
        const int CH = per / VW;  // 16-byte pieces per env; the wave walks 64 * CH pieces in CH rounds
        EmWalk w(lane, CH);
        for (int it0 = 0; it0 < CH; it0 += 4) {
          T v[4][VW];
          int es[4], js[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            es[u] = w.e;
            js[u] = w.j * VW;
            const int er = (w.e < EM_LANES) ? w.e : EM_LANES - 1;  // rounds past CH stay inside the tile (masked below)
            load_v<T, VW>(&tobs[er * LDO + js[u]], v[u]);
            w.next();
          }
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (it0 + u < CH && es[u] < nenv) store_v<T, VW>(dst + (int64_t)es[u] * (N + 1) * OW + js[u], v[u]);
        }
      } else if (vec_rows) {
        for (EmWalk w(lane, per / VW); w.e < nenv; w.next()) {
          const int j = w.j * VW;
          T v[VW];
          load_v<T, VW>(&tobs[w.e * LDO + j], v);
          store_v<T, VW>(dst + (int64_t)w.e * (N + 1) * OW + j, v);
        }
      } else {
        for (EmWalk w(lane, per); w.e < nenv; w.next()) dst[(int64_t)w.e * (N + 1) * OW + w.j] = tobs[w.e * LDO + w.j];
      }
      if (with_states) {
        if (cnt == TK) {  // lane -> (env e0 + it * 64/TK, step j): TK rounds per leaf, all reads of a leaf in flight together
          constexpr int EPR = EM_LANES / TK;  // envs covered per round
          const int e0 = lane / TK, j = lane % TK;
#pragma unroll
          for (int s = 0; s < S; ++s) {
            T v[TK];
#pragma unroll
            for (int it = 0; it < TK; ++it) v[it] = tst[(s * EM_LANES + e0 + it * EPR) * LDS_ + j];
            T* sd = ka.straj[s] + b0 * (N + 1) + n0 + j;
#pragma unroll
            for (int it = 0; it < TK; ++it) {
              const int e = e0 + it * EPR;
              if (e < nenv) sd[(int64_t)e * (N + 1)] = v[it];
            }
          }
        } else {
          const EmWalk w0(lane, cnt);
#pragma unroll
          for (int s = 0; s < S; ++s) {
            T* sd = ka.straj[s] + b0 * (N + 1) + n0;
            for (EmWalk w = w0; w.e < nenv; w.next()) sd[(int64_t)w.e * (N + 1) + w.j] = tst[(s * EM_LANES + w.e) * LDS_ + w.j];
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
  }
}

}  // namespace excenv
