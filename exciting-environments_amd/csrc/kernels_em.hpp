// Fused env-major (reference row-major) trajectory kernel: actions [B][K][A] in, observations [B][N+1][OW] and state
// leaves [B][N+1] out, with NO transposition pass. Time is the contiguous axis of these arrays while the parallelism
// runs across environments, so each wave (64 environments, one lane each) stages the SAVED STATES of TK solver steps in
// LDS and then writes every environment's TK-step run contiguously:
//   * state leaves: TK-word runs (64 bytes for fp32 at TK = 16), lanes (env, step) walk the tile leaf by leaf;
//   * observations: not staged at all — lane (env, step) re-reads that saved state from LDS, evaluates
//     generate_observation on it and stores the OW-word row, so TK consecutive lanes write one env's TK rows = one
//     contiguous run (512 bytes for PMSM fp32). Same device function on the same saved state as the lane-major kernel:
//     same bits.
// Only the states (S words per env-step) and the action tile live in LDS: 38.3 KB per wave for PMSM at TK = 16 (fp32; fp64
// stages 8 steps in the same bytes), four waves per CU. Measured on PMSM Euler fp32, B = 2^22: TK = 4 / 8 /
// 16 -> 16.9 / 13.5 / 11.9 ms per 100-step launch: the 4-byte-per-step state leaves want long runs more than the CU
// wants more resident waves. One wave per workgroup, so the two barriers per tile are wave-local.
#pragma once
#include "kernels.hpp"

namespace excenv {

#ifndef EXCENV_EM_TK
#define EXCENV_EM_TK 16  // solver steps staged per tile for 4-byte elements (8-byte elements: half, same LDS bytes)
#endif
// Observation rows are written as non-temporal stores: complete lines that nothing reads back. The state leaves and the
// action loads stay cacheable — non-temporal state stores measured -40 %, non-temporal action loads -30 %: the L2 merges
// part of the state leaves' partial bursts and re-serves the action lines shared by consecutive tiles.
constexpr int EM_LANES = 64;  // one wave per workgroup
static_assert((EXCENV_EM_TK & (EXCENV_EM_TK - 1)) == 0 && EXCENV_EM_TK >= 2 && EXCENV_EM_TK <= EM_LANES,
              "EXCENV_EM_TK must be a power of two in [2, 64]");
template <typename T> __host__ __device__ constexpr int em_tk() { return sizeof(T) == 4 ? EXCENV_EM_TK : EXCENV_EM_TK / 2; }

template <typename T> __host__ __device__ constexpr size_t em_lds_elems(int A, int S) {
  return (size_t)EM_LANES * (em_tk<T>() + 1) * A + (size_t)S * EM_LANES * (em_tk<T>() + 1);
}

template <class M, typename T, int SOLVER, bool AHEAD, bool BATCHED>
__global__ void __launch_bounds__(EM_LANES) sim_ahead_em_kernel(const SimArgs<T, M> ka) {
  constexpr int S = M::S, A = M::A, O = M::O, TK = em_tk<T>();
  constexpr int VW = 16 / (int)sizeof(T);  // elements per 16-byte piece
  constexpr int EPR = EM_LANES / TK;       // environments covered by one flush round of the wave
  extern __shared__ __align__(16) unsigned char excenv_em_smem[];
  const int OW = O + ka.n_control;
  const bool with_states = ka.straj[0] != nullptr;
  constexpr int LDS_ = TK + 1;  // odd leading dimension: conflict-free columns
  T* tact = reinterpret_cast<T*>(excenv_em_smem);  // [TK + 1][64][A]: action row t of the tile, lane-major
  T* tst = tact + (TK + 1) * EM_LANES * A;  // [S][64][TK + 1]: saved state j of env e at step t -> tst[(j * 64 + e) * LDS_ + t]

  const int lane = threadIdx.x;
  const int64_t b0 = (int64_t)blockIdx.x * EM_LANES;
  const int64_t i0 = b0 + lane;
  const bool active = i0 < ka.B;
  const int nenv = (int)((ka.B - b0 < EM_LANES) ? (ka.B - b0) : EM_LANES);  // envs of this workgroup
  Ctx<T, M> c;
  load_ctx<BATCHED>(c, ka.kp, active ? i0 : 0, ka.dt, ka.env_tau, ka.adv_coef);
  c.lin_stop = ka.lin_stop;
  c.lin_div = T(ka.K - 1);
  c.lin_last = ka.K - 1;

  T st[S];
#pragma unroll
  for (int j = 0; j < S; ++j) st[j] = active ? ka.state_in[j][i0] : T(0);
  AheadAux<T> aux;
  if constexpr (AHEAD && M::IS_PMSM) {
    aux.eps0 = st[2];
    aux.prev_clip[0] = st[0];
    aux.prev_clip[1] = st[1];
  }
  const bool deadtime_on = M::IS_PMSM ? (c.P[M::P - 1] > T(0)) : false;

  const int64_t N = ka.K;  // substeps == 1 on this path (host); K >= 1 (host)
  // Action rows n0 .. n0+TK of THIS lane's environment (one row past the tile: RK stages with c == 1 and the prefetch of the
  // next step read row n+1) are loaded one tile ahead into registers — unconditional loads of always-valid rows (the row
  // index is clamped to K-1), each lane walking its own contiguous [K][A] run — and parked lane-major in LDS after the
  // compute phase, so their HBM latency hides behind TK solver steps and no cross-lane index arithmetic is needed.
  constexpr int NA = TK + 1;
  T areg[NA][A];
  const T* arow = ka.actions + (active ? i0 : 0) * ka.K * A;
  auto fetch_actions = [&](int64_t n0) {  // global -> registers, no wait
#pragma unroll
    for (int t = 0; t < NA; ++t) {
      int64_t k = n0 + t;
      k = (k < ka.K) ? k : ka.K - 1;
      load_row<T, A>(arow + k * A, areg[t]);
    }
  };
  auto park_actions = [&]() {  // registers -> LDS [NA][64][A]
#pragma unroll
    for (int t = 0; t < NA; ++t) store_row<T, A>(&tact[(t * EM_LANES + lane) * A], areg[t]);
  };
  // flush-time role of this lane: (environment offset fel within a round, step ft); per-workgroup bases + 32-bit offsets
  const int ft = lane % TK, fel = lane / TK;
  T* const wg_obs = ka.obs + b0 * (N + 1) * OW;
  const unsigned obs_round = (unsigned)(EPR * (N + 1) * OW);  // element offset between consecutive flush rounds
  const unsigned st_round = (unsigned)(EPR * (N + 1));

  fetch_actions(0);
  park_actions();
  T sv[S];
  for (int64_t n0 = 0; n0 <= N; n0 += TK) {
    const int cnt = (int)((N + 1 - n0 < TK) ? (N + 1 - n0) : TK);  // rows n0 .. n0+cnt-1
    __syncthreads();                                               // previous flush has read its rows of tst
    fetch_actions(n0 + TK);                                        // next tile's loads fly during this tile's steps
    // ---- TK solver steps, saved states staged in LDS ----
    for (int t = 0; t < cnt; ++t) {
      const int64_t n = n0 + t;
#pragma unroll
      for (int j = 0; j < S; ++j) sv[j] = st[j];
      if constexpr (AHEAD) {
        M::post(sv, c);
        if constexpr (M::IS_PMSM) {
          if (deadtime_on) {
            sv[0] = aux.prev_clip[0];  // row 0: still the initial buffer
            sv[1] = aux.prev_clip[1];
          } else {
            sv[0] = T(0);
            sv[1] = T(0);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < S; ++j) tst[(j * EM_LANES + lane) * LDS_ + t] = sv[j];
      if (n < N) {
        T a_cur[A], a_nxt[A];
        load_row<T, A>(&tact[(t * EM_LANES + lane) * A], a_cur);
        load_row<T, A>(&tact[((t + 1) * EM_LANES + lane) * A], a_nxt);
        if constexpr (AHEAD) {
          const int64_t k1 = (n + 1 < ka.K) ? n + 1 : ka.K - 1;
          env_advance_raw<M, SOLVER>(st, a_cur, a_nxt, n, k1, c, aux);
        } else {
          env_step<M, SOLVER>(st, a_cur, c);
        }
      }
    }
    if (n0 + cnt > N) {  // the tile that holds row N: publish last_state from registers
      if (active) {
#pragma unroll
        for (int j = 0; j < S; ++j) ka.last_state[j][i0] = sv[j];
      }
    }
    __syncthreads();
    // The prefetched action rows must have LANDED before the flush issues its stores: their loads were requested a whole
    // tile ago, so waiting here costs nothing, whereas a wait after the flush would — vmcnt counts in order — drain every
    // trajectory store of this tile. The empty asm makes each register a use at this point (the compiler puts its
    // s_waitcnt here); the rows are parked in LDS after the flush, which borrows the action tile as staging space.
#pragma unroll
    for (int t = 0; t < NA; ++t)
#pragma unroll
      for (int q = 0; q < A; ++q) asm volatile("" : "+v"(areg[t][q]));
    // ---- flush ----
    // observations: EPR environments per round, lane (fel, ft) re-creates the row of env (r * EPR + fel) at step ft
    if constexpr (!BATCHED && (O % VW) == 0) {
      // Dense form (no control columns, rows made of whole 16-byte pieces): the 64 rows of a round go through a staging
      // buffer in the (idle) action tile so that every store instruction of the wave writes 64 consecutive pieces = whole
      // rows of EPR / PR... environments back to back (1 KiB per instruction), non-temporal.
      constexpr int PR = O / VW;  // 16-byte pieces per row
      static_assert((TK + 1) * A >= O, "the action tile must be able to stage one round of observation rows");
      T* const stage = tact;
      for (int r = 0; r < TK; ++r) {
        T fs[S], ob[O];
#pragma unroll
        for (int j = 0; j < S; ++j) fs[j] = tst[(j * EM_LANES + r * EPR + fel) * LDS_ + ft];
        M::observe(fs, c, ob);
#pragma unroll
        for (int q = 0; q < O; q += VW) {
          T v[VW];
#pragma unroll
          for (int h = 0; h < VW; ++h) v[h] = ob[q + h];
          store_v<T, VW>(stage + lane * O + q, v);  // row index within the round == lane (fel * TK + ft)
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < PR; ++i) {
          const int p = lane + EM_LANES * i;  // piece index in round order == memory order within each environment
          const int row = p / PR, piece = p % PR;
          const int el = row / TK, step = row % TK, e = r * EPR + el;
          T v[VW];
          load_v<T, VW>(stage + p * VW, v);
          if (step < cnt && e < nenv)
            store_stream<T, VW>(wg_obs + (unsigned)((e * (N + 1) + n0 + step) * O + piece * VW), v);
        }
        __syncthreads();
      }
    } else {
      unsigned off = (unsigned)((fel * (N + 1) + n0 + ft) * OW);
      for (int r = 0; r < TK; ++r, off += obs_round) {
        const int e = r * EPR + fel;
        T fs[S];
#pragma unroll
        for (int j = 0; j < S; ++j) fs[j] = tst[(j * EM_LANES + e) * LDS_ + ft];
        if (ft < cnt && e < nenv) {
          T ob[O];
          T* row = wg_obs + off;
          if constexpr (BATCHED) {  // general path: env e's own normalisation bounds / reference columns
            Ctx<T, M> ce;
            load_ctx<true, T, M, false>(ce, ka.kp, b0 + e, ka.dt, ka.env_tau, ka.adv_coef);  // used once: plain division
            M::observe(fs, ce, ob);
#pragma unroll
            for (int j = 0; j < EXCENV_MAX_CONTROL; ++j) {
              if (j < ka.n_control) {
                T x, lo, hi;
                pick_field<M, T>(fs, ce, ka.control_idx[j], x, lo, hi);
                row[O + j] = normalize(ka.reference[j][b0 + e], lo, hi);
              }
            }
          } else {
            M::observe(fs, c, ob);
          }
          if ((O % VW) == 0 && (OW % VW) == 0) {  // every row start is 16-byte aligned
#pragma unroll
            for (int q = 0; q + VW <= O; q += VW) {
              T v[VW];
#pragma unroll
              for (int h = 0; h < VW; ++h) v[h] = ob[q + h];
              store_v<T, VW>(row + q, v);
            }
          } else {
#pragma unroll
            for (int q = 0; q < O; ++q) row[q] = ob[q];
          }
        }
      }
    }
    // state leaves: TK rounds per leaf, all LDS reads of a leaf in flight before its stores
    if (with_states) {
      const unsigned off0 = (unsigned)(fel * (N + 1) + n0 + ft);
#pragma unroll
      for (int s = 0; s < S; ++s) {
        T v[TK];
#pragma unroll
        for (int it = 0; it < TK; ++it) v[it] = tst[(s * EM_LANES + fel + it * EPR) * LDS_ + ft];
        T* sd = ka.straj[s] + b0 * (N + 1);
        if (ft < cnt) {
#pragma unroll
          for (int it = 0; it < TK; ++it) {
            if (fel + it * EPR < nenv) sd[off0 + it * st_round] = v[it];  // plain: L2 merges part of these partial bursts
          }
        }
      }
    }
    park_actions();  // every lane parks and later reads only its own column of tact; the loop-top barrier publishes it
  }
}

}  // namespace excenv
