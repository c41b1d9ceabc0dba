// Fused env-major (reference row-major) trajectory kernel: actions [B][K][A] in, observations [B][N+1][OW] and state
// leaves [B][N+1] out, with NO transposition pass. Time is the contiguous axis of these arrays while the parallelism
// runs across environments, so each wave (64 environments, one lane each) keeps the last TK SAVED STATES of every
// environment in an LDS ring and writes them out as per-environment runs of TK steps:
//   * ALIGNED WINDOWS (round 3). An environment's run is flushed when it completes a window whose first element sits on a
//     64-byte boundary of the state leaf: env e (row start e*(N+1) elements into the leaf) flushes steps [n-TK+1, n] at the
//     step n with (e*(N+1) + n + 1) % TK == 0. Every state-leaf run is then one whole, aligned 64-byte segment (round 2
//     flushed all environments at the same steps: 64-byte runs at arbitrary 4-byte offsets, i.e. partial bursts and 1.35x
//     the algorithmic HBM traffic) and every observation run (TK rows) starts on a TK*O*sizeof(T)-byte boundary. With
//     N + 1 odd, 64 / TK environments of a wave are due at every step — ONE flush round per step instead of TK rounds
//     every TK steps: the same work, spread evenly. Heads ([0, first boundary)) and tails (.., N]) are partial windows.
//   * state leaves: lanes (env, step) of a round store TK-word runs leaf by leaf (plain stores);
//   * observations: never kept per step — lane (env, step) re-reads that saved state from LDS, evaluates
//     generate_observation on it, the 64 rows of the round pass through a small staging buffer and leave as 64 consecutive
//     16-byte pieces per store instruction (whole rows of whole environments, non-temporal). Same device function on the
//     same saved state as the lane-major kernel: same bits.
// Actions: every 128-byte line of the [B][K][A] array is fetched from HBM exactly once. A lane's K*A-element row run is walked
// line by line: the line it is in lives in a per-lane LDS slot, the following line is prefetched into registers a whole line
// (128 / (A*sizeof(T)) steps) ahead and parked into the slot when the lane's next row crosses into it. Round 2 loaded TK + 1
// rows per lane and tile with 8-byte loads: a wave then touched ~100 of its 400 lines per tile, over the trajectory every line
// ~3 times, and with a few thousand waves in flight per XCD none of them survived in L2 — 10.4 GB fetched for 3.4 GB of actions.
// One wave per workgroup, so barriers are wave-local.
#pragma once
#include "kernels.hpp"

namespace excenv {

#ifndef EXCENV_EM_TK
#define EXCENV_EM_TK 8  // steps per flush window for 4-byte elements (8-byte elements: half, same LDS bytes): 32-byte runs.
                        // With aligned windows the run length no longer decides the write efficiency (every run is made of
                        // whole 32-byte sectors), the LDS ring of TK saved states per environment decides how many waves fit
                        // a CU: TK = 8 -> 26.5 KB per wave (PMSM), six waves per CU; TK = 16 -> 41 KB, three.
#endif
// Observation rows are written as non-temporal stores: complete lines that nothing reads back. The state leaves and the
// action loads stay cacheable — non-temporal state stores measured -40 %, non-temporal action loads -30 %: the L2 merges
// part of the state leaves' partial bursts and re-serves the action lines shared by consecutive tiles.
constexpr int EM_LANES = 64;  // one wave per workgroup
static_assert((EXCENV_EM_TK & (EXCENV_EM_TK - 1)) == 0 && EXCENV_EM_TK >= 2 && EXCENV_EM_TK <= EM_LANES,
              "EXCENV_EM_TK must be a power of two in [2, 64]");
// One wave per workgroup: LDS instructions of a wave execute in issue order, so a value one lane wrote is visible to the lane
// that reads it later in program order — no s_barrier and no s_waitcnt lgkmcnt(0) (which __syncthreads() implies) are needed,
// only the compiler must not reorder the accesses.
static_assert(EM_LANES == 64, "the fused env-major kernel relies on a single wave64 per workgroup");
__device__ __forceinline__ void wave_sync() { asm volatile("" ::: "memory"); }

#ifndef EXCENV_EM_DEBUG
#define EXCENV_EM_DEBUG 0  // experiments only (results are wrong): 1 skip the observation stores, 2 skip the flush, 4 skip the action-line walk
#endif

template <typename T> __host__ __device__ constexpr int em_tk() { return sizeof(T) == 4 ? EXCENV_EM_TK : EXCENV_EM_TK / 2; }

// LDS elements per wave: the per-lane action line (128 bytes + one 16-byte pad), the ring of saved states, one round of
// observation rows
template <typename T> __host__ __device__ constexpr size_t em_lds_elems(int /*A*/, int S, int O) {
  return (size_t)EM_LANES * (128 / sizeof(T) + 16 / sizeof(T)) + (size_t)S * EM_LANES * (em_tk<T>() + 1) + (size_t)EM_LANES * O;
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
  T* tact = reinterpret_cast<T*>(excenv_em_smem);  // [64][128 B + 16 B]: the action line each lane is in
  T* tst = tact + EM_LANES * (128 / sizeof(T) + 16 / sizeof(T));  // [S][64][TK + 1]: saved state j of env e at step n -> tst[(j * 64 + e) * LDS_ + n % TK]
  T* stage = tst + S * EM_LANES * LDS_;     // [64][O]: the observation rows of one flush round
  __shared__ unsigned row_off[EM_LANES];    // element offset (env, step) of each staged row within the workgroup, or ~0

  const int lane = threadIdx.x;
  const int64_t b0 = (int64_t)blockIdx.x * EM_LANES;
  const int64_t i0 = b0 + lane;
  const bool active = i0 < ka.B;
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
  // ---- actions: line store (see the header) ----
  constexpr int WPL = 128 / (int)sizeof(T);  // elements per 128-byte line
  constexpr int RPL = WPL / A;               // action rows per line (A divides a line: rows never straddle)
  constexpr int NPC = WPL / VW;              // 16-byte pieces per line
  constexpr int LPAD = WPL + VW;             // lane stride of the LDS line slots
  static_assert(WPL % A == 0, "an action row must not straddle a 128-byte line");
  T* const my_line = tact + lane * LPAD;
  // Lines are handled as ELEMENT offsets from ka.actions (a global-memory pointer the compiler can see through), counted from
  // the 128-byte boundary at or below the array start: off128 = elements between that boundary and ka.actions.
  const int64_t off128 = (int64_t)(((uintptr_t)ka.actions & 127u) / sizeof(T));
  const int64_t n_act = ka.B * ka.K * A;                                // elements in the array
  const int64_t row0 = off128 + (active ? i0 : 0) * ka.K * A;           // this lane's action row 0 (element offset from the boundary)
  int64_t cur_line = row0 & ~(int64_t)(WPL - 1);                        // the line in the LDS slot
  // a whole line into registers, no wait. Unconditional loads (a load under a condition would have to be merged with the
  // register's old value, which puts a wait in front of it): a piece that lies outside the array — possible only in the first
  // and the last line of the whole array; the host routes here only arrays that start on a 16-byte boundary and consist of
  // whole 16-byte pieces, so no piece straddles an end — is read from the nearest piece inside it instead and never used
  auto load_line = [&](int64_t line, T (&dst)[WPL]) {
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
      int64_t p = line + VW * i - off128;  // element index into ka.actions
      p = (p < 0) ? 0 : p;
      p = (p + VW > n_act) ? n_act - VW : p;
      T v[VW];
      load_v<T, VW>(ka.actions + p, v);
#pragma unroll
      for (int h = 0; h < VW; ++h) dst[i * VW + h] = v[h];
    }
  };
  auto park_line = [&](const T (&src)[WPL]) {
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
      T v[VW];
#pragma unroll
      for (int h = 0; h < VW; ++h) v[h] = src[i * VW + h];
      store_v<T, VW>(my_line + i * VW, v);
    }
  };
  auto read_row = [&](int64_t r, T (&a)[A]) {  // row at element offset r, which lies in the slot's line
    load_row<T, A>(my_line + (unsigned)(r & (WPL - 1)), a);
  };
  // the two register sets of the prefetched line alternate per window of RPL steps: the set a window parks from was loaded
  // during the previous window and made a plain register value at this window's start (one wait per window, at a point every
  // lane passes), the set it loads into is first read a window later — so no wait sits inside the per-step code
  T lineA[WPL], lineB[WPL];
  // flush-time role of this lane: (slot fel of the round, step ft of the window); per-workgroup bases + 32-bit offsets
  const int ft = lane % TK, fel = lane / TK;
  T* const wg_obs = ka.obs + b0 * (N + 1) * OW;
  // element index of this workgroup's first trajectory element within a leaf, modulo the window: the phase of env e's row is
  // (ph0 + e * (N + 1)) % TK (leaf bases are allocation-aligned; a base that is not shifts every window by the same amount)
  const unsigned rowlen = (unsigned)(N + 1);
  const unsigned ph0 = (unsigned)((b0 % TK) * ((N + 1) % TK)) % TK;
  const unsigned my_ph = (ph0 + (unsigned)lane * (rowlen % TK)) % TK;

  // ---- one flush round at step n: the (up to) EPR lowest environments of `mask` write their windows [a_e, n],
  // a_e = max(0, n - ((ph_e + n) % TK)); the served bits are cleared ----
  auto flush_round = [&](unsigned long long& mask, int64_t n) __attribute__((always_inline)) {
    {
      int e = -1;  // this lane's environment of the round: the fel-th lowest set bit of the wave-uniform mask
#pragma unroll
      for (int q = 0; q < EPR; ++q) {
        const int b = mask ? __builtin_ctzll(mask) : -1;
        if (mask) mask &= mask - 1;
        e = (q == fel) ? b : e;
      }
      const bool have = e >= 0;
      const int ee = have ? e : 0;
      const unsigned ph = (ph0 + (unsigned)ee * (rowlen % TK)) % TK;
      const int back = (int)((ph + (unsigned)(n % TK)) % TK);       // steps of the window before n
      const int64_t a = (n - back > 0) ? n - back : 0;               // first step of the window
      const int64_t sstep = a + ft;                                   // this lane's step
      const bool valid = have && sstep <= n;
      const int slot = (int)(sstep % TK);
      const unsigned row_el = (unsigned)(ee * (N + 1) + sstep);       // (env, step) as an element offset within the workgroup
      if constexpr (!BATCHED && (O % VW) == 0) {
        // Dense form (no control columns, rows made of whole 16-byte pieces): the 64 rows of the round go through the staging
        // buffer so that every store instruction of the wave writes 64 consecutive pieces = whole rows back to back
        constexpr int PR = O / VW;  // 16-byte pieces per row
        T fs[S], ob[O];
#pragma unroll
        for (int j = 0; j < S; ++j) fs[j] = tst[(j * EM_LANES + ee) * LDS_ + slot];
        M::observe(fs, c, ob);
#pragma unroll
        for (int q = 0; q < O; q += VW) {
          T v[VW];
#pragma unroll
          for (int h = 0; h < VW; ++h) v[h] = ob[q + h];
          store_v<T, VW>(stage + lane * O + q, v);  // row index within the round == lane (fel * TK + ft)
        }
        // the (env, step) offset of every row of the round, for the lanes that will store its pieces
        row_off[lane] = valid ? row_el : 0xffffffffu;
        wave_sync();
#pragma unroll
        for (int i = 0; i < PR; ++i) {
          const int p = lane + EM_LANES * i;  // piece index in round order == memory order within each environment's run
          const int row = p / PR, piece = p % PR;
          const unsigned ro = row_off[row];
          T v[VW];
          load_v<T, VW>(stage + p * VW, v);
          if (!(EXCENV_EM_DEBUG & 1) && ro != 0xffffffffu) store_stream<T, VW>(wg_obs + ro * (unsigned)O + (unsigned)(piece * VW), v);
        }
      } else {
        T fs[S];
#pragma unroll
        for (int j = 0; j < S; ++j) fs[j] = tst[(j * EM_LANES + ee) * LDS_ + slot];
        if (valid) {
          T ob[O];
          T* row = wg_obs + row_el * (unsigned)OW;
          if constexpr (BATCHED) {  // general path: env e's own normalisation bounds / reference columns
            Ctx<T, M> ce;
            load_ctx<true, T, M, false>(ce, ka.kp, b0 + ee, ka.dt, ka.env_tau, ka.adv_coef);  // used once: plain division
            M::observe(fs, ce, ob);
#pragma unroll
            for (int j = 0; j < EXCENV_MAX_CONTROL; ++j) {
              if (j < ka.n_control) {
                T x, lo, hi;
                pick_field<M, T>(fs, ce, ka.control_idx[j], x, lo, hi);
                row[O + j] = normalize(ka.reference[j][b0 + ee], lo, hi);
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
      // state leaves: the same (env, step) per lane, leaf by leaf — TK consecutive lanes write one aligned run
      if (with_states) {
        T v[S];
#pragma unroll
        for (int j = 0; j < S; ++j) v[j] = tst[(j * EM_LANES + ee) * LDS_ + slot];
        if (valid) {
#pragma unroll
          for (int j = 0; j < S; ++j) (ka.straj[j] + b0 * (N + 1))[row_el] = v[j];
        }
      }
      wave_sync();  // staging buffer / row_off free for the next round
    }
  };

  load_line(cur_line, lineA);
  park_line(lineA);
  load_line(cur_line + WPL, lineA);  // window 0 parks from A
  T a_cur[A], sv[S];
  read_row(row0, a_cur);

  // ---- one solver step: save row n, flush the environments whose window ends, advance ----
  auto do_step = [&](int64_t n, const T (&park_from)[WPL], T (&load_into)[WPL]) __attribute__((always_inline)) {
    const int slot = (int)(n % TK);
    // Row n + 1 of the actions (clamped) is what this step still needs (row n is in a_cur). When it starts the lane's next
    // line, that line moves from the registers into the slot and the one after it is requested. This block comes FIRST in the
    // step: whatever wait the compiler puts in front of it (the registers it reuses may be targets of an earlier request) then
    // only has to outlast the stores of the PREVIOUS step's flush, not stores issued a few instructions ago.
    const int64_t k1 = (n + 1 < ka.K) ? n + 1 : ka.K - 1;
    const int64_t r1 = row0 + k1 * A;
    if (!(EXCENV_EM_DEBUG & 4) && n < N && (r1 & ~(int64_t)(WPL - 1)) != cur_line) {
      park_line(park_from);
      cur_line += WPL;
      load_line(cur_line + WPL, load_into);
    }
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
    for (int j = 0; j < S; ++j) tst[(j * EM_LANES + lane) * LDS_ + slot] = sv[j];
    wave_sync();
    // environments whose window ends with step n (or with the trajectory)
    const bool due = active && ((((my_ph + (unsigned)slot + 1u) % TK) == 0u) || n == N);
    unsigned long long mask = __ballot(due);
    if (EXCENV_EM_DEBUG & 2) mask = 0;
    while (mask) flush_round(mask, n);
    if (n < N) {
      T a_nxt[A];
      read_row(r1, a_nxt);
      if constexpr (AHEAD) {
        env_advance_raw<M, SOLVER>(st, a_cur, a_nxt, n, k1, c, aux);
      } else {
        env_step<M, SOLVER>(st, a_cur, c);
      }
#pragma unroll
      for (int q = 0; q < A; ++q) a_cur[q] = a_nxt[q];
    }
  };
  auto landed = [&](T (&r)[WPL]) {  // the compiler's wait for this register set goes HERE
#pragma unroll
    for (int i = 0; i < WPL; ++i) asm volatile("" : "+v"(r[i]));
  };
  for (int64_t n0 = 0; n0 <= N; n0 += 2 * RPL) {
    landed(lineA);
    for (int64_t n = n0; n < n0 + RPL && n <= N; ++n) do_step(n, lineA, lineB);
    landed(lineB);
    for (int64_t n = n0 + RPL; n < n0 + 2 * RPL && n <= N; ++n) do_step(n, lineB, lineA);
  }
  if (active) {  // row N was saved last: publish last_state from registers
#pragma unroll
    for (int j = 0; j < S; ++j) ka.last_state[j][i0] = sv[j];
  }
}

}  // namespace excenv
