// Fused env-major (reference row-major) trajectory kernel, register-ring form (round 3): actions [B][K][A] in, observations
// [B][N+1][O] and state leaves [B][N+1] out, no transposition pass — and every store instruction writes whole aligned runs
// (128-byte lines, or 64-byte half lines for the models with many state leaves).
//
// Why a second form. The LDS-ring kernel (kernels_em.hpp) keeps TK = 8 saved states per environment in LDS (more does not fit
// next to enough waves), writes the state leaves as 32-byte runs and runs 1.2 waves per SIMD; what its stores cost is set by
// the memory system (tools/em_placement.py: the same launch takes 9.3 ... 13.1 ms depending only on where the leaves lie).
// Here a lane keeps a window of W saved states of ITS environment in registers — one native W-element vector per state leaf,
// written with a wave-uniform dynamic index (s_set_gpr_idx_on + v_mov) — so a window of a leaf is one aligned run per
// environment, LDS holds no history, and two to three waves fit a SIMD.
//   * UNIFORM PHASE. A run of env e's leaf row starts where (e * (N + 1) + n) % W == 0. Lanes of a wave take environments P
//     apart, P = the period of that phase in e (a power of two <= W; the host computes it together with the same period of the
//     action rows' line phase and passes the larger one): all 64 environments of a wave cross their run boundaries at the same
//     steps, every ring index and every branch of the flush is wave-uniform.
//   * FLUSH, once per W steps (plus head and tail): per leaf the NPC = W * sizeof(T) / 16 sixteen-byte pieces of a lane's run go
//     through an LDS buffer (64 runs) in which each group of NPC lanes transposes its NPC x NPC pieces, so that a store
//     instruction's adjacent lanes write the adjacent pieces of ONE environment's run: 64 lanes x 16 bytes = 64 / NPC whole runs
//     per instruction, non-temporal. The observation rows of the window (O runs per environment) are evaluated from the ring
//     at flush time (same device function on the same saved state as every other kernel: same bits) and leave the same way.
//     Head / tail windows and ragged waves use the same code with a slot range; pieces cut by the range fall back to element
//     stores.
//   * ACTIONS. As in the LDS-ring kernel every 128-byte line is fetched once and parked in a per-lane LDS slot when the walk
//     crosses into it; with the uniform phase the crossing is wave-uniform.
// One wave per workgroup (LDS accesses of a wave execute in order: compiler fences only).
#pragma once
#include <type_traits>
#include "kernels_em.hpp"

namespace excenv {

template <typename T, int W> struct EmrVec { typedef T type __attribute__((ext_vector_type(W))); };

// Leaves that need no window: one that never changes along a trajectory — PMSM's omega_el (pmsm_env.py:509-523: the ODE has
// no equation for it; sim_ahead keeps it constant, :785-791) — and one that is a function of other saved leaves: PMSM's torque
// in the reference-structured trajectory, where every saved row is post-processed (pmsm_env.py:573-578, 680-688: torque from
// the saved currents; the same device function M::torque here). On the step-semantics path row 0 carries the caller's torque
// as it came in, so there it stays in the ring. -1: none.
template <class M> constexpr int emr_const_leaf() { return M::IS_PMSM ? 6 : -1; }
template <class M, bool AHEAD> constexpr int emr_derived_leaf() { return (M::IS_PMSM && AHEAD) ? 5 : -1; }
template <class M, bool AHEAD> constexpr int emr_ring_leaves() {
  return M::S - (emr_const_leaf<M>() >= 0 ? 1 : 0) - (emr_derived_leaf<M, AHEAD>() >= 0 ? 1 : 0);
}
// Steps per window. Two waves must share a SIMD (one wave alone leaves the VALU half idle: 9.9 ms for the headline launch with
// 128-byte windows at one wave per SIMD, 7.7 ms with 64-byte windows at two), so a lane has 256 registers and the windows of
// all ring leaves must fit next to the integration's own: 128-byte runs (whole lines, 32 registers per leaf) while the ring
// stays within EXCENV_EMR_MAX_RING_REGS, else 64-byte runs (half lines, written 4 lanes x 16 bytes). PMSM in fp64 (5 ... 6 leaves
// x 8 doubles next to a double-precision integration) fits since the torque leaf left the ring and the action line is loaded at
// the crossing: two registers are spilled, reloaded only on the IEEE-division fallback path of the flush.
#ifndef EXCENV_EMR_MAX_RING_REGS
#define EXCENV_EMR_MAX_RING_REGS 128
#endif
template <class M, typename T, bool AHEAD> constexpr int emr_rows() {  // a double-precision integration needs twice the registers itself
  return (emr_ring_leaves<M, AHEAD>() * 32 <= EXCENV_EMR_MAX_RING_REGS / ((int)sizeof(T) / 4) ? 128 : 64) / (int)sizeof(T);
}
template <class M, typename T> constexpr bool emr_supported() { return !M::HAS_LUT; }  // the look-up model keeps the LDS-ring kernel

// LDS bytes per wave: the transposition buffer (64 lanes x 128 bytes) and the action line slots (64 x 128 bytes)
template <class M, typename T, bool AHEAD> constexpr size_t emr_lds_bytes() { return (size_t)EM_LANES * (128 + 128); }

#ifndef EXCENV_EMR_DEBUG
#define EXCENV_EMR_DEBUG 0  // experiments only (results are wrong): 1 never walk to the next action line, 2 no flush, 4 flush without global stores
#endif
#ifndef EXCENV_EMR_NT
#define EXCENV_EMR_NT 1  // whole-run stores of the flush are non-temporal (plain stores: 10 ... 11 ms instead of 7 for the headline launch)
#endif

// ka.a_wg carries P (environments between consecutive lanes of a wave) on this path.
template <class M, typename T, int SOLVER, bool AHEAD>
__global__ void __launch_bounds__(EM_LANES) __attribute__((amdgpu_waves_per_eu(2))) sim_ahead_emr_kernel(const SimArgs<T, M> ka) {
  constexpr int S = M::S, A = M::A, O = M::O;
  constexpr int VW = 16 / (int)sizeof(T);   // elements per 16-byte piece
  constexpr int W = emr_rows<M, T, AHEAD>();  // steps per window == elements per run of a state leaf
  constexpr int NPC = W / VW;               // 16-byte pieces per run (4 or 8); the lanes of a wave transpose in groups of NPC
  constexpr int WL = 128 / (int)sizeof(T);  // elements per 128-byte line (action array, observation lines)
  constexpr int NPL = 8;                    // pieces per line
  // Observation rows always leave as whole 128-byte lines (8 pieces, groups of 8 lanes), whatever the leaves' run length: the
  // W rows of a window are W * O / WL lines per environment
  constexpr int RPO = WL / O;               // saved rows per observation line
  constexpr int NLO = W / RPO;              // observation lines per window and environment
  static_assert(WL % O == 0 && W % RPO == 0, "the observation rows of a window must be whole 128-byte lines");
  static_assert(WL % A == 0 && VW % A == 0, "an action row must not straddle a 16-byte piece");
  using Vec = typename EmrVec<T, W>::type;
  extern __shared__ __align__(16) unsigned char excenv_emr_smem[];
  T* const xp = reinterpret_cast<T*>(excenv_emr_smem);  // [NPC pieces][64 lanes]: piece p of lane l at lane position l ^ p (both directions conflict-free)
  T* const slot_a = xp + EM_LANES * WL;                 // [8 pieces][64 lanes]: the action line each lane is in

  const int lane = threadIdx.x;
  const int P = (int)ka.a_wg;
  const int64_t wv = blockIdx.x;
  const int r = (int)(wv % P);
  const int64_t env0 = (wv / P) * ((int64_t)EM_LANES * P) + r;  // lane 0's environment
  const int64_t env = env0 + (int64_t)P * lane;
  const bool active = env < ka.B;
  Ctx<T, M> c;
  load_ctx<false>(c, ka.kp, 0, ka.dt, ka.env_tau, ka.adv_coef);
  c.lin_stop = ka.lin_stop;
  c.lin_div = T(ka.K - 1);
  c.lin_last = ka.K - 1;
  T st[S];
#pragma unroll
  for (int j = 0; j < S; ++j) st[j] = active ? ka.state_in[j][env] : T(0);
  AheadAux<T> aux;
  if constexpr (AHEAD && M::IS_PMSM) {
    aux.eps0 = st[2];
    aux.prev_clip[0] = st[0];
    aux.prev_clip[1] = st[1];
  }
  const bool deadtime_on = M::IS_PMSM ? (c.P[M::P - 1] > T(0)) : false;
  const bool with_states = ka.straj[0] != nullptr;

  const int N = (int)ka.K;  // substeps == 1 on this path; 1 <= K and 64 * P * (K + 1) * O * sizeof(T) < 2^31 (host)
  const int64_t rowlen = N + 1;
  // wave-uniform window phase: slot of row n in the ring = (ph + n) % W (leaf and observation bases are 128-byte aligned, host)
  const int ph = (int)(((int64_t)r * (rowlen % W)) % W);

  // ---- actions: line store with a wave-uniform phase ----
  const int64_t off128 = (int64_t)(((uintptr_t)ka.actions & 127u) / sizeof(T));
  const int64_t n_act = ka.B * ka.K * A;
  const int64_t row0 = off128 + (active ? env : env0) * ka.K * A;  // this lane's action row 0 (element offset from the boundary)
  const int pha = (int)((off128 + (int64_t)r * ((ka.K * A) % WL)) % WL);  // == row0 % WL for every lane
  const int64_t line0 = row0 - pha;                                  // this lane's first line
  auto load_line = [&](int64_t li, T (&dst)[WL]) {  // line number li of the lane's walk into registers, no wait
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      int64_t p = line0 + li * WL + VW * i - off128;  // element index into ka.actions
      p = (p < 0) ? 0 : p;
      p = (p + VW > n_act) ? n_act - VW : p;
      T v[VW];
      load_v<T, VW>(ka.actions + p, v);
#pragma unroll
      for (int h = 0; h < VW; ++h) dst[i * VW + h] = v[h];
    }
  };
  auto park_line = [&](const T (&src)[WL]) {
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      T v[VW];
#pragma unroll
      for (int h = 0; h < VW; ++h) v[h] = src[i * VW + h];
      store_v<T, VW>(slot_a + (i * EM_LANES + lane) * VW, v);
    }
  };
  auto read_row = [&](int idx, T (&a)[A]) {  // row at element idx of the parked line
    load_row<T, A>(slot_a + ((idx / VW) * EM_LANES + lane) * VW + idx % VW, a);
  };

  if (env0 >= ka.B) return;  // no barrier is ever used: a wave without environments may leave
  constexpr int CL = emr_const_leaf<M>();           // this leaf's saved value is st[CL] at every step
  constexpr int DL = emr_derived_leaf<M, AHEAD>();  // this leaf's saved value is a function of other saved leaves
  constexpr int NR = emr_ring_leaves<M, AHEAD>();
  auto ridx = [](int j) constexpr { return j - ((CL >= 0 && j > CL) ? 1 : 0) - ((DL >= 0 && j > DL) ? 1 : 0); };
  Vec ring[NR];
#pragma unroll
  for (int j = 0; j < NR; ++j) ring[j] = (Vec)(T(0));
  auto ring_get = [&](int j, int s_) __attribute__((always_inline)) -> T {  // j: compile-time constant at every call
    if (j == CL) return st[CL >= 0 ? CL : 0];
    if constexpr (M::IS_PMSM) {
      if (j == DL) return M::torque(ring[ridx(3)][s_], ring[ridx(4)][s_], c);
    }
    return ring[ridx(j)][s_];
  };
  T* const obs_base = ka.obs;

  // ---- flush the ring slots [s_lo, s_hi]; slot 0 is row n_slot0 of every environment of the wave ----
  const bool full_wave = env0 + (int64_t)P * (EM_LANES - 1) < ka.B;  // every lane has an environment (wave-uniform)
  // One run per environment through the transposition buffer: put_pieces writes this lane's run, emit_lines stores this lane's
  // piece of the runs of its group's environments (groups of NP lanes, NP pieces per run); `slot_of(h)`: ring slot that element
  // h of this lane's piece belongs to.
  auto put_pieces = [&](auto&& elem) __attribute__((always_inline)) {  // elem(h): element h of this lane's run, produced piece by piece
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
      T v[VW];
#pragma unroll
      for (int h = 0; h < VW; ++h) v[h] = elem(i * VW + h);
      store_v<T, VW>(xp + (i * EM_LANES + (lane ^ i)) * VW, v);
    }
    wave_sync();
  };
  // Addresses: a wave-uniform base (scalar registers) plus a 32-bit lane offset — global_store with an SGPR base. Per-lane 64-bit
  // pointers per leaf were hoisted out of the step loop by the compiler, spilled, and every reload (scratch_load + s_waitcnt
  // vmcnt(0)) then waited for the previous leaf's stores to complete.
  auto emit_lines = [&](auto np_tag, T* ubase, int64_t q_stride, unsigned lane_off, bool fast, int s_lo, int s_hi, auto&& slot_of) __attribute__((always_inline)) {
    constexpr int NP = decltype(np_tag)::value;
    const int pi = lane % NP;  // the piece this lane stores
    const int g8 = lane - pi;  // first lane of its group
    if (fast) {  // whole window, whole wave: NPC reads, NPC whole-run stores, no lane-dependent control flow
#pragma unroll
      for (int q0 = 0; q0 < NP; q0 += 4) {  // four reads in flight, four stores: 16 registers
        T v[4][VW];
#pragma unroll
        for (int q = 0; q < 4; ++q) load_v<T, VW>(xp + (pi * EM_LANES + ((g8 + q0 + q) ^ pi)) * VW, v[q]);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#if EXCENV_EMR_DEBUG & 4
          asm volatile("" ::"v"(v[q][0]), "v"(v[q][VW - 1]), "v"(lane_off));
#elif EXCENV_EMR_NT
          store_stream<T, VW>(ubase + (q0 + q) * q_stride + lane_off, v[q]);
#else
          store_v<T, VW>(ubase + (q0 + q) * q_stride + lane_off, v[q]);
#endif
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        T v[VW];
        load_v<T, VW>(xp + (pi * EM_LANES + ((g8 + q) ^ pi)) * VW, v);
        T* const p = ubase + q * q_stride + lane_off;
        if (env0 + (int64_t)P * (g8 + q) < ka.B) {
          bool ok[VW], all = true, any = false;
#pragma unroll
          for (int h = 0; h < VW; ++h) {
            const int sl = slot_of(h);
            ok[h] = s_lo <= sl && sl <= s_hi;
            all &= ok[h];
            any |= ok[h];
          }
          if (all) {
            store_v<T, VW>(p, v);
          } else if (any) {
#pragma unroll
            for (int h = 0; h < VW; ++h)
              if (ok[h]) p[h] = v[h];
          }
        }
      }
    }
    wave_sync();
  };
  auto flush = [&](int s_lo, int s_hi, int64_t n_slot0) __attribute__((always_inline)) {
    const bool fast = full_wave && (s_lo == 0) && (s_hi == W - 1);
    const int64_t row_u = env0 * rowlen + n_slot0;                               // (lane 0's environment, slot 0), in rows: uniform
    if (with_states) {
      const int pi = lane % NPC;
      const unsigned lane_rows = (unsigned)((int64_t)P * (lane - pi) * rowlen);  // rows between lane 0's and the group's first environment
#pragma unroll
      for (int j = 0; j < S; ++j) {
        put_pieces([&](int h) __attribute__((always_inline)) { return ring_get(j, h); });
        emit_lines(std::integral_constant<int, NPC>{}, ka.straj[j] + row_u, (int64_t)P * rowlen, lane_rows + (unsigned)(pi * VW), fast, s_lo,
                   s_hi, [&](int h) { return pi * VW + h; });
      }
    }
    // observation lines: line l of the window holds rows [l * RPO, (l + 1) * RPO)
    const int po = lane % NPL;
    const unsigned lane_rows_o = (unsigned)((int64_t)P * (lane - po) * rowlen);
#pragma unroll 1
    for (int l = 0; l < NLO; ++l) {
      if ((l + 1) * RPO - 1 < s_lo || l * RPO > s_hi) continue;  // wave-uniform
      // the rows' observation values go into the transposition buffer piece by piece as they are produced (a whole line of them in
      // registers would cost 32 more)
      auto row = [&](int t) __attribute__((always_inline)) {
        T fs[S], ob[O];
#pragma unroll
        for (int j = 0; j < S; ++j) fs[j] = ring_get(j, l * RPO + t);
        M::observe(fs, c, ob);
        if constexpr (O >= VW) {
#pragma unroll
          for (int m = 0; m < O / VW; ++m) {
            T v[VW];
#pragma unroll
            for (int h = 0; h < VW; ++h) v[h] = ob[m * VW + h];
            const int i = t * (O / VW) + m;
            store_v<T, VW>(xp + (i * EM_LANES + (lane ^ i)) * VW, v);
          }
        } else {  // several rows per piece: element stores (O = 1 or 2 values per row)
          const int i = (t * O) / VW;
#pragma unroll
          for (int q = 0; q < O; ++q) xp[(i * EM_LANES + (lane ^ i)) * VW + (t * O + q) % VW] = ob[q];
        }
      };
      if constexpr (NR <= 2) {  // small models: the rows of a line as straight-line code
#pragma unroll
        for (int t = 0; t < RPO; ++t) row(t);
      } else {  // one row at a time: the rows' chains interleaved by the scheduler cost registers the ring needs
#pragma unroll 1
        for (int t = 0; t < RPO; ++t) row(t);
      }
      wave_sync();
      // element (row s, column o) of the window sits at ((env * rowlen + n_slot0 + s) * O + o); line l starts at s = l * RPO
      emit_lines(std::integral_constant<int, NPL>{}, obs_base + row_u * O + (int64_t)l * WL, (int64_t)P * rowlen * O,
                 lane_rows_o * (unsigned)O + (unsigned)(po * VW), fast, s_lo, s_hi, [&](int h) { return l * RPO + (po * VW + h) / O; });
    }
  };

  // The next action line: prefetched a whole line ahead into 32 registers that stay live through the loop (PMSM fp32: +5 ... 8 %
  // over loading at the crossing, and its 64-byte windows leave the room), or loaded at the crossing, the SIMD's other waves
  // covering the latency (the smaller models: those 32 registers are a third wave per SIMD; cart-pole 0.45 -> 0.50, pendulum
  // 0.51 -> 0.54; PMSM fp64: they are what keeps the kernel from spilling).
  constexpr bool PREFETCH = M::IS_PMSM && sizeof(T) == 4;
  T lineR[PREFETCH ? WL : 1];
  {
    T first[WL];
    load_line(0, first);
    park_line(first);
  }
  int lidx = 0;  // number of the line in the slot
  const int last_line = (pha + (N - 1) * A) / WL;  // the line that holds the lane's last action row (wave-uniform)
  if constexpr (PREFETCH) load_line(last_line < 1 ? last_line : 1, lineR);
  T a_cur[A], sv[S];
  wave_sync();
  read_row(pha, a_cur);

  for (int n = 0; n <= N; ++n) {
    const int slot = (ph + n) % W;
    // Row n + 1 of the actions (clamped) is what this step still needs (row n is in a_cur). When it starts the next line, that
    // line is fetched into the slot (wave-uniform).
    const int k1 = (n + 1 < N) ? n + 1 : N - 1;
    const int pos1 = pha + k1 * A;
    if (!(EXCENV_EMR_DEBUG & 1) && n < N && pos1 / WL != lidx) {
      ++lidx;
      if constexpr (PREFETCH) {
        park_line(lineR);
        load_line(lidx + 1 < last_line ? lidx + 1 : last_line, lineR);  // never past the lane's own rows (a neighbour's line)
      } else {
        T nl[WL];
        load_line(lidx, nl);
        park_line(nl);
      }
      wave_sync();
    }
    T a_nxt[A];  // requested here, used by the integration below: the save in between covers the LDS latency
    read_row(pos1 % WL, a_nxt);
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
    for (int j = 0; j < S; ++j)
      if (j != CL && j != DL) ring[ridx(j)][slot] = sv[j];
    if (!(EXCENV_EMR_DEBUG & 2) && (slot == W - 1 || n == N)) {
      const int back = (n < slot) ? n : slot;  // rows of the window before row n
      flush(slot - back, slot, n - slot);
    }
    if (n < N) {
      if constexpr (AHEAD) {
        env_advance_raw<M, SOLVER>(st, a_cur, a_nxt, n, k1, c, aux);
      } else {
        env_step<M, SOLVER>(st, a_cur, c);
      }
#pragma unroll
      for (int q = 0; q < A; ++q) a_cur[q] = a_nxt[q];
    }
  }
  if (active) {  // row N was saved last: publish last_state from registers
#pragma unroll
    for (int j = 0; j < S; ++j) ka.last_state[j][env] = sv[j];
  }
}

}  // namespace excenv
