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
//     apart, P = the period of that phase in e (a power of two <= W, computed by the host): all 64 environments of a wave cross
//     their run boundaries at the same steps, every ring index and every branch of the flush is wave-uniform.
//   * FLUSH, once per W steps (plus head and tail): per leaf the NPC = W * sizeof(T) / 16 sixteen-byte pieces of a lane's run go
//     through an LDS buffer (64 runs) in which each group of NPC lanes transposes its NPC x NPC pieces, so that a store
//     instruction's adjacent lanes write the adjacent pieces of ONE environment's run: 64 lanes x 16 bytes = 64 / NPC whole runs
//     per instruction, non-temporal. The observation rows of the window (O runs per environment) are evaluated from the ring
//     at flush time (same device function on the same saved state as every other kernel: same bits) and leave the same way.
//     Head / tail windows and ragged waves use the same code with a slot range; pieces cut by the range fall back to element
//     stores.
//   * ACTIONS (round 4). 64-byte windows of every environment's row, fetched by four adjacent lanes of one LDS-direct load
//     (one 64-byte request per environment and window; the scheme of sim_ahead_kernel's AEM instantiations): no registers, no
//     line phase. Rows that are not made of whole 16-byte pieces take the LDS-ring kernel.
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
// LDS bytes per wave: the transposition buffer (64 lanes x 128 bytes) and the action windows (EMR_ANP load instructions' blocks)
constexpr int EMR_ANP = 4;  // 16-byte pieces per action window (64 bytes, fetched by 4 adjacent lanes of one LDS-direct load)
template <class M, typename T, bool AHEAD> constexpr size_t emr_lds_bytes() { return (size_t)EM_LANES * 128 + (size_t)EMR_ANP * AEM_BLOCK_BYTES; }

#ifndef EXCENV_EMR_DEBUG
#define EXCENV_EMR_DEBUG 0  // experiments only (results are wrong): 1 never walk to the next action line, 2 no flush, 4 flush without global stores
#endif
#ifndef EXCENV_EMR_ROW_UNROLL
#define EXCENV_EMR_ROW_UNROLL 1  // observation rows evaluated together at flush time (models with more than two ring leaves)
#endif
#ifndef EXCENV_EMR_DEFER
#define EXCENV_EMR_DEFER 1       // the observation rows of a flush as branch-free blocks of EXCENV_EMR_DEFER_ROWS rows (see flush)
#endif
#ifndef EXCENV_EMR_DEFER_ROWS
#define EXCENV_EMR_DEFER_ROWS 4
#endif
#ifndef EXCENV_EMR_UNROLL_LINES
#define EXCENV_EMR_UNROLL_LINES 0  // 1: the lines of a window unrolled — static ring indices instead of s_set_gpr_idx reads, 4 x the row code
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
  T* const slot_a = xp + EM_LANES * WL;                 // EMR_ANP blocks of AEM_BLOCK_BYTES: the action window of every lane's environment

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

  // ---- actions: 64-byte windows of every environment's row by LDS-direct loads (round 4; the scheme of sim_ahead_kernel's AEM
  // instantiations, kernels.hpp). Lane t of load instruction i fetches piece t % ANP of the window of the environment that lane
  // i * 64 / ANP + t / ANP integrates: ANP adjacent lanes = one 64-byte request per environment and window, landing contiguously
  // in the instruction's 1 KiB block (+16 bytes of bank skew per block for the readers). Rows are 16-byte aligned (host:
  // K * A * sizeof(T) % 16 == 0), so no line phase is involved and the lane spacing P follows from the trajectory rows alone.
  // Round 3 walked 128-byte lines per lane (8 x 16-byte loads of ONE lane per line: FETCH_SIZE 1.5 x the action bytes) and
  // held the next line in 32 registers.
  constexpr int ANP = EMR_ANP, SP = VW / A, ARW = ANP * SP, AEPI = EM_LANES / ANP;
  const unsigned act_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)excenv_emr_smem + (unsigned)(EM_LANES * WL * sizeof(T));
  const unsigned rd_lane = (unsigned)(lane / AEPI) * AEM_BLOCK_BYTES + (unsigned)(lane % AEPI) * (ANP * 16u);
  const int n_pieces = (int)((ka.K * A) / VW);
  int w_hi = 0;  // highest window requested (wave-uniform)
  auto dma_window = [&](int w) __attribute__((always_inline)) {
    int pc = w * ANP + lane % ANP;
    pc = pc < n_pieces ? pc : n_pieces - 1;  // a short last window: the spare lanes fetch the last piece again (never read)
#pragma unroll
    for (int i = 0; i < ANP; ++i) {
      int64_t e = env0 + (int64_t)P * (i * AEPI + lane / ANP);
      e = (e < ka.B) ? e : env0;  // ragged wave: the loaders of absent environments fetch lane 0's row (never read)
      const T* src = ka.actions + e * ka.K * A + (int64_t)pc * VW;
      // inline assembly: see kernels.hpp (the builtin makes the compiler drain vmcnt in front of every LDS read). M0 = LDS
      // address of the block; written and consumed inside this one statement and declared as clobbered (the ring's s_set_gpr_idx
      // sequences also use M0); s_nop 0 = the wait state between an SALU write of M0 and the LDS-direct load.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"  // "reserved register on the clobber list": intended, see kernels.hpp
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(act_lds + (unsigned)i * AEM_BLOCK_BYTES) : "memory", "m0");
#pragma clang diagnostic pop
    }
  };
  auto read_row = [&](int krow, T (&a)[A]) __attribute__((always_inline)) {  // row krow of this lane's environment out of its window
    const unsigned r = (unsigned)krow % ARW, off = (r / SP) * 16u + (r % SP) * (unsigned)(A * sizeof(T));
    load_row<T, A>(reinterpret_cast<const T*>(reinterpret_cast<const unsigned char*>(slot_a) + rd_lane + off), a);
  };

  if (env0 >= ka.B) return;  // no barrier is ever used: a wave without environments may leave
  constexpr int CL = emr_const_leaf<M>();           // this leaf's saved value is st[CL] at every step
  constexpr int DL = emr_derived_leaf<M, AHEAD>();  // this leaf's saved value is a function of other saved leaves
  constexpr int NR = emr_ring_leaves<M, AHEAD>();
  auto in_ring = [](int j) constexpr { return j != CL && j != DL; };
  // index of leaf j among the ring leaves
  auto ridx = [in_ring](int j) constexpr {
    int n = 0;
    for (int q = 0; q < j; ++q) n += in_ring(q) ? 1 : 0;
    return n;
  };
  Vec ring[NR > 0 ? NR : 1];
#pragma unroll
  for (int j = 0; j < NR; ++j) ring[j] = (Vec)(T(0));
  // value of leaf j at slot s_ of the window (j: compile-time constant at every call)
  auto ring_get = [&](int j, int s_) __attribute__((always_inline)) -> T {
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
  auto put_pieces = [&](auto np_tag, auto&& elem) __attribute__((always_inline)) {  // elem(h): element h of this lane's run, produced piece by piece
    constexpr int NPP = decltype(np_tag)::value;
#pragma unroll
    for (int i = 0; i < NPP; ++i) {
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
  // the state leaves (W steps, NPC pieces per run), slots [s_lo, s_hi]
  auto flush_leaves = [&](auto np_tag, int s_lo, int s_hi, int64_t n_slot0) __attribute__((always_inline)) {
    constexpr int NPG = decltype(np_tag)::value;
    const bool fast = full_wave && (s_lo == 0) && (s_hi == NPG * VW - 1);
    const int64_t row_u = env0 * rowlen + n_slot0;
    const int pi = lane % NPG;
    const unsigned lane_rows = (unsigned)((int64_t)P * (lane - pi) * rowlen);  // rows between lane 0's and the group's first environment
#pragma unroll
    for (int j = 0; j < S; ++j) {
      put_pieces(np_tag, [&](int h) __attribute__((always_inline)) { return ring_get(j, h); });
      emit_lines(np_tag, ka.straj[j] + row_u, (int64_t)P * rowlen, lane_rows + (unsigned)(pi * VW), fast, s_lo, s_hi,
                 [&](int h) { return pi * VW + h; });
    }
  };
  auto flush = [&](int s_lo, int s_hi, int64_t n_slot0) __attribute__((always_inline)) {
    const bool fast = full_wave && (s_lo == 0) && (s_hi == W - 1);
    const int64_t row_u = env0 * rowlen + n_slot0;                               // (lane 0's environment, slot 0), in rows: uniform
    if (with_states) flush_leaves(std::integral_constant<int, NPC>{}, s_lo, s_hi, n_slot0);
    // observation lines: line l of the window holds rows [l * RPO, (l + 1) * RPO)
    const int po = lane % NPL;
    const unsigned lane_rows_o = (unsigned)((int64_t)P * (lane - po) * rowlen);
#if EXCENV_EMR_UNROLL_LINES
#pragma unroll
#else
#pragma unroll 1
#endif
    for (int l = 0; l < NLO; ++l) {
      if ((l + 1) * RPO - 1 < s_lo || l * RPO > s_hi) continue;  // wave-uniform
      // the rows' observation values go into the transposition buffer piece by piece as they are produced (a whole line of them in
      // registers would cost 32 more)
      auto put_row = [&](int t, const T (&ob)[O]) __attribute__((always_inline)) {
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
      auto row_state = [&](int t, T (&fs)[S]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < S; ++j) fs[j] = ring_get(j, l * RPO + t);
      };
      auto row = [&](int t) __attribute__((always_inline)) {
        T fs[S], ob[O];
        row_state(t, fs);
        M::observe(fs, c, ob);
        put_row(t, ob);
      };
      if constexpr (EXCENV_EMR_DEFER && observe_defer_ok<M, T>()) {
        // Round 5: G rows as ONE straight-line block. A wave of this kernel carries one environment per lane, so a row's
        // generate_observation is one dependent chain (sin / cos -> ..., six normalisations) and the guards inside sincos_t /
        // InvDiv::div end basic blocks: unrolled rows could not interleave (round 4: two / four rows "together" measured flat).
        // observe_defer has no branch; its guards are tested once per block, and a block that fails (|angle| > 65536, a quotient
        // outside the moderate range, NaN / inf) is redone with the guarded M::observe — same bits either way.
        // (rows per block: four, two where the ring already holds 96 registers — PMSM's six leaves on the step-semantics path —
        // so that nothing spills)
        constexpr int GW = (NR * W * (int)sizeof(T) / 4 >= 96) ? 2 : EXCENV_EMR_DEFER_ROWS;
        constexpr int G = RPO < GW ? RPO : GW;
#pragma unroll
        for (int t0 = 0; t0 < RPO; t0 += G) {
          T obv[G][O];
          bool bad = false;
#pragma unroll
          for (int g = 0; g < G; ++g) {
            T fs[S];
            row_state(t0 + g, fs);
            observe_defer<M, T>(fs, c, obv[g], bad);
          }
          if (__builtin_expect(__builtin_amdgcn_ballot_w64(bad) != 0, 0)) {
#pragma unroll 1
            for (int g = 0; g < G; ++g) {
              T fs[S], ob[O];
              row_state(t0 + g, fs);
              M::observe(fs, c, ob);
#pragma unroll
              for (int gg = 0; gg < G; ++gg)
                if (gg == g) {
#pragma unroll
                  for (int q = 0; q < O; ++q) obv[gg][q] = ob[q];
                }
            }
          }
#pragma unroll
          for (int g = 0; g < G; ++g) put_row(t0 + g, obv[g]);
        }
      } else if constexpr (NR <= 2 || EXCENV_EMR_ROW_UNROLL >= RPO) {  // the rows of a line as straight-line code
#pragma unroll
        for (int t = 0; t < RPO; ++t) row(t);
      } else if constexpr (EXCENV_EMR_ROW_UNROLL == 2 && RPO % 2 == 0) {  // two rows' chains interleaved by the scheduler
#pragma unroll 1
        for (int t = 0; t < RPO; t += 2) {
          row(t);
          row(t + 1);
        }
      } else {  // one row at a time: interleaved chains cost registers the ring needs
#pragma unroll 1
        for (int t = 0; t < RPO; ++t) row(t);
      }
      wave_sync();
      // element (row s, column o) of the window sits at ((env * rowlen + n_slot0 + s) * O + o); line l starts at s = l * RPO
      emit_lines(std::integral_constant<int, NPL>{}, obs_base + row_u * O + (int64_t)l * WL, (int64_t)P * rowlen * O,
                 lane_rows_o * (unsigned)O + (unsigned)(po * VW), fast, s_lo, s_hi, [&](int h) { return l * RPO + (po * VW + h) / O; });
    }
  };

  dma_window(0);
  T a_cur[A], sv[S];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // once per trajectory: the first window (and the initial state) is there
  read_row(0, a_cur);

  for (int n = 0; n <= N; ++n) {
    const int slot = (ph + n) % W;
    // Row n + 1 of the actions (clamped) is what this step still needs (row n is in a_cur). Its window was requested during the
    // previous step, behind that step's flush and in front of its integration: nothing younger is in flight, so waiting for
    // everything outstanding waits for exactly that fill (and for older stores, which retire before it anyway).
    const int k1 = (n + 1 < N) ? n + 1 : N - 1;
    if (!(EXCENV_EMR_DEBUG & 1) && k1 % ARW == 0 && k1 / ARW == w_hi && w_hi > 0) asm volatile("s_waitcnt vmcnt(0) expcnt(6)" ::: "memory");  // expcnt(6): never blocks, marks the hand-written wait (tools/isa_guards.py)
    T a_nxt[A];  // requested here, used by the integration below: the save in between covers the LDS latency
    read_row(k1, a_nxt);
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
    for (int j = 0; j < S; ++j) {
      if (j == CL || j == DL) continue;
      ring[ridx(j)][slot] = sv[j];
    }
    if (!(EXCENV_EMR_DEBUG & 2) && (slot == W - 1 || n == N)) {
      const int back = (n < slot) ? n : slot;  // rows of the window before row n
      flush(slot - back, slot, n - slot);
    }
    // row k1 was the last of its action window: the window's LDS is dead (its reads have returned by now; made formal) and
    // takes the next one — behind the flush, so that the wait at the top of the next step does not drain this step's stores
    if (!(EXCENV_EMR_DEBUG & 1) && n < N && k1 % ARW == ARW - 1 && k1 / ARW == w_hi && (w_hi + 1) * ANP < n_pieces) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      ++w_hi;
      dma_window(w_hi);
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
