"""Output memory of the hot path: the slot pool behind `vmap_step` and the pooled, placed output sets behind large `vmap_sim_ahead`
calls. Nothing here is part of the reference's API (`core_env.py` is the mirror of that); it exists because (a) at RL batch sizes the
host side decides the step rate and creating fresh tensor objects per call was its largest item, and (b) where tens of GB of
trajectory buffers lie in physical device memory moves the trajectory kernel by 15 - 20 %.

Interface used by `CoreEnvironment`:
    StepSlotPool(env):          take(gym, stream, capturing) -> (slots, i)
    TrajectoryPlacement(env):   acquire(shape ...) -> TrajSet     timed_launch(set, fn, nbytes)     note_launch(set)
                                settled     release()     wait_stream(stream) / drain_waits()     memory_budget(...)
Both hand a buffer out again ONLY when nothing outside can observe it: no Python reference to any of its tensors (sys.getrefcount
back at the value recorded when the buffer was made), no C++ holder (Tensor._use_count() == 1: autograd, DLPack, a view keeps its
base), no foreign view of the storage (storage use count back at its recorded value) and the same stream as before (the kernels
that read the buffer are then ordered before the one that overwrites it). The functional contract of the reference's API holds.
"""
from __future__ import annotations

import os
import sys

import torch

from . import _native

_storage_use_count = getattr(torch._C, "_storage_Use_Count", None)
_tensor_use_count = getattr(torch.Tensor, "_use_count", None)


def liveness_available() -> bool:
    """This torch build can tell whether a tensor is still visible to anyone (else nothing is ever recycled)."""
    return _storage_use_count is not None and _tensor_use_count is not None


# ---------------------------------------------------------------------------------------------------------------------------------
# vmap_step: output slots
# ---------------------------------------------------------------------------------------------------------------------------------
# The reference's HOT LOOP #1 is a Python loop of vmap_step calls (README.md:28-32); at RL batch sizes the kernel takes a few
# microseconds, so the host side decides the rate. Outputs are fresh memory every call (functional contract): slots are carved from
# one allocation per `n` calls. Once a pool is used up its oldest slot is handed out again only if it is dead (see the module
# docstring). In `obs, state = env.vmap_step(state, act)` the outputs of two calls ago are dead, so the loop runs on recycled tensor
# objects: creating the S + 1 views per call was the largest item of the host time. Anything still referenced makes the test fail
# and a new pool is allocated.
class Slots:
    __slots__ = ("n", "i", "leaves", "obs", "out_ptrs", "obs_ptrs", "gym", "gym_ptrs", "phys", "objs", "tens", "rc0",
                 "storages", "use0", "stream", "obs_width")


class StepSlotPool:
    SPLIT_POOL_BYTES = 256 << 20

    def __init__(self, env):
        self.env = env
        self.slots = {False: None, True: None}

    def new_slots(self, n: int, gym: bool) -> Slots:
        env = self.env
        B, S, O = env.batch_size, env.physical_state_dim, env._obs_dim()
        isz = torch.empty((), dtype=env.dtype).element_size()
        al = 16 // isz
        Bp = (B + al - 1) // al * al
        obs_elems = (B * O + al - 1) // al * al
        rew_elems = Bp if gym else 0
        slot = S * Bp + obs_elems + rew_elems
        # Large pools: the observations get their own allocation, so that a caller who keeps only observations (a rollout
        # buffer) does not pin the state leaves of the pool as well. (Placing the two a region apart, DESIGN.md §6.1, was
        # measured and does nothing for this short streaming kernel.) Small pools stay one allocation (host time).
        split = n * slot * isz >= self.SPLIT_POOL_BYTES
        if split:
            slot -= obs_elems
        buf = torch.empty(n * slot, dtype=env.dtype, device=env.device)
        base = buf.data_ptr()
        sl = Slots()
        sl.n, sl.i = n, 0
        sl.leaves = [t.unbind(0) for t in buf.as_strided((n, S, B), (slot, Bp, 1)).unbind(0)]
        obuf = None
        if split:
            obuf = torch.empty(n * obs_elems, dtype=env.dtype, device=env.device)
            obase = obuf.data_ptr()
            sl.obs = obuf.as_strided((n, B, O), (obs_elems, O, 1)).unbind(0)
            sl.obs_ptrs = [obase + i * obs_elems * isz for i in range(n)]
            obs_elems = 0  # the reward column (gym) follows the leaves directly
        else:
            sl.obs = buf.as_strided((n, B, O), (slot, O, 1), S * Bp).unbind(0)
            sl.obs_ptrs = [base + (i * slot + S * Bp) * isz for i in range(n)]
        sl.out_ptrs = [_native.ptr_array([base + (i * slot + j * Bp) * isz for j in range(S)]) for i in range(n)]
        sl.gym = sl.gym_ptrs = None
        flags = None
        if gym:
            TW = _native.truncated_width(env.ENV_ID, len(env.control_state))
            flags = torch.empty((n, B * (1 + TW)), dtype=torch.bool, device=env.device)
            fbase = flags.data_ptr()
            rew = buf.as_strided((n, B, 1), (slot, 1, 1), S * Bp + obs_elems).unbind(0)
            term = flags.as_strided((n, B, 1), (B * (1 + TW), 1, 1)).unbind(0)
            trunc = flags.as_strided((n, B, TW), (B * (1 + TW), TW, 1), B).unbind(0)
            sl.gym = list(zip(rew, term, trunc))
            sl.gym_ptrs = [(base + (i * slot + S * Bp + obs_elems) * isz, fbase + i * B * (1 + TW),
                            fbase + i * B * (1 + TW) + B) for i in range(n)]
        sl.obs_width = O
        sl.phys = [env.PhysicalState(*lv) for lv in sl.leaves]
        sl.tens = [tuple(sl.leaves[i]) + (sl.obs[i],) + (tuple(sl.gym[i]) if gym else ()) for i in range(n)]
        sl.objs = [sl.tens[i] + (sl.phys[i],) for i in range(n)]
        sl.storages = [buf.untyped_storage()] + ([obuf.untyped_storage()] if obuf is not None else []) + ([flags.untyped_storage()] if gym else [])
        sl.rc0 = sl.use0 = sl.stream = None
        return sl

    def is_free(self, sl: Slots, i: int, stream) -> bool:
        if sl.rc0 is None or sl.stream != stream:
            return False
        if tuple(map(sys.getrefcount, sl.objs[i])) != sl.rc0[i]:
            return False
        tens = sl.tens[i]
        if sum(map(_tensor_use_count, tens)) != len(tens):
            return False
        return [_storage_use_count(st._cdata) for st in sl.storages] == sl.use0

    def arm(self, sl: Slots, stream):
        """Record the reference counts of a fresh pool (nothing outside `sl` refers to its tensors yet)."""
        if not liveness_available():
            return  # this torch build cannot tell whether a slot is still visible: never recycle
        sl.stream = stream
        sl.rc0 = [tuple(map(sys.getrefcount, o)) for o in sl.objs]
        sl.use0 = [_storage_use_count(st._cdata) for st in sl.storages]

    def per_alloc(self, gym: bool) -> int:
        """Slots per pool: ~4 MiB worth, at most 32; never fewer than 3 (the input state, the output and one dead slot —
        below that nothing can ever be recycled) unless three slots would exceed 1 GiB."""
        env = self.env
        isz = 4 if env.dtype == torch.float32 else 8
        per = env.batch_size * (env.physical_state_dim + env._obs_dim() + (1 if gym else 0)) * isz
        n = max(1, min(32, (4 << 20) // max(per, 1)))
        if n < 3 and 3 * per <= (1 << 30):
            n = 3
        return n

    def take(self, gym: bool, stream, capturing: bool, obs_width: int):
        """(slots, index) of the output slot of this call: the next one of the current pool, its oldest one again when that is dead,
        else the first of a new pool. Memory allocated during graph capture belongs to the graph's pool: never handed out later."""
        sl = None if capturing else self.slots[gym]
        i = 0
        if sl is not None:
            i = sl.i
            if i >= sl.n:
                i %= sl.n
                if not self.is_free(sl, i, stream):
                    sl = None
            if sl is not None and sl.obs_width != obs_width:
                sl = None
        if sl is None:
            sl = self.new_slots(1 if capturing else self.per_alloc(gym), gym)
            i = 0
            if not capturing:
                self.slots[gym] = sl
                self.arm(sl, stream)
        sl.i += 1
        return sl, i


# ---------------------------------------------------------------------------------------------------------------------------------
# vmap_sim_ahead: large output sets
# ---------------------------------------------------------------------------------------------------------------------------------
# Device memory behaves as a few large physical regions (profiles/r03_placement_regions.md): write traffic that falls into ONE region
# at a time — observations and state leaves allocated back to back by a fresh process — runs at ~5.0 TB/s where the same kernel over
# buffers in two regions runs at ~5.9 TB/s; a plain sequential fill shows the same two levels, so this is the platform, not the
# kernel. Virtual addresses say nothing about the region, so a new set of output buffers is placed by MEASUREMENT and then pooled:
#   * the first two sets of a shape are one arena [observations A | observations B | states A | states B] (the other set's
#     observations are the distance between a set's two kinds of write streams), judged by the absolute criterion below; no probe
#     launches of the trajectory kernel, no spacers;
#   * otherwise a SEARCH: candidate state blocks are judged one after the other while the rejected blocks and a spacer (hipMalloc
#     outside torch's cache) stay allocated, so that the next candidate lands a region further on;
#   * the judge is ABSOLUTE where the launch's access pattern can be replayed without arithmetic (excenv_stream_pattern over the
#     same buffers: it follows the trajectory kernel's time with a correlation of -0.92 across placements, tools/placement_classify.py):
#     pattern rate / fill rate >= PATTERN_ACCEPT (fast level 0.82 ... 0.84, slow placements 0.70 ... 0.79). Sets accepted that way are
#     final. Sets that cannot be judged that way (row-major actions or trajectories) are judged by their REAL launches: every
#     large launch into a pooled set is bracketed by two HIP events, read when the set comes round again; a set that then runs
#     > REPLACE_RATIO slower than its sibling is replaced by a searched one — which must beat it, or the old set stays.
#   * THE FIRST REAL LAUNCH INTO A SET IS NEVER A JUDGEMENT (round 5): it carries clock ramp-up, first-touch page mapping and TLB
#     fills (the driver's round-4 bench run: 10.4 ms, then 4.9 / 4.9) and is recorded separately (`first_ms`). A set has a
#     `steady_ms` from its second timed launch on.
class TrajSet:
    __slots__ = ("key", "obs_buf", "st_buf", "lbuf", "observations", "st_views", "last", "obs_ptr", "traj_ptrs", "last_ptrs",
                 "tens", "storages", "rc0", "use0", "stream", "placement", "ev", "ev_pending", "steady_ms", "first_ms", "uses")

    def __init__(self, key=None):
        self.key = key
        self.ev, self.ev_pending, self.steady_ms, self.first_ms, self.uses = None, False, None, None, 0
        self.placement = None
        self.rc0 = self.use0 = self.stream = None

    def record_ms(self, ms: float):
        """Time of one real launch into this set. The first one is kept apart (cold clocks / first touch): never a judgement."""
        if self.first_ms is None:
            self.first_ms = ms
            return
        self.uses += 1
        self.steady_ms = ms if self.steady_ms is None else min(self.steady_ms, ms)

    @property
    def judged_by_pattern(self) -> bool:
        return bool(self.placement) and self.placement.get("pattern_over_fill") is not None and \
            self.placement["pattern_over_fill"] >= self.placement.get("accept_at", 2.0)


class TrajectoryPlacement:
    PLACED_BYTES = 1 << 30        # output sets at least this large are placed (smaller ones: plain allocations)
    TRIES = 4                     # candidate state blocks per search (two more for blocks <= 10 GiB)
    ACCEPT = 0.93                 # relative judge: fastest / slowest candidate at or below this -> both levels have been seen
    POOL_SETS = 2
    REPLACEMENTS = 2              # searched replacements per shape, at most
    REPLACE_RATIO = 1.05          # a set this much slower than its sibling in real (steady) launches is up for replacement
    DECIDE_USES = 3               # ... while it has at most this many steady timings: afterwards it stays (no search in a long run)
    SPACER_BYTES = 16 << 30       # a rejected block + this much memory stay allocated while the next block is made
    # absolute judge: pattern rate / fill rate. Fast level 0.82 ... 0.86, slow placements 0.70 ... 0.79 (tools/placement_classify.py);
    # 0.81 (round 4) let an arena set through at 0.8158 that then ran 6 % behind its sibling (5.10 / 4.80 ms, round 5's bench run)
    PATTERN_ACCEPT = 0.825
    QUAD_MIN_DISTANCE = 17 << 30  # arena pair: observations -> states of one set at least this far apart
    QUAD_MIN_SET_BYTES = 4 << 30  # smaller sets keep the search (an artificial gap measured 0.57 for C2)
    # Row-major (reference-shaped) sets do NOT take the arena by default (round 5): the access-pattern replay says nothing about
    # the register-ring kernel's writes, so an arena pair has no absolute criterion, and the relative one (a set 5 % slower than its
    # sibling is replaced) cannot see a pair that is slow as a whole — a default `bench.py --traj-layout env_major` run sat at
    # 8.5 / 8.9 ms per launch on such a pair where searched sets run 6.2 ... 6.7 (five runs). EXCENV_EM_ARENA=1 brings it back.
    QUAD_ENV_MAJOR = os.environ.get("EXCENV_EM_ARENA", "0") == "1"

    def __init__(self, env):
        self.env = env
        self.sets = []
        self.best = {}       # relative judge: best probe time known per shape
        self.replaced = {}   # searched replacements made per set key
        self.target = None   # sibling's steady time a replacement has to match
        self.quad_made = set()
        self.fill_gbs = None
        self.wait_events = []
        self.last = None     # diagnostics of the most recent placement decision

    # -- switches that live on the environment (public knobs) -------------------------------------------------------------------
    @property
    def mode(self) -> str:
        return self.env.trajectory_placement

    @property
    def active(self) -> bool:
        return self.mode in ("auto", "search")

    # -- capacity planning ---------------------------------------------------------------------------------------------------
    @classmethod
    def memory_budget(cls, B: int, rows: int, OW: int, S: int, itemsize: int, free_bytes: int) -> dict:
        """Upper bounds (bytes) of what pooling and placing the large output sets of one shape can hold on a device with
        `free_bytes` free (C5: 2^22 environments per GPU, 101 rows: 25.5 GB per set):
          steady      : the pooled sets that stay allocated (POOL_SETS sets: observations + state block + last states)
          search_peak : the most a placement search holds at once on top of the OTHER pooled set — the new set's observations,
                        up to TRIES candidate state blocks and the spacers between them (each at most a third of what is free
                        when it is taken, never more than SPACER_BYTES)
        A search that runs out of memory stops early and keeps the best candidate seen (torch.OutOfMemoryError is caught)."""
        obs = rows * OW * B * itemsize
        block = S * rows * B * itemsize
        one = obs + block + S * B * itemsize
        spacer = min(cls.SPACER_BYTES, max(free_bytes // 3, 0))
        peak = obs + cls.TRIES * block + (cls.TRIES - 1) * spacer + spacer  # + the observation spacer of a replacement
        return {"set": one, "steady": cls.POOL_SETS * one, "search_peak": 2 * one + peak,
                "searches_at_most": 1 + cls.REPLACEMENTS + (cls.POOL_SETS - 1)}

    def release(self):
        """Drop the pooled (dead) sets so that their memory returns to torch's allocator; the next large call starts over."""
        self.sets = []
        self.quad_made = set()

    # -- foreign streams -------------------------------------------------------------------------------------------------------
    def wait_stream(self, stream=None):
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.env.device) if stream is None else stream)
        self.wait_events.append(ev)

    def drain_waits(self):
        if self.wait_events:
            cur = torch.cuda.current_stream(self.env.device)
            for ev in self.wait_events:
                cur.wait_event(ev)
            self.wait_events = []

    # -- real-launch timing ----------------------------------------------------------------------------------------------------
    def note_launch(self, ts: TrajSet):
        """Read the HIP events of the previous real launch into a pooled set (finished long ago when the set comes round again)."""
        if ts.ev_pending and ts.ev is not None and ts.ev[1].query():
            ts.ev_pending = False
            ts.record_ms(float(ts.ev[0].elapsed_time(ts.ev[1])))

    def timed_launch(self, ts: TrajSet, launch_fn, nbytes: int):
        """Launch into a pooled, placed set with a pair of HIP events around it (two event records per multi-millisecond launch)."""
        timed = (ts.rc0 is not None and self.active and nbytes >= self.PLACED_BYTES and not torch.cuda.is_current_stream_capturing())
        if not timed:
            launch_fn()
            return
        self.note_launch(ts)
        if ts.ev is None:
            ts.ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        if ts.ev_pending:  # the previous launch has not finished yet (back-to-back reuse through out=...): leave its events alone
            launch_fn()
            return
        ts.ev[0].record()
        launch_fn()
        ts.ev[1].record()
        ts.ev_pending = True

    @staticmethod
    def _set_bytes(key) -> int:
        B, rows, OW, S, _, dt = key[:6]
        return rows * (OW + S) * B * (4 if dt is torch.float32 else 8)

    def replacement_due(self, ts: TrajSet, siblings) -> bool:
        """A dead set is up for replacement when its STEADY real launches run clearly slower than a sibling's — decided early (never in
        the middle of a long run), at most REPLACEMENTS times per shape, never for sets the absolute criterion accepted and never on
        a first launch (record_ms keeps that one apart)."""
        if not self.active or ts.judged_by_pattern or self._set_bytes(ts.key) < (1 << 30):
            return False
        if self.replaced.get(ts.key, 0) >= self.REPLACEMENTS or ts.uses > self.DECIDE_USES:
            return False
        sib = [t.steady_ms for t in siblings if t.steady_ms is not None]
        best = min(sib) if sib else self.best.get(self._pkey(ts.key))
        ms = ts.steady_ms if sib else (ts.placement or {}).get("chosen_ms")
        return best is not None and ms is not None and ms > self.REPLACE_RATIO * best

    @staticmethod
    def _pkey(key):
        return tuple(key[:4]) + tuple(key[6:])

    @property
    def settled(self) -> bool:
        """True once no later call can run a placement search: every pooled shape has its POOL_SETS sets, and each set is either
        accepted by the absolute criterion (final at once) or past its decision window in real launches (DECIDE_USES steady timings;
        or the shape's replacements are used up)."""
        if not self.sets or not self.active or not self.env.trajectory_pool:
            return True
        for ts in self.sets:
            self.note_launch(ts)
        by_key = {}
        for ts in self.sets:
            by_key.setdefault(ts.key, []).append(ts)
        for key, sets in by_key.items():
            if len(sets) < self.POOL_SETS:
                return False
            if self._set_bytes(key) < (1 << 30) or self.replaced.get(key, 0) >= self.REPLACEMENTS:
                continue
            for t in sets:
                if t.judged_by_pattern:
                    continue
                # judged by real launches: a set can be replaced while it has at most DECIDE_USES steady timings — until every such
                # set is past that window a later call may still run a search (the running minimum of a sibling can still move the
                # comparison: round 5, a search landed in a timed region 12 calls after `settled` had said True)
                if t.steady_ms is None or t.uses <= self.DECIDE_USES:
                    return False
        return True

    # -- judges ------------------------------------------------------------------------------------------------------------------
    def fill_rate(self, buf) -> float:
        """GB/s of a plain fill of `buf` (once per environment: it does not depend on where the buffer lies)."""
        if self.fill_gbs is None:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            buf.fill_(0)
            for e in ev[:-1]:
                e.record()
                buf.fill_(0)
            ev[-1].record()
            ev[-1].synchronize()
            ms = min(float(a.elapsed_time(b)) for a, b in zip(ev[:-1], ev[1:]))
            self.fill_gbs = buf.numel() * buf.element_size() / ms / 1e6
        return self.fill_gbs

    def pattern_score(self, obs_buf, block_ptr, leaf_e, B, rows, OW, S, isz, act_ptr, A):
        """(ms, pattern rate / fill rate) of the trajectory launch's access pattern over (obs_buf, the state block at block_ptr)."""
        dev = self.env.device
        rb = B * isz
        ob = obs_buf.data_ptr()
        wr = [ob + c * rb for c in range(OW)] + [block_ptr + j * leaf_e * isz for j in range(S)]
        wrs = [OW * rb] * OW + [rb] * S
        rd, rds = [act_ptr + c * rb for c in range(A)], [A * rb] * A
        R = rows - 2
        stream = _native._raw_stream(dev)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        with _native._on_device(dev):
            _native.stream_pattern(rd, rds, wr, wrs, rb, R, stream)
            for e in ev[:-1]:
                e.record()
                _native.stream_pattern(rd, rds, wr, wrs, rb, R, stream)
            ev[-1].record()
        ev[-1].synchronize()
        ms = min(float(a.elapsed_time(b)) for a, b in zip(ev[:-1], ev[1:]))
        return ms, ((A + OW + S) * rb * R / ms / 1e6) / self.fill_rate(obs_buf)

    # -- search ------------------------------------------------------------------------------------------------------------------
    def place_state_block(self, obs_buf, B, rows, OW, S, isz, time_launch, block_shape=None, known_ms=None, pattern=None):
        """A [S, rows, B] block for the state leaves of a new set whose traffic, together with the observations', does not fall into
        one physical region. `pattern(block)` -> (ms, pattern / fill) judges absolutely; else `time_launch(block)` runs the
        trajectory launch of the current call into (obs_buf, block) and returns its time in ms. Returns (block, diagnostics)."""
        dt, dev = self.env.dtype, self.env.device
        block_shape = (S, rows, B) if block_shape is None else block_shape  # env-major sets: (S, padded leaf elements)
        block = torch.empty(block_shape, dtype=dt, device=dev)
        nbytes = (OW + S) * rows * B * isz
        if (not self.active or (time_launch is None and pattern is None) or nbytes < self.PLACED_BYTES
                or torch.cuda.is_current_stream_capturing()):
            return block, None
        pkey = (B, rows, OW, S) + (() if len(block_shape) == 3 else ("env_major",))
        known = self.best.get(pkey) if known_ms is None else known_ms  # known_ms: a sibling set's steady-state time
        tried, spacers = [], []
        try:
            # smaller blocks are cheap to probe and their first candidates land in the slow level more often (C2: all four in
            # one of two fresh processes): two more tries
            tries = self.TRIES + (2 if S * rows * B * isz <= (10 << 30) else 0)
            ratios = []
            for k in range(tries):
                if pattern is not None:
                    t, ratio = pattern(block)
                    ratios.append(round(ratio, 4))
                else:
                    t = time_launch(block)
                tried.append((t, block))
                times = [x for x, _ in tried]
                if pattern is not None:
                    good = ratio >= self.PATTERN_ACCEPT
                elif known is not None:
                    good = t <= 1.02 * known
                else:
                    good = len(times) >= 2 and min(times) <= self.ACCEPT * max(times)
                if good or k == tries - 1:
                    break
                # Blocks torch holds in its cache (an earlier set's rejected candidates, for one) would be handed out again at
                # their old addresses whatever the spacer does: they go back to the driver first (once per search).
                if k == 0:
                    torch.cuda.empty_cache()
                # the spacer never takes more than a third of what the device has free right now (other processes may share it)
                free_b = torch.cuda.mem_get_info(dev)[0]
                want_b = req_b = max(self.SPACER_BYTES - S * rows * B * isz, 1 << 20)
                if free_b < 3 * (want_b + S * rows * B * isz):
                    want_b = max(0, free_b // 3 - S * rows * B * isz)
                if want_b < min(req_b, 1 << 30):
                    break  # not enough room to move the next candidate a region further: keep the best seen so far
                with _native._on_device(dev):
                    sp = _native.raw_malloc(want_b)
                if sp is not None:
                    spacers.append(sp)
                try:
                    block = torch.empty(block_shape, dtype=dt, device=dev)
                except torch.OutOfMemoryError:
                    break
        finally:
            for sp in spacers:
                _native.raw_free(sp)
        t_best, best = min(tried, key=lambda tb: tb[0])
        chosen = [t for t, _ in tried].index(t_best)
        if pattern is not None:
            diag = {"candidate_pattern_ms": [round(t, 4) for t, _ in tried], "candidate_pattern_over_fill": ratios, "chosen": chosen,
                    "pattern_over_fill": ratios[chosen], "accept_at": self.PATTERN_ACCEPT, "fill_gbs": round(self.fill_gbs, 1),
                    "what": "no-arithmetic access pattern of the launch (excenv_stream_pattern) timed over (observations, candidate "
                            "state block) against the fill rate; rejected blocks and a spacer stay allocated while the next "
                            "candidate is made"}
        else:
            self.best[pkey] = t_best if known is None else min(known, t_best)
            diag = {"candidate_ms": [round(t, 4) for t, _ in tried], "chosen": chosen, "chosen_ms": t_best,
                    "best_known_ms_before": known, "spacer_gib": self.SPACER_BYTES / 2**30,
                    "what": "trajectory launch of the call timed into (observations, candidate state block); rejected blocks and a "
                            "spacer stay allocated while the next candidate is made"}
        del tried, block
        return best, diag

    # -- liveness ----------------------------------------------------------------------------------------------------------------
    @staticmethod
    def set_is_free(ts: TrajSet, stream) -> bool:
        if ts.rc0 is None or ts.stream != stream:
            return False
        if tuple(map(sys.getrefcount, ts.tens)) != ts.rc0:
            return False
        if sum(map(_tensor_use_count, ts.tens)) != len(ts.tens):
            return False
        return [_storage_use_count(st._cdata) for st in ts.storages] == ts.use0

    @staticmethod
    def _arm(ts: TrajSet, stream):
        ts.stream = stream
        ts.rc0 = tuple(map(sys.getrefcount, ts.tens))
        ts.use0 = [_storage_use_count(st._cdata) for st in ts.storages]

    def _finish_views(self, ts: TrajSet, B, rows, S, leaf_e, last_e, isz, env_major, want_states):
        """Pointer arrays and the returned views of a set whose obs_buf / st_buf / lbuf exist."""
        lb = ts.lbuf.data_ptr()
        ts.last_ptrs = _native.ptr_array([lb + j * last_e * isz for j in range(S)])
        ts.obs_ptr = ts.obs_buf.data_ptr()
        if want_states:
            sb = ts.st_buf.data_ptr()
            strides = (leaf_e, rows, 1) if env_major else (rows * B, 1, B)
            ts.st_views = tuple(ts.st_buf.as_strided((S, B, rows), strides).unbind(0))
            ts.traj_ptrs = _native.ptr_array([sb + j * leaf_e * isz for j in range(S)])
        else:
            ts.st_buf, ts.st_views, ts.traj_ptrs = None, None, None
        ts.observations = ts.obs_buf[:] if env_major else ts.obs_buf.permute(2, 0, 1)  # a view object of its own (liveness test)
        ts.last = tuple(ts.lbuf[:, :B].unbind(0))
        ts.tens = (ts.observations,) + (ts.st_views or ()) + ts.last
        ts.storages = [t.untyped_storage() for t in ((ts.obs_buf, ts.lbuf) + ((ts.st_buf,) if want_states else ()))]

    # -- acquire -----------------------------------------------------------------------------------------------------------------
    def acquire(self, B, rows, OW, S, want_states, last_e, isz, launch, env_major=False, pattern_ctx=None) -> TrajSet:
        """The output set of one large call: a dead pooled set of this shape, else a newly placed one. `launch(obs_ptr, traj_ptrs,
        last_ptrs)` enqueues the call's trajectory launch into the given buffers (the relative judge times it); `pattern_ctx` =
        (actions pointer, A) when the launch's access pattern can be replayed without arithmetic (lane-major actions).
        env_major: the reference's row-major arrays (observations [B, rows, OW], state leaves [B, rows], every leaf starting on a
        128-byte boundary of one block) instead of views of lane-major memory; pooled and placed the same way."""
        env = self.env
        dt, dev = env.dtype, env.device
        key = (B, rows, OW, S, want_states, dt) + (("env_major",) if env_major else ())
        leaf_e = (rows * B * isz + 127) // 128 * 128 // isz if env_major else rows * B  # elements between consecutive leaves
        capturing = torch.cuda.is_current_stream_capturing()
        pooled = env.trajectory_pool and not capturing and liveness_available()
        stream = _native._raw_stream(dev)
        replacing = None
        if pooled:
            for k, ts in enumerate(self.sets):
                if ts.key == key and self.set_is_free(ts, stream):
                    self.note_launch(ts)
                    sibs = [t for t in self.sets if t is not ts and t.key == key]
                    if launch is not None and self.replacement_due(ts, sibs):
                        self.replaced[key] = self.replaced.get(key, 0) + 1
                        sib_ms = [t.steady_ms for t in sibs if t.steady_ms is not None]
                        self.target = min(sib_ms) if sib_ms else None
                        replacing = self.sets.pop(k)  # stays alive until the new set has shown that it is faster
                        break
                    self.sets.append(self.sets.pop(k))  # most recently used last
                    return ts
            self.sets = [t for t in self.sets if t.key == key][-(self.POOL_SETS - 1):] if self.POOL_SETS > 1 else []
        # the arena pair only where something can judge it as a whole: the access-pattern replay (lane-major sets of calls whose
        # pattern it can replay), or on request for row-major sets (see QUAD_ENV_MAJOR); everything else is searched
        if (pooled and want_states and ((not env_major and pattern_ctx is not None) or (env_major and self.QUAD_ENV_MAJOR))
                and self.mode == "auto" and self.target is None
                and replacing is None and self.POOL_SETS == 2 and not self.sets and key not in self.quad_made
                and (OW + S) * rows * B * isz >= max(self.PLACED_BYTES, self.QUAD_MIN_SET_BYTES)):
            first = self._ordered_pair(key, B, rows, OW, S, last_e, isz, stream, env_major, pattern_ctx)
            if first is not None:
                return first
        ts = TrajSet(key)
        known_ms, self.target = self.target, None
        obs_spacer = None
        if replacing is not None:
            # replacement of a set that ran slower than its sibling: its observation buffer moves as well — torch would hand the
            # block just released straight back, so the cache is emptied and a bounded spacer taken first (freed below)
            torch.cuda.empty_cache()
            free_b = torch.cuda.mem_get_info(dev)[0]
            want_b = min(self.SPACER_BYTES, free_b // 3 - (OW + S) * rows * B * isz)
            if want_b >= (1 << 30):
                with _native._on_device(dev):
                    obs_spacer = _native.raw_malloc((self.replaced.get(key, 1) % 2 + 1) * want_b // 2)
        obs_shape = (B, rows, OW) if env_major else (rows, OW, B)
        try:
            ts.obs_buf = torch.empty(obs_shape, dtype=dt, device=dev)
        except torch.OutOfMemoryError:
            # dead pooled sets live outside torch's cache: give them (and the cache) back and try once more
            self.sets = []
            torch.cuda.empty_cache()
            ts.obs_buf = torch.empty(obs_shape, dtype=dt, device=dev)
        finally:
            if obs_spacer is not None:
                _native.raw_free(obs_spacer)
        ts.lbuf = torch.empty((S, last_e), dtype=dt, device=dev)
        if want_states:
            lb = ts.lbuf.data_ptr()
            last_ptrs = _native.ptr_array([lb + j * last_e * isz for j in range(S)])
            obs_ptr = ts.obs_buf.data_ptr()

            def time_launch(block):
                ptrs = _native.ptr_array([block.data_ptr() + j * leaf_e * isz for j in range(S)])
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                launch(obs_ptr, ptrs, last_ptrs)  # warm (clocks, TLB)
                for e in ev[:-1]:  # two timed launches, the faster counts: the first ones of a process run a few % slow
                    e.record()
                    launch(obs_ptr, ptrs, last_ptrs)
                ev[-1].record()
                ev[-1].synchronize()
                return min(float(a.elapsed_time(b)) for a, b in zip(ev[:-1], ev[1:]))

            # a set that is not pooled is written once: probing its placement (four extra launches per candidate) would never pay
            pattern = None
            if pattern_ctx is not None and pooled and not env_major and (B * isz) % 16 == 0 and rows >= 10:
                pattern = lambda block: self.pattern_score(ts.obs_buf, block.data_ptr(), leaf_e, B, rows, OW, S, isz, *pattern_ctx)
            shape = (S, leaf_e) if env_major else None
            try:
                ts.st_buf, ts.placement = self.place_state_block(ts.obs_buf, B, rows, OW, S, isz,
                                                                 time_launch if (launch is not None and pooled) else None,
                                                                 shape, known_ms, pattern)
            except torch.OutOfMemoryError:
                self.sets = []
                torch.cuda.empty_cache()
                ts.st_buf, ts.placement = self.place_state_block(ts.obs_buf, B, rows, OW, S, isz, None, shape)
            self.last = ts.placement
        self._finish_views(ts, B, rows, S, leaf_e, last_e, isz, env_major, want_states)
        if replacing is not None:
            # the searched replacement must beat the set it replaces in the same currency (its probe time against the old set's
            # steady real launches, which run a little faster than probes): otherwise the old set stays
            new_ms = (ts.placement or {}).get("chosen_ms")
            new_ratio, old_ratio = (ts.placement or {}).get("pattern_over_fill"), (replacing.placement or {}).get("pattern_over_fill")
            if new_ratio is not None:  # judged by the pattern: the new set must be clearly better placed than the old one was
                keep_old = old_ratio is not None and new_ratio < old_ratio + 0.015
            else:
                keep_old = new_ms is None or replacing.steady_ms is None or new_ms > 0.99 * replacing.steady_ms
            if keep_old:
                self.sets.append(replacing)
                if ts.placement is not None:
                    ts.placement["kept_old_set_ms"] = replacing.steady_ms
                    self.last = ts.placement
                return replacing
        if pooled:
            self._arm(ts, stream)
            self.sets.append(ts)
        return ts

    # Deterministic placement of the FIRST two sets of a shape (round 4): one arena laid out
    #     [observations A | observations B | state block A | state block B]
    # so that a launch's two kinds of write streams — 8 observation components, 7 state leaves for PMSM — start at least
    # QUAD_MIN_DISTANCE apart inside one allocation (the other set's observations are the distance), which is what turned the slow
    # placement level into the fast one in every experiment of profiles/r03_placement_regions.md. tools/placement_arena.py: 0.709 /
    # 0.716 of the roof for the two sets of the headline launch, as good as the search, with no probe launches of the kernel, no
    # spacers, and the two sets run alike (driver, round 4: 4.884 / 4.887 ms). Round 5 tried FOUR allocations in the same order
    # instead (so that a kept view would pin one array, not the arena): the driver does not place consecutive allocations like one
    # block — the pair failed the absolute criterion, the search took over and the two sets ran 3 % apart (5.03 / 4.89 ms, bench
    # 0.718): one allocation it stays. Its price: both sets are views of one storage — a foreign alias of ANY returned array (a kept
    # view, `detach()`, DLPack) makes both sets look busy, the calls that follow make single searched sets (at most POOL_SETS of them
    # stay pooled), and the arena's memory returns when the alias dies (`trajectory_placement = "search"` restores one allocation per
    # returned array). The plain chain `obs, states, state = env.vmap_sim_ahead(state, ...)` alternates between the two sets.
    def _ordered_pair(self, key, B, rows, OW, S, last_e, isz, stream, env_major=False, pattern_ctx=None):
        env = self.env
        dt, dev = env.dtype, env.device
        up = lambda n: (n + 63) // 64 * 64  # every sub-buffer starts on a 256-byte boundary
        leaf_e = (rows * B * isz + 127) // 128 * 128 // isz if env_major else rows * B
        obs_e, blk_e = up(rows * OW * B), up(S * leaf_e)
        if min(2 * obs_e, obs_e + blk_e) * isz < self.QUAD_MIN_DISTANCE:
            return None  # the other set's observations are not enough distance (an artificial gap measured 0.57 for C2): search
        total = 2 * obs_e + 2 * blk_e
        if total * isz > torch.cuda.mem_get_info(dev)[0] * 0.8:
            return None  # not worth crowding the device: the searched single sets take over
        try:
            arena = torch.empty(total, dtype=dt, device=dev)
        except torch.OutOfMemoryError:
            return None
        self.quad_made.add(key)
        sets = []
        for k in range(2):
            ts = TrajSet(key)
            ts.obs_buf = arena[k * obs_e: k * obs_e + rows * OW * B].view((B, rows, OW) if env_major else (rows, OW, B))
            b0 = 2 * obs_e + k * blk_e
            ts.st_buf = arena[b0: b0 + S * leaf_e].view((S, leaf_e) if env_major else (S, rows, B))
            ts.lbuf = torch.empty((S, last_e), dtype=dt, device=dev)
            ts.placement = {"arena_gib": round(total * isz / 2**30, 2), "set": k,
                            "what": "one arena [obs A | obs B | states A | states B]: no probe launches of the kernel"}
            self._finish_views(ts, B, rows, S, leaf_e, last_e, isz, env_major, True)
            sets.append(ts)
        del arena
        if pattern_ctx is not None and not env_major and (B * isz) % 16 == 0 and rows >= 10:
            # both sets must be in the fast level by the absolute criterion, else the arena goes back and the sets are searched
            for ts in sets:
                ms, ratio = self.pattern_score(ts.obs_buf, ts.st_buf.data_ptr(), leaf_e, B, rows, OW, S, isz, *pattern_ctx)
                ts.placement["pattern_over_fill"] = round(ratio, 4)
                ts.placement["pattern_ms"] = round(ms, 4)
                ts.placement["accept_at"] = self.PATTERN_ACCEPT
            if min(t.placement["pattern_over_fill"] for t in sets) < self.PATTERN_ACCEPT:
                self.last = {"arena_rejected": [t.placement["pattern_over_fill"] for t in sets], "accept_at": self.PATTERN_ACCEPT}
                del sets, ts
                torch.cuda.empty_cache()
                return None
        for ts in sets:  # the counts of an untouched pair: every view of both sets exists, nothing outside refers to any
            self._arm(ts, stream)
        self.sets.extend(reversed(sets))  # set B waits in the pool (dead: nothing refers to it), set A is handed out
        self.last = sets[0].placement
        return sets[0]
