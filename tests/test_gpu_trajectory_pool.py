"""Large vmap_sim_ahead outputs: pooled output sets (a dead set is written again, never a live one), the placement check
that runs once per new set, and the calibration entry point excenv_stream_pattern behind it. The functional contract of the
reference (core_env.py:571-616: every call returns arrays nothing else refers to, inputs are never written) must hold with
the pool on. ``-m gpu``."""
import ctypes

import numpy as np
import pytest
import torch

from exciting_environments_amd import _native
from helpers import NP_DTYPE, make_env, random_state, spec_of, to_state

pytestmark = pytest.mark.gpu


def _env(env_name="pmsm", B=4096, dtype=torch.float32, pool=True, placed=True):
    env, props, keep, spec = make_env(env_name, B, dtype)
    env._SHARED_TRAJ_BYTES = 0  # test sizes through the path of the large outputs
    env._placement.PLACED_BYTES = 0 if placed else (1 << 62)
    env.trajectory_pool = pool
    st = random_state(env_name, B, NP_DTYPE[dtype], spec, seed=5)
    return env, to_state(env, st)


def _actions(env, K, seed):
    a = env.new_actions_buffer(K)
    g = torch.Generator(device=env.device)
    g.manual_seed(seed)
    a.copy_((torch.rand((K, env.action_dim, env.batch_size), generator=g, device=env.device, dtype=env.dtype) * 2 - 1).permute(2, 0, 1))
    return a


def _leaves(env, st):
    return [getattr(st.physical_state, n) for n in env.STATE_FIELDS]


@pytest.mark.parametrize("env_name", ["pmsm", "pendulum"])
def test_chained_run_alternates_between_two_sets_and_equals_the_unpooled_run(env_name):
    K = 16
    env, s0 = _env(env_name)
    ref_env, r0 = _env(env_name, pool=False, placed=False)
    acts = [_actions(env, K, 100 + i) for i in range(6)]
    ptrs, state, rstate = [], s0, r0
    for a in acts:
        obs, states, state = env.vmap_sim_ahead(state, a, env.tau, env.tau)
        robs, rstates, rstate = ref_env.vmap_sim_ahead(rstate, a, env.tau, env.tau)
        assert torch.equal(obs, robs)
        for x, y in zip(_leaves(env, states), _leaves(ref_env, rstates)):
            assert torch.equal(x, y)
        for x, y in zip(_leaves(env, state), _leaves(ref_env, rstate)):
            assert torch.equal(x, y)
        ptrs.append(obs.data_ptr())
        del obs, states
    assert len(set(ptrs)) == 2 and ptrs[0] == ptrs[2] == ptrs[4] and ptrs[1] == ptrs[3] == ptrs[5]
    assert len(env._placement.sets) == 2
    lp = env.last_placement  # lane-major actions: judged by the access pattern against the fill rate (round 4), else by launch times
    assert lp is not None and len(lp.get("candidate_pattern_over_fill") or lp.get("candidate_ms")) >= 1


@pytest.mark.parametrize("held", ["observations", "state_leaf", "last_leaf", "view", "states_object", "detach"])
def test_a_set_somebody_can_still_see_is_never_written_again(held):
    K = 12
    env, s0 = _env("pmsm", B=2048)
    a = _actions(env, K, 7)
    obs, states, last = env.vmap_sim_ahead(s0, a, env.tau, env.tau)
    keep_obs, keep_leaf, keep_last = obs.clone(), states.physical_state.i_q.clone(), last.physical_state.i_q.clone()
    if held == "observations":
        holder = obs
    elif held == "state_leaf":
        holder = states.physical_state.i_q
    elif held == "last_leaf":
        holder = last.physical_state.i_q
    elif held == "view":
        holder = obs[:, -1, :]
    elif held == "states_object":
        holder = states
    else:
        holder = obs.detach()
    first_ptr = obs.data_ptr()
    del obs, states, last
    for i in range(4):  # other inputs: a reused set would show different values
        o2, s2, l2 = env.vmap_sim_ahead(s0, _actions(env, K, 50 + i), env.tau, env.tau)
        assert o2.data_ptr() != first_ptr
        del o2, s2, l2
    if held in ("observations", "detach"):
        assert torch.equal(holder, keep_obs)
    elif held == "state_leaf":
        assert torch.equal(holder, keep_leaf)
    elif held == "last_leaf":
        assert torch.equal(holder, keep_last)
    elif held == "view":
        assert torch.equal(holder, keep_obs[:, -1, :])
    else:
        assert torch.equal(holder.physical_state.i_q, keep_leaf)
    del holder
    o3, s3, l3 = env.vmap_sim_ahead(s0, a, env.tau, env.tau)  # nothing refers to the first set any more: it comes back
    assert o3.data_ptr() == first_ptr
    assert torch.equal(o3, keep_obs)


def test_a_set_is_not_reused_from_another_stream_and_inputs_are_never_written():
    K = 10
    env, s0 = _env("pendulum", B=1024)
    a = _actions(env, K, 3)
    before = [t.clone() for t in _leaves(env, s0)]
    obs, states, last = env.vmap_sim_ahead(s0, a, env.tau, env.tau)
    p0 = obs.data_ptr()
    del obs, states, last
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        o2, _, _ = env.vmap_sim_ahead(s0, a, env.tau, env.tau)
        assert o2.data_ptr() != p0
    side.synchronize()
    for x, y in zip(_leaves(env, s0), before):
        assert torch.equal(x, y)
    env.release_trajectory_buffers()
    assert env._placement.sets == []


def test_last_state_of_a_pooled_set_feeds_the_next_call_without_aliasing():
    """`state = last` keeps the previous set's last_state alive as the input of the next call: that set must stay busy."""
    K = 9
    env, s0 = _env("mass_spring_damper", B=1024)
    ref_env, r0 = _env("mass_spring_damper", B=1024, pool=False, placed=False)
    state, rstate = s0, r0
    for i in range(5):
        a = _actions(env, K, 20 + i)
        _, _, state = env.vmap_sim_ahead(state, a, env.tau, env.tau)
        _, _, rstate = ref_env.vmap_sim_ahead(rstate, a, env.tau, env.tau)
    for x, y in zip(_leaves(env, state), _leaves(ref_env, rstate)):
        assert torch.equal(x, y)


def test_stream_pattern_touches_exactly_the_rows_it_is_given():
    dev = torch.device("cuda")
    row_elems, rows_total, rows = 4096 + 8, 12, 7  # a ragged last 4 KiB piece
    rb = row_elems * 4
    rd = torch.ones((rows_total, 2, row_elems), device=dev)
    wr = torch.full((3, rows_total, row_elems), -5.0, device=dev)
    stream = _native.raw_stream(torch.cuda.current_device())
    _native.stream_pattern([rd.data_ptr(), rd.data_ptr() + rb], [2 * rb, 2 * rb], [wr[q].data_ptr() for q in range(3)], [rb] * 3, rb,
                           rows, stream)
    torch.cuda.synchronize()
    w = wr.cpu().numpy()
    assert np.all(w[:, rows:, :] == -5.0)
    for q in range(3):  # row n holds the running sum of the two read streams (2 per row) + the stream index
        for n in range(rows):
            assert np.all(w[q, n] == 2.0 * (n + 1) + q)


def test_stream_pattern_rejects_bad_arguments():
    lib = _native.lib()
    buf = torch.zeros(1024, device="cuda")
    p = (ctypes.c_void_p * 1)(buf.data_ptr())
    rs = (ctypes.c_int64 * 1)(4096)
    assert lib.excenv_stream_pattern(0, None, None, 1, p, rs, 4096, 1, 1, None) == 0
    assert lib.excenv_stream_pattern(0, None, None, 1, p, rs, 4090, 1, 1, None) == -1  # row_bytes not a multiple of 16
    assert lib.excenv_stream_pattern(0, None, None, 33, p, rs, 4096, 1, 1, None) == -1
    assert lib.excenv_stream_pattern(0, None, None, 1, None, rs, 4096, 1, 1, None) == -2
    p_odd = (ctypes.c_void_p * 1)(buf.data_ptr() + 4)
    assert lib.excenv_stream_pattern(0, None, None, 1, p_odd, rs, 4096, 1, 1, None) == -1
    torch.cuda.synchronize()


def test_placement_search_logic_with_scripted_timings():
    """The decision rules of the placement search, with the launch timing replaced by a script: the first set of a shape tries
    at least two candidates and stops at a 7 % contrast; a later set stops at the first candidate within 2 % of the best time
    known; without contrast all tries are used and the fastest wins; rejected blocks are released."""
    env, _ = _env("pmsm", B=1024)
    env._placement.SPACER_BYTES = 1 << 20
    B, rows, OW, S, isz = 1024, 9, 8, 7, 4

    def run(times):
        seen, it = [], iter(times)

        def fake(block):
            seen.append(block.data_ptr())
            return next(it)

        obs_buf = torch.empty((rows, OW, B), dtype=torch.float32, device=env.device)
        block, diag = env._placement.place_state_block(obs_buf, B, rows, OW, S, isz, fake)
        return block, diag, seen

    env._placement.best.clear()
    block, diag, seen = run([5.4, 4.9, 9.9, 9.9])          # contrast after two candidates: stop, keep the second
    assert diag["candidate_ms"] == [5.4, 4.9] and diag["chosen"] == 1 and block.data_ptr() == seen[1]
    assert env._placement.best[(B, rows, OW, S)] == 4.9
    block, diag, seen = run([4.95, 9.9])                   # a later set: within 2 % of the best known -> first candidate
    assert diag["candidate_ms"] == [4.95] and diag["best_known_ms_before"] == 4.9
    block, diag, seen = run([5.3, 5.35, 5.1, 5.2, 5.25, 5.4])  # never matches 4.9: all six tries of a small block, the fastest kept
    assert len(diag["candidate_ms"]) == 6 and diag["chosen"] == 2 and block.data_ptr() == seen[2]
    assert len(set(seen)) == 6                             # six distinct blocks were alive at the same time
    env._placement.best.clear()
    block, diag, seen = run([5.0, 5.05, 5.02, 4.98, 5.01, 5.03])  # first set, no contrast: all tries, fastest kept
    assert len(diag["candidate_ms"]) == 6 and diag["chosen"] == 3
    env.trajectory_placement = "off"
    block, diag, seen = run([1.0])
    assert diag is None and seen == []


@pytest.mark.parametrize("env_name", ["pmsm", "pendulum"])
def test_env_major_sets_are_pooled_and_equal_the_unpooled_run(env_name):
    """Reference-shaped (row-major) trajectories go through the same pooled, placed sets: contiguous arrays, two alternating
    sets in a chained run, the bits of the plain allocation per call."""
    K, B = 16, 4096
    env, s0 = _env(env_name, B=B)
    ref_env, r0 = _env(env_name, B=B, pool=False, placed=False)
    env.traj_layout = ref_env.traj_layout = "env_major"
    ref_env._placement.PLACED_BYTES = 1 << 62  # the plain path of the small env-major outputs
    g = torch.Generator(device=env.device)
    ptrs, state, rstate = [], s0, r0
    for i in range(6):
        g.manual_seed(300 + i)
        a = (torch.rand((B, K, env.action_dim), generator=g, device=env.device, dtype=env.dtype) * 2 - 1)
        obs, states, state = env.vmap_sim_ahead(state, a, env.tau, env.tau)
        robs, rstates, rstate = ref_env.vmap_sim_ahead(rstate, a, env.tau, env.tau)
        assert obs.is_contiguous() and torch.equal(obs, robs)
        for x, y in zip(_leaves(env, states), _leaves(ref_env, rstates)):
            assert x.is_contiguous() and x.data_ptr() % 128 == 0 and torch.equal(x, y)
        for x, y in zip(_leaves(env, state), _leaves(ref_env, rstate)):
            assert torch.equal(x, y)
        ptrs.append(obs.data_ptr())
        del obs, states
    assert len(set(ptrs)) == 2 and ptrs[0] == ptrs[2] == ptrs[4] and ptrs[1] == ptrs[3] == ptrs[5]
    held = env.vmap_sim_ahead(s0, a, env.tau, env.tau)[0]
    keep = held.clone()
    for i in range(3):
        o2 = env.vmap_sim_ahead(s0, a * 0.5, env.tau, env.tau)[0]
        assert o2.data_ptr() != held.data_ptr()
        del o2
    assert torch.equal(held, keep)


def test_unpooled_sets_are_not_probed():
    """Without the pool every call allocates its outputs and writes them once: the placement search (extra launches) is skipped."""
    env, s0 = _env("pendulum", B=2048, pool=False, placed=True)
    a = _actions(env, 8, 11)
    o1 = env.vmap_sim_ahead(s0, a, env.tau, env.tau)[0]
    o2 = env.vmap_sim_ahead(s0, a, env.tau, env.tau)[0]
    assert env.last_placement is None and env._placement.sets == []
    assert o1.data_ptr() != o2.data_ptr() and torch.equal(o1, o2)


def test_pool_wait_stream_orders_the_reuse_behind_a_foreign_reader():
    """the pools cannot see Tensor.record_stream. A consumer that reads a returned tensor on a side stream and
    drops its reference early calls env.pool_wait_stream(side): the launch that writes the set again then waits for that stream, so
    the reader still sees the values it was handed."""
    env, st = _env("pmsm", B=1 << 15)
    K = 24
    acts = [_actions(env, K, 60 + i) for i in range(3)]
    obs, states, last = env.vmap_sim_ahead(st, acts[0], env.tau, env.tau)
    want = obs.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        torch.cuda._sleep(400_000_000)      # the reader is slow: ~0.2 s before it touches the buffer
        got = obs + 0                       # reads the pooled observation buffer on the side stream
    env.pool_wait_stream(side)
    ptr = obs.data_ptr()
    del obs, states
    second = env.vmap_sim_ahead(last, acts[1], env.tau, env.tau)
    del last                                                           # nothing of the first set is referenced any more
    third = env.vmap_sim_ahead(second[2], acts[2], env.tau, env.tau)   # ... so the pool hands it out again
    assert third[0].data_ptr() == ptr, "the dead set was expected to be written again"
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    assert not torch.equal(third[0], want)


def test_placement_settles_and_reports_real_launch_times():
    """Sets the absolute criterion cannot judge (here: row-major trajectories) are compared by HIP-event times of their REAL launches —
    the first launch into a set is recorded apart and never a judgement; `trajectory_placement_settled` turns True once every pooled
    set of the shape has a steady time and none is up for replacement. Sets accepted by the absolute criterion are final at once."""
    env, st = _env("pendulum", B=1 << 14)
    K = 32
    g = torch.Generator(device=env.device)
    g.manual_seed(70)
    acts = torch.rand((env.batch_size, K, 1), generator=g, device=env.device) * 2 - 1
    env.traj_layout = "env_major"  # the register-ring kernel's write pattern has no replay: judged by real launches
    assert env.trajectory_placement_settled  # nothing pooled yet
    out = env.vmap_sim_ahead(st, acts, env.tau, env.tau)
    seen = [env.trajectory_placement_settled]
    for _ in range(12):
        out = env.vmap_sim_ahead(out[2], acts, env.tau, env.tau)
        torch.cuda.synchronize()
        seen.append(env.trajectory_placement_settled)
    assert seen[0] is False and seen[-1] is True
    sets = env._placement.sets
    assert len(sets) == 2 and all(t.first_ms is not None and t.steady_ms is not None and t.steady_ms > 0 for t in sets)
    env.trajectory_placement = "off"
    assert env.trajectory_placement_settled
    # lane-major actions: judged by the pattern -> settled as soon as both sets of the shape exist
    env2, st2 = _env("pendulum", B=1 << 14)
    env2._placement.PATTERN_ACCEPT = 0.0  # test-sized sets: whatever the pattern says is accepted
    a2 = _actions(env2, K, 71)
    o = env2.vmap_sim_ahead(st2, a2, env2.tau, env2.tau)
    o = env2.vmap_sim_ahead(o[2], a2, env2.tau, env2.tau)
    assert len(env2._placement.sets) == 2 and all(t.judged_by_pattern for t in env2._placement.sets)
    assert env2.trajectory_placement_settled


def test_arena_pair_is_made_once_reused_and_equals_the_unpooled_run():
    """The first two sets of a shape are views of ONE arena [obs A | obs B | states A | states B] — no probe launches. The chained
    run alternates between them, holds the bits of the unpooled run, never writes a set somebody still sees, and a caller that holds
    on to outputs gets searched single sets from the third call on (no second arena, at most POOL_SETS further sets pooled)."""
    env, st = _env("pmsm", B=2048)
    pl = env._placement
    pl.QUAD_MIN_SET_BYTES, pl.QUAD_MIN_DISTANCE, pl.PATTERN_ACCEPT = 0, 0, 0.0  # test-sized sets: accept whatever the pattern says
    ref_env, ref_st = _env("pmsm", B=2048, pool=False)
    K = 12
    acts = [_actions(env, K, 80 + i) for i in range(5)]
    ptrs = []
    s, rs = st, ref_st
    for i in range(5):
        o, states, s = env.vmap_sim_ahead(s, acts[i], env.tau, env.tau)
        ro, rstates, rs = ref_env.vmap_sim_ahead(rs, acts[i], ref_env.tau, ref_env.tau)
        assert torch.equal(o, ro) and torch.equal(states.physical_state.i_q, rstates.physical_state.i_q)
        ptrs.append(o.data_ptr())
        if i == 0:
            assert env.last_placement["what"].startswith("one arena") and len(pl.sets) == 2
            a0, a1 = pl.sets[0], pl.sets[1]
            assert a0.obs_buf.untyped_storage().data_ptr() == a1.obs_buf.untyped_storage().data_ptr()  # one allocation
            for t in (a0, a1):  # observations of both sets first, then the state blocks: each set's streams start apart
                assert t.st_buf.data_ptr() - t.obs_buf.data_ptr() >= min(2 * a0.obs_buf.numel(), a0.obs_buf.numel() + a0.st_buf.numel()) * 4
        del o, states
    assert ptrs[0] == ptrs[2] == ptrs[4] and ptrs[1] == ptrs[3] and ptrs[0] != ptrs[1]
    # a kept view of a returned array makes BOTH arena sets busy (one storage): the calls that follow alternate between two single
    # sets, no third, fourth ... set is made, and the view still shows what it was handed
    held = env.vmap_sim_ahead(s, acts[0], env.tau, env.tau)
    view, keep = held[0][:, -1], held[0][:, -1].clone()
    arena_ptrs = {ptrs[0], ptrs[1]}
    s = held[2]
    del held
    seen = []
    for i in range(6):
        o, _, s = env.vmap_sim_ahead(s, acts[i % 5], env.tau, env.tau)
        assert o.data_ptr() not in arena_ptrs
        seen.append(o.data_ptr())
        del o
    assert len(set(seen)) == 2 and seen[0] == seen[2] == seen[4] and len(pl.sets) == 2
    assert torch.equal(view, keep)
    del view
    # a caller that keeps every output: the pair is used up after two calls, then single sets (no second arena)
    env2, st2 = _env("pmsm", B=2048)
    p2 = env2._placement
    p2.QUAD_MIN_SET_BYTES, p2.QUAD_MIN_DISTANCE, p2.PATTERN_ACCEPT = 0, 0, 0.0
    held, s2 = [], st2
    for i in range(4):
        out = env2.vmap_sim_ahead(s2, acts[i], env2.tau, env2.tau)
        held.append(out)
        s2 = out[2]
    assert len({o[0].untyped_storage().data_ptr() for o in held}) == 3  # calls 1 and 2 share the arena, 3 and 4 have their own
    torch.cuda.synchronize()
    s3 = st
    for i in range(4):  # and what they hold is still what the unpooled run computes
        ro, _, s3 = ref_env.vmap_sim_ahead(s3 if i else ref_st, acts[i], ref_env.tau, ref_env.tau)
        assert torch.equal(held[i][0], ro)


def test_pattern_judged_placement_accepts_on_the_absolute_criterion():
    """The absolute placement criterion (pattern rate / fill rate >= PATTERN_ACCEPT) with a scripted score: the search stops at
    the first candidate that meets it, walks on while none does (keeping rejected blocks alive) and then keeps the best one."""
    env, _ = _env("pmsm", B=1024)
    env._placement.SPACER_BYTES = 1 << 20
    env._placement.fill_gbs = 6800.0
    B, rows, OW, S, isz = 1024, 12, 8, 7, 4

    def run(scores):
        it, seen = iter(scores), []

        def fake(block):
            seen.append(block.data_ptr())
            return next(it)

        obs_buf = torch.empty((rows, OW, B), dtype=torch.float32, device=env.device)
        block, diag = env._placement.place_state_block(obs_buf, B, rows, OW, S, isz, None, pattern=fake)
        return block, diag, seen

    block, diag, seen = run([(5.4, 0.76), (5.0, 0.83), (9.9, 0.9)])
    assert diag["candidate_pattern_over_fill"] == [0.76, 0.83] and diag["chosen"] == 1 and block.data_ptr() == seen[1]
    block, diag, seen = run([(5.0, 0.84)])
    assert diag["candidate_pattern_over_fill"] == [0.84] and "candidate_ms" not in diag
    block, diag, seen = run([(5.5, 0.70), (5.3, 0.74), (5.6, 0.69), (5.4, 0.72), (5.35, 0.73), (5.45, 0.71)])
    assert len(diag["candidate_pattern_over_fill"]) == 6 and diag["chosen"] == 1 and len(set(seen)) == 6
