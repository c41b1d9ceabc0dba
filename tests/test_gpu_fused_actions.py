"""Row-major actions read INSIDE the lane-major trajectory kernel (csrc/kernels.hpp, AEM instantiations): a plain
``actions[B, K, A]`` tensor — what the reference's ``vmap_sim_ahead`` is handed (core_env.py:571-616) — with the library's default
lane-major outputs needs no transposition pass; every wave fetches 64-byte windows of its environments' rows into LDS.
The results must be the bits of the same launch with lane-major actions (``env.new_actions_buffer``), of the transposition path,
and — directly — the oracle's values, for every window phase (K shorter than a window, a short last window, K * A not a multiple
of a piece or a ragged batch -> not fused, sub-steps, both dtypes, all solvers). ``-m gpu``."""
import numpy as np
import pytest
import torch

import oracle
from exciting_environments_amd import _native
from helpers import ANGLE_STATES, NP_DTYPE, circ_close, make_env, max_err, random_state, to_state

pytestmark = pytest.mark.gpu


def _vmax(dtype):
    return 4 if dtype is torch.float32 else 2


def _fuses(env, acts, K, opts):
    props, _keep = env._props_for(env.env_properties, env.batch_size)
    return _native.sim_ahead_fuses_actions(env.ENV_ID, env._solver.id, env.dtype, env.batch_size, K, props, len(env.control_state), False,
                                           _native.LAYOUT_ENV_MAJOR, _native.LAYOUT_LANE_MAJOR, acts.data_ptr(), opts)


def _same(env, a, b):
    assert torch.equal(a[0], b[0])
    for n in env.STATE_FIELDS:
        assert torch.equal(getattr(a[2].physical_state, n), getattr(b[2].physical_state, n)), n
        if a[1] is not None:
            assert torch.equal(getattr(a[1].physical_state, n), getattr(b[1].physical_state, n)), n


@pytest.mark.parametrize("env_name,dtype,solver", [
    ("pmsm", torch.float32, "euler"), ("pmsm", torch.float64, "euler"), ("pmsm", torch.float32, "tsit5"), ("pmsm", torch.float32, "rk4"),
    ("pendulum", torch.float32, "euler"), ("pendulum", torch.float64, "rk4"), ("fluid_tank", torch.float32, "euler"),
    ("cartpole", torch.float32, "euler"), ("acrobot", torch.float64, "euler"), ("acrobot", torch.float32, "tsit5"),
    ("mass_spring_damper", torch.float64, "tsit5"), ("mass_spring_damper", torch.float32, "euler")])
@pytest.mark.parametrize("B,K", [(4096, 100), (4096 + 192, 100), (2048, 128), (1024 + 512, 64), (4096, 8), (1000, 36), (4096, 4), (2048, 33), (512, 22)])
@pytest.mark.parametrize("semantics", ["ahead", "step"])
def test_fused_row_major_actions_equal_lane_major_actions(env_name, dtype, solver, B, K, semantics):
    env, props, keep, spec = make_env(env_name, B, dtype, solver=solver)
    env.sim_ahead_semantics = semantics
    env.launch_opts = _native.launch_opts(envs_per_lane=_vmax(dtype))  # the widest form at a test-sized batch
    st = random_state(env_name, B, NP_DTYPE[dtype], spec, seed=51)
    acts_np = np.random.default_rng(52).uniform(-1, 1, (B, K, env.action_dim)).astype(NP_DTYPE[dtype])
    plain = torch.as_tensor(acts_np, device=env.device)
    lane = env.new_actions_buffer(K)
    lane.copy_(plain)
    whole_pieces = (K * env.action_dim * plain.element_size()) % 16 == 0 and B % (64 * _vmax(dtype)) == 0
    assert _fuses(env, plain, K, env.launch_opts) == whole_pieces
    want = env.vmap_sim_ahead(to_state(env, st), lane, env.tau, env.tau)
    got = env.vmap_sim_ahead(to_state(env, st), plain, env.tau, env.tau)
    torch.cuda.synchronize()
    _same(env, got, want)
    # the transposition path of the same call (EXCENV_OPT_NO_FUSED_ACTIONS): same bits again
    env.launch_opts = _native.launch_opts(envs_per_lane=_vmax(dtype), flags=_native.OPT_NO_FUSED_ACTIONS)
    assert not _fuses(env, plain, K, env.launch_opts)
    _same(env, env.vmap_sim_ahead(to_state(env, st), plain, env.tau, env.tau), want)


@pytest.mark.parametrize("env_name,dtype", [("pendulum", torch.float32), ("mass_spring_damper", torch.float64), ("fluid_tank", torch.float32)])
@pytest.mark.parametrize("sub", [2, 5])
def test_fused_row_major_actions_with_substeps(env_name, dtype, sub):
    """action_stepsize = sub * obs_stepsize: every action row is held for `sub` solver steps; the ring must re-fill once per
    group of rows, not once per solver step."""
    B, K = 2048, 48
    env, props, keep, spec = make_env(env_name, B, dtype)
    env.launch_opts = _native.launch_opts(envs_per_lane=_vmax(dtype))
    st = random_state(env_name, B, NP_DTYPE[dtype], spec, seed=53)
    plain = torch.as_tensor(np.random.default_rng(54).uniform(-1, 1, (B, K, 1)).astype(NP_DTYPE[dtype]), device=env.device)
    lane = env.new_actions_buffer(K)
    lane.copy_(plain)
    assert _fuses(env, plain, K, env.launch_opts)
    want = env.vmap_sim_ahead(to_state(env, st), lane, env.tau / sub, env.tau)
    got = env.vmap_sim_ahead(to_state(env, st), plain, env.tau / sub, env.tau)
    assert tuple(got[0].shape) == (B, K * sub + 1, want[0].shape[-1])
    _same(env, got, want)


def test_fused_row_major_actions_observations_only_and_unaligned_fallback():
    B, K = 4096, 60
    env, props, keep, spec = make_env("pmsm", B, torch.float32)
    env.launch_opts = _native.launch_opts(envs_per_lane=4)
    env.store_state_trajectory = False
    st = random_state("pmsm", B, np.float32, spec, seed=55)
    acts_np = np.random.default_rng(56).uniform(-1, 1, (B, K, 2)).astype(np.float32)
    plain = torch.as_tensor(acts_np, device=env.device)
    lane = env.new_actions_buffer(K)
    lane.copy_(plain)
    want = env.vmap_sim_ahead(to_state(env, st), lane, env.tau, env.tau)
    got = env.vmap_sim_ahead(to_state(env, st), plain, env.tau, env.tau)
    assert got[1] is None and torch.equal(got[0], want[0])
    # an action array that does not start on a 16-byte boundary is not fused (and still right)
    off = torch.empty(B * K * 2 + 1, dtype=torch.float32, device=env.device)
    shifted = off[1:].view(B, K, 2)
    shifted.copy_(plain)
    assert not _fuses(env, shifted, K, env.launch_opts)
    got2 = env.vmap_sim_ahead(to_state(env, st), shifted, env.tau, env.tau)
    assert torch.equal(got2[0], want[0])


@pytest.mark.parametrize("env_name,dtype,solver,K", [("pmsm", torch.float32, "euler", 40), ("pmsm", torch.float64, "tsit5", 16),
                                                      ("pendulum", torch.float32, "rk4", 64)])
@pytest.mark.parametrize("semantics", ["ahead", "step"])
def test_fused_row_major_actions_match_the_oracle_at_a_headline_sized_batch(env_name, dtype, solver, K, semantics):
    """Default options at a batch that takes the widest form by itself (2^18 environments in fp32, 2^17 in fp64): the first and
    the last workgroups' environments against the oracle's trajectory from the same inputs — the fused path compared with the
    CPU restatement directly, not only with another kernel."""
    B = 1 << (18 if dtype is torch.float32 else 17)
    env, props, keep, spec = make_env(env_name, B, dtype, solver=solver)
    env.sim_ahead_semantics = semantics
    st = random_state(env_name, B, NP_DTYPE[dtype], spec, seed=57)
    acts_np = np.random.default_rng(58).uniform(-1, 1, (B, K, env.action_dim)).astype(NP_DTYPE[dtype])
    plain = torch.as_tensor(acts_np, device=env.device)
    assert _fuses(env, plain, K, None)
    obs, states, last = env.vmap_sim_ahead(to_state(env, st), plain, env.tau, env.tau)
    sem = oracle.SEM_STEP if semantics == "step" else oracle.SEM_AHEAD
    n = 1536
    for sl in (slice(0, n), slice(B - n, B), slice(B // 2 - 3, B // 2 - 3 + n)):
        sub_props, sub_keep = oracle.make_props(env_name, spec["params"], spec["phys_norm"], spec["act_norm"], NP_DTYPE[dtype], n)
        o_ref, s_ref, l_ref = oracle.sim_ahead(env_name, solver, [s[sl] for s in st], acts_np[sl], sub_props, spec["tau"], semantics=sem)
        rtol, atol = (1e-5, 2e-5) if dtype is torch.float32 else (1e-9, 1e-9)
        got = obs[sl].cpu().numpy()
        angle_cols = {"pendulum": [0]}.get(env_name, [])
        assert circ_close(got, o_ref, angle_cols, rtol, atol), max_err(got, o_ref)
        for j, name in enumerate(env.STATE_FIELDS):
            g = getattr(states.physical_state, name)[sl].cpu().numpy()
            scale = max(1.0, float(np.nanmax(np.abs(s_ref[j]))))
            if j in ANGLE_STATES.get(env_name, []):
                assert circ_close(g[..., None], s_ref[j][..., None], [0], rtol, atol * scale, period=2 * np.pi), name
            else:
                assert np.allclose(g, s_ref[j], rtol=rtol, atol=atol * scale), (name, max_err(g, s_ref[j]))


@pytest.mark.parametrize("env_name,control,dtype", [("pmsm", ["i_d", "i_q"], torch.float32), ("pendulum", ["theta"], torch.float32),
                                                    ("cartpole", ["theta", "deflection"], torch.float64)])
def test_fused_row_major_actions_with_control_columns(env_name, control, dtype):
    """Round 5: a control_state alone no longer sends a plain [B, K, A] call to the transposition pass — the lean kernel reads the
    row-major actions itself and control_fill_kernel fills the reference columns behind it. Same bits as the same call with
    lane-major actions, and the columns hold the normalised references."""
    B, K = 4096, 40
    env, props, keep, spec = make_env(env_name, B, dtype, control_state=list(control))
    env.launch_opts = _native.launch_opts(envs_per_lane=_vmax(dtype))
    st = random_state(env_name, B, NP_DTYPE[dtype], spec, seed=61)
    rng = np.random.default_rng(62)
    refs = {}
    for n in control:
        lo, hi = spec["phys_norm"][n]
        refs[n] = rng.uniform(lo, hi, B).astype(NP_DTYPE[dtype])
    plain = torch.as_tensor(rng.uniform(-1, 1, (B, K, env.action_dim)).astype(NP_DTYPE[dtype]), device=env.device)
    lane = env.new_actions_buffer(K)
    lane.copy_(plain)
    assert _fuses(env, plain, K, env.launch_opts)
    want = env.vmap_sim_ahead(to_state(env, st, reference=refs), lane, env.tau, env.tau)
    got = env.vmap_sim_ahead(to_state(env, st, reference=refs), plain, env.tau, env.tau)
    torch.cuda.synchronize()
    assert _native.last_launch() == "sim_ahead_kernel (row-major actions fused)"
    _same(env, got, want)
    O = got[0].shape[-1]
    for j, n in enumerate(control):
        lo, hi = spec["phys_norm"][n]
        col = got[0][:, :, O - len(control) + j]
        want_col = torch.as_tensor(2 * (refs[n] - lo) / (hi - lo) - 1, device=env.device)[:, None].expand_as(col)
        assert torch.allclose(col, want_col, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("env_name,dtype,K", [("pmsm", torch.float64, 40), ("pmsm", torch.float32, 40), ("fluid_tank", torch.float32, 64),
                                              ("pendulum", torch.float64, 48), ("mass_spring_damper", torch.float32, 16)])
@pytest.mark.parametrize("offset_bytes", [16, 32, 48])
def test_fused_row_major_actions_from_a_pointer_inside_a_sector(env_name, dtype, K, offset_bytes):
    """The actions tensor starts 16 / 32 / 48 bytes into a 64-byte sector (a slice of a larger allocation): the first window of an
    environment's row then holds three, two or ONE piece. The single-piece case re-requests the window in the kernel's prologue, with no
    saved row between that fill and the loop's first read — tools/isa_guards.py found that nothing stood behind the fill for the counted
    wait to count (round 5, second half; the row-wise form of rounds 4 - 5 had the hole for one-row pieces, i.e. PMSM fp64). Same bits as
    the lane-major call, repeatedly (a fill that has not landed would show as stale LDS in the rows of the second window)."""
    B = 4096
    env, props, keep, spec = make_env(env_name, B, dtype, solver="euler")
    env.launch_opts = _native.launch_opts(envs_per_lane=_vmax(dtype))
    st = random_state(env_name, B, NP_DTYPE[dtype], spec, seed=71)
    A = env.action_dim
    n = B * K * A
    isz = 4 if dtype is torch.float32 else 8
    whole = torch.zeros(n + 64, dtype=dtype, device=env.device)
    skip = ((-whole.data_ptr()) % 64 + offset_bytes) // isz
    plain = whole[skip:skip + n].view(B, K, A)
    assert plain.data_ptr() % 64 == offset_bytes
    plain.copy_(torch.as_tensor(np.random.default_rng(72).uniform(-1, 1, (B, K, A)).astype(NP_DTYPE[dtype]), device=env.device))
    lane = env.new_actions_buffer(K)
    lane.copy_(plain)
    assert _fuses(env, plain, K, env.launch_opts)
    want = env.vmap_sim_ahead(to_state(env, st), lane, env.tau, env.tau)
    for _ in range(5):
        whole[:skip].fill_(float("nan"))  # what lies in front of the first row is never used
        got = env.vmap_sim_ahead(to_state(env, st), plain, env.tau, env.tau)
        torch.cuda.synchronize()
        assert _native.last_launch() == "sim_ahead_kernel (row-major actions fused)"
        _same(env, got, want)
