"""The register-ring form of the fused env-major trajectory kernel (csrc/kernels_emr.hpp): whole-line stores out of a per-lane
window of saved states. Reference row-major arrays (core_env.py:571-616: observations [B, K+1, O], state leaves [B, K+1]) must
hold the same bits as the LDS-ring form and as the lane-major kernel, for every window phase (head / whole / tail windows),
ragged batches (waves with idle lanes) and both dtypes. ``-m gpu``."""
import numpy as np
import pytest
import torch

from exciting_environments_amd import _native
from helpers import NP_DTYPE, make_env, random_state, to_state

pytestmark = pytest.mark.gpu


def _run(env, st, acts, mode):
    env.traj_layout = "env_major"
    env.env_major_fused, env.env_major_workspace = True, True
    env.launch_opts = _native.launch_opts(env_major_mode=mode)
    out = env.vmap_sim_ahead(to_state(env, st), acts, env.tau, env.tau)
    torch.cuda.synchronize()
    return out


def _same(env, a, b):
    assert a[0].is_contiguous() and torch.equal(a[0], b[0])
    for n in env.STATE_FIELDS:
        assert torch.equal(getattr(a[2].physical_state, n), getattr(b[2].physical_state, n))
        if a[1] is not None:
            assert torch.equal(getattr(a[1].physical_state, n), getattr(b[1].physical_state, n))


@pytest.mark.parametrize("env_name,dtype,solver", [
    ("pmsm", torch.float32, "euler"), ("pmsm", torch.float64, "euler"), ("pmsm", torch.float32, "tsit5"),
    ("pendulum", torch.float32, "euler"), ("pendulum", torch.float64, "rk4"), ("fluid_tank", torch.float32, "euler"),
    ("cartpole", torch.float32, "euler"), ("acrobot", torch.float64, "euler"), ("mass_spring_damper", torch.float32, "tsit5")])
@pytest.mark.parametrize("B,K", [(4096, 100), (4096 + 192, 100), (2048, 127), (2049, 64), (4096, 7), (1000, 33), (4096, 160)])
def test_register_ring_equals_lds_ring_and_lane_major(env_name, dtype, solver, B, K):
    if (B * K * (2 if env_name == "pmsm" else 1) * (4 if dtype is torch.float32 else 8)) % 16:
        pytest.skip("the fused kernels need an action array made of whole 16-byte pieces")
    env, props, keep, spec = make_env(env_name, B, dtype, solver=solver)
    st = random_state(env_name, B, NP_DTYPE[dtype], spec, seed=41)
    acts = torch.as_tensor(np.random.default_rng(42).uniform(-1, 1, (B, K, env.action_dim)).astype(NP_DTYPE[dtype]), device=env.device)
    lds = _run(env, st, acts, 2)
    ring = _run(env, st, acts, 3)
    _same(env, ring, lds)
    env.traj_layout, env.launch_opts = "lane_major", None
    lane = env.vmap_sim_ahead(to_state(env, st), acts, env.tau, env.tau)
    _same(env, ring, lane)


@pytest.mark.parametrize("semantics", ["ahead", "step"])
def test_register_ring_both_semantics_and_observations_only(semantics):
    B, K = 4096, 75
    env, props, keep, spec = make_env("pmsm", B, torch.float32)
    env.sim_ahead_semantics = semantics
    st = random_state("pmsm", B, np.float32, spec, seed=43)
    acts = torch.as_tensor(np.random.default_rng(44).uniform(-1, 1, (B, K, 2)).astype(np.float32), device=env.device)
    _same(env, _run(env, st, acts, 3), _run(env, st, acts, 2))
    env.store_state_trajectory = False
    a, b = _run(env, st, acts, 3), _run(env, st, acts, 2)
    assert a[1] is None and b[1] is None
    _same(env, a, b)


def test_register_ring_is_what_a_large_default_call_runs():
    """Default options, a batch large enough for the heuristic: the result equals the forced LDS-ring form."""
    B, K = 65536, 100
    env, props, keep, spec = make_env("pmsm", B, torch.float32)
    st = random_state("pmsm", B, np.float32, spec, seed=45)
    acts = torch.as_tensor(np.random.default_rng(46).uniform(-1, 1, (B, K, 2)).astype(np.float32), device=env.device)
    _same(env, _run(env, st, acts, 0), _run(env, st, acts, 2))


def test_env_major_mode_values_are_validated():
    env, props, keep, spec = make_env("pendulum", 256, torch.float32)
    st = random_state("pendulum", 256, np.float32, spec, seed=1)
    acts = torch.zeros((256, 8, 1), device=env.device)
    for bad in (-1, 4):
        with pytest.raises(RuntimeError, match="env_major_mode must be 0, 1, 2 or 3"):
            _run(env, st, acts, bad)
    for ok in (0, 1, 2, 3):
        _run(env, st, acts, ok)


@pytest.mark.filterwarnings("ignore:the reference's 1 . int")
def test_register_ring_fuzz_against_lane_major():
    """Random (model, dtype, solver, semantics, B, K, observations-only) draws: the register-ring form, forced on whatever the
    batch size, holds the bits of the lane-major kernel."""
    rng = np.random.default_rng(2026)
    names = ["pmsm", "pendulum", "fluid_tank", "cartpole", "acrobot", "mass_spring_damper"]
    done = 0
    for it in range(60):
        env_name = names[rng.integers(len(names))]
        dtype = [torch.float32, torch.float64][rng.integers(2)]
        solver = ["euler", "rk4", "tsit5"][rng.integers(3)]
        B = int(rng.integers(1, 40)) * 64 + int(rng.integers(0, 64)) * int(rng.integers(0, 2))
        K = int(rng.integers(1, 140))
        A = 2 if env_name == "pmsm" else 1
        if (B * K * A * (4 if dtype is torch.float32 else 8)) % 16:
            continue
        env, props, keep, spec = make_env(env_name, B, dtype, solver=solver)
        env.sim_ahead_semantics = ["ahead", "step"][rng.integers(2)]
        env.store_state_trajectory = bool(rng.integers(0, 4))
        st = random_state(env_name, B, NP_DTYPE[dtype], spec, seed=1000 + it)
        acts = torch.as_tensor(rng.uniform(-1, 1, (B, K, env.action_dim)).astype(NP_DTYPE[dtype]), device=env.device)
        ring = _run(env, st, acts, 3)
        env.traj_layout, env.launch_opts = "lane_major", None
        lane = env.vmap_sim_ahead(to_state(env, st), acts, env.tau, env.tau)
        _same(env, ring, lane)
        done += 1
    assert done >= 30


# ---- the register-ring form against the reference's fixtures and the oracle DIRECTLY (not only against other kernels) --------
@pytest.mark.parametrize("env_name", ["pendulum", "mass_spring_damper", "cartpole", "acrobot", "fluid_tank", "pmsm"])
def test_fixture_sim_ahead_fp64_through_the_register_ring(env_name, golden):
    """The whole reference fixture (10 000 steps; PMSM 1 000) in ONE launch of the register-ring kernel (forced: 64 environments
    would take the LDS-ring form otherwise), reference-shaped row-major arrays, at the reference's own tolerance
    (tests/envs/<env>/test_<env>.py: allclose(rtol 1e-16 | 1e-8, atol 1e-8))."""
    from conftest import golden_rtol

    g = golden[env_name]
    B = 64
    env, props, keep, spec = make_env(env_name, B, torch.float64)
    env.sim_ahead_semantics = "step"
    env.traj_layout = "env_major"
    env.launch_opts = _native.launch_opts(env_major_mode=3)
    obs0 = torch.as_tensor(np.repeat(g["observations"][:1], B, axis=0), device=env.device)
    state = env.vmap_generate_state_from_observation(obs0)
    acts = torch.as_tensor(np.repeat(g["actions"][None], B, axis=0), device=env.device)
    obs, states, last = env.vmap_sim_ahead(state, acts, env.tau, env.tau)
    assert _native.last_launch() == "sim_ahead_emr_kernel"
    got = obs.cpu().numpy()
    assert obs.is_contiguous() and got.shape == (B,) + g["observations"].shape
    assert np.allclose(got[0][1:], g["observations"][1:], rtol=golden_rtol(env_name), atol=1e-8)
    assert np.allclose(got[B - 1][1:], g["observations"][1:], rtol=golden_rtol(env_name), atol=1e-8)
    assert np.array_equal(got, np.repeat(got[:1], B, axis=0))
    for n in env.STATE_FIELDS:
        assert torch.equal(getattr(last.physical_state, n), getattr(states.physical_state, n)[:, -1])


@pytest.mark.parametrize("obs_only", [False, True])
@pytest.mark.parametrize("semantics", ["step", "ahead"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_register_ring_matches_the_oracle_at_65536_environments(dtype, semantics, obs_only):
    """PMSM, default options at a batch the heuristic gives to the register-ring kernel: observations, state trajectories and the
    last state of three slices of the batch (first, last and a middle one that straddles waves) against the oracle's trajectory
    from the same inputs, at the tolerances of test_vmap_sim_ahead_matches_oracle."""
    import oracle
    from helpers import ANGLE_STATES, circ_close, max_err

    B, K = 65536, 64
    env, props, keep, spec = make_env("pmsm", B, dtype)
    env.sim_ahead_semantics = semantics
    env.traj_layout = "env_major"
    env.store_state_trajectory = not obs_only
    st = random_state("pmsm", B, NP_DTYPE[dtype], spec, seed=47)
    acts_np = np.random.default_rng(48).uniform(-1, 1, (B, K, 2)).astype(NP_DTYPE[dtype])
    obs, states, last = env.vmap_sim_ahead(to_state(env, st), torch.as_tensor(acts_np, device=env.device), env.tau, env.tau)
    assert _native.last_launch() == "sim_ahead_emr_kernel" and obs.is_contiguous()
    sem = oracle.SEM_STEP if semantics == "step" else oracle.SEM_AHEAD
    rtol, atol = (1e-5, 1e-5) if dtype is torch.float32 else (1e-9, 1e-9)
    n = 1024
    for sl in (slice(0, n), slice(B - n, B), slice(B // 2 - 37, B // 2 - 37 + n)):
        sp, sk = oracle.make_props("pmsm", spec["params"], spec["phys_norm"], spec["act_norm"], NP_DTYPE[dtype], n)
        o_ref, s_ref, l_ref = oracle.sim_ahead("pmsm", "euler", [s[sl] for s in st], acts_np[sl], sp, spec["tau"], semantics=sem)
        got = obs[sl].cpu().numpy()
        assert np.allclose(got, o_ref, rtol=rtol, atol=atol), max_err(got, o_ref)
        for j, name in enumerate(env.STATE_FIELDS):
            gl = getattr(last.physical_state, name)[sl].cpu().numpy()
            scale = max(1.0, float(np.nanmax(np.abs(s_ref[j]))))
            if states is not None:
                gs = getattr(states.physical_state, name)[sl].cpu().numpy()
                assert np.array_equal(gs[:, -1], gl)
                if j in ANGLE_STATES["pmsm"]:
                    assert circ_close(gs[..., None], s_ref[j][..., None], [0], rtol, atol * scale, period=2 * np.pi), name
                else:
                    assert np.allclose(gs, s_ref[j], rtol=rtol, atol=atol * scale), (name, max_err(gs, s_ref[j]))
            else:
                ref_last = l_ref[j]
                if j in ANGLE_STATES["pmsm"]:
                    assert circ_close(gl[:, None, None], np.asarray(ref_last)[:, None, None], [0], rtol, atol * scale, period=2 * np.pi), name
                else:
                    assert np.allclose(gl, ref_last, rtol=rtol, atol=atol * scale), name
