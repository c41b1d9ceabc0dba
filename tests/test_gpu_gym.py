"""reward / truncated / terminated fused into the step (excenv_gym_step) and the GymWrapper
mirror. The reference pins these only by shape (tests/test_gym_wrapper.py:19-48); here the fused kernel is
checked against the CPU oracle's literal restatement and against the package's torch mirrors. ``-m gpu``."""
import numpy as np
import pytest
import torch

import exciting_environments_amd as excenvs
import oracle
from conftest import ENV_NAMES
from helpers import NP_DTYPE, TRIG_FREE, make_env, random_state, spec_of, to_state

pytestmark = pytest.mark.gpu

CONTROL = {
    "pendulum": ["theta", "omega"], "mass_spring_damper": ["velocity"], "cartpole": ["theta", "deflection"],
    "acrobot": ["theta_2", "omega_1", "theta_1"], "fluid_tank": ["height"], "pmsm": ["i_d", "i_q", "torque"],
}


def _problem(env_name, B, dtype, control_state, seed):
    env, props, keep, spec = make_env(env_name, B, dtype, control_state=control_state)
    st = random_state(env_name, B, NP_DTYPE[dtype], spec, seed=seed)
    rng = np.random.default_rng(seed + 1)
    for j in range(len(st)):  # push a third of the envs out of the normalisation box so truncation fires
        if env_name != "pmsm" or oracle.STATE_FIELDS[env_name][j] in ("i_d", "i_q"):
            st[j][::3] = st[j][::3] * NP_DTYPE[dtype](1.4)
    refs = {}
    for n in control_state:
        lo, hi = spec["phys_norm"][n]
        refs[n] = ((rng.uniform(-0.8, 0.8, B) + 1) / 2 * (hi - lo) + lo).astype(NP_DTYPE[dtype])
    act = rng.uniform(-1, 1, (B, env.action_dim)).astype(NP_DTYPE[dtype])
    return env, props, spec, st, refs, act


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("with_control", [True, False])
@pytest.mark.parametrize("env_name", ENV_NAMES)
def test_fused_gym_step_matches_oracle(env_name, with_control, dtype):
    B = 3000
    cs = CONTROL[env_name] if with_control else []
    env, props, spec, st, refs, act = _problem(env_name, B, dtype, cs, seed=201)
    state = to_state(env, st, reference=refs)
    obs, reward, terminated, truncated, new = env.vmap_gym_step(state, torch.as_tensor(act, device=env.device))
    control = [(n, refs[n]) for n in cs]
    o_ref, s_ref, r_ref, te_ref, tr_ref = oracle.gym_step(env_name, "euler", st, act, props, spec["tau"], control=control)
    assert reward.shape == (B, 1) and terminated.shape == (B, 1) and truncated.shape == tr_ref.shape
    assert terminated.dtype == torch.bool and truncated.dtype == torch.bool
    tol = 0.0 if env_name in TRIG_FREE else (1e-9 if dtype == torch.float64 else 2e-5)
    if tol == 0.0:
        assert np.array_equal(reward.cpu().numpy(), r_ref) and np.array_equal(obs.cpu().numpy(), o_ref)
    else:
        assert np.allclose(reward.cpu().numpy(), r_ref, rtol=tol, atol=tol)
    assert np.array_equal(terminated.cpu().numpy(), te_ref)
    assert np.array_equal(truncated.cpu().numpy(), tr_ref)
    if env_name not in ("fluid_tank",):
        assert bool(truncated.any()) and not bool(truncated.all())
    # the fused outputs equal vmap_step + the package's torch mirrors of the reference functions
    obs2, new2 = env.vmap_step(state, torch.as_tensor(act, device=env.device))
    assert torch.equal(obs, obs2)
    for n in env.STATE_FIELDS:
        assert torch.equal(getattr(new.physical_state, n), getattr(new2.physical_state, n))
    r_t = env.generate_reward(new2, None, env.env_properties)
    assert r_t.shape == (B, 1) and torch.allclose(r_t, reward, rtol=1e-4 if dtype == torch.float32 else 1e-9, atol=1e-5 if dtype == torch.float32 else 1e-9)
    assert torch.equal(env.generate_truncated(new2, env.env_properties), truncated)
    if not with_control and env_name not in ("pmsm", "fluid_tank"):
        assert bool(terminated.all())  # reward == 0 without control_state (e.g. pendulum_env.py:387-390)


@pytest.mark.parametrize("env_name", ENV_NAMES)
def test_gym_wrapper_step_returns_correct_outputs(env_name):
    """reference tests/test_gym_wrapper.py:19-36."""
    from exciting_environments_amd import GymWrapper

    env, props, keep, spec = make_env(env_name, 4, torch.float32)
    gym_env = GymWrapper(env=env)
    action = torch.ones((env.batch_size, env.action_dim), device=env.device)
    _, state = env.vmap_reset()
    new_obs, state = env.vmap_step(state, action)
    _ = gym_env.reset()
    new_obs_gym, reward, terminated, truncated = gym_env.step(action)
    assert torch.equal(new_obs, new_obs_gym), "gym_step generates different observation compared to standalone env"
    assert reward.shape == (4, 1), "Unexpected reward shape"
    assert terminated.shape == (4, 1), "Unexpected terminated shape"
    for _ in range(5):
        o, r, te, tr = gym_env.step(action)
    assert o.shape == new_obs.shape


@pytest.mark.parametrize("env_name", ENV_NAMES)
def test_gym_wrapper_ref_generation(env_name):
    """reference tests/test_gym_wrapper.py:39-48 (+ the generated reference shows up in obs and reward)."""
    from exciting_environments_amd import GymWrapper

    env, props, keep, spec = make_env(env_name, 4, torch.float32)
    gym_env = GymWrapper(env=env, control_state=CONTROL[env_name])
    obs, _ = gym_env.reset(rng_env=0, rng_ref=1)
    assert gym_env.ref_gen is True
    assert gym_env.reference_hold_steps.shape == (env.batch_size, 1)
    assert obs.shape == (4, len(env.obs_description)) and bool(torch.isfinite(obs).all())
    hold0 = gym_env.reference_hold_steps.clone()
    action = torch.zeros((4, env.action_dim), device=env.device)
    o, r, te, tr = gym_env.step(action)
    assert torch.equal(gym_env.reference_hold_steps, hold0 - 1)
    assert bool(torch.isfinite(r).all()) and r.shape == (4, 1)


def test_vmap_generate_rew_trunc_term_ahead_shapes():
    """core_env.py:618-647: rewards on rows 1.., truncated on all rows."""
    env, props, keep, spec = make_env("pendulum", 8, torch.float32, control_state=["theta"])
    _, state = env.vmap_reset()
    state.reference.theta = torch.full((8,), 0.5, device=env.device)
    acts = torch.ones((8, 10, 1), device=env.device)
    obs, states, last = env.vmap_sim_ahead(state, acts, env.tau, env.tau)
    reward, truncated, terminated = env.vmap_generate_rew_trunc_term_ahead(states, acts)
    assert reward.shape == (8, 10, 1) and truncated.shape == (8, 11, 3) and terminated.shape == (8, 10, 1)
    th = states.physical_state.theta[:, 1:]
    want = -((torch.sin(th) - np.sin(0.5)) ** 2 + (torch.cos(th) - np.cos(0.5)) ** 2)
    assert torch.allclose(reward[..., 0], want, atol=1e-6)


# ------------------------------------------------------------ trajectories: fused into the sim_ahead launch / stored
@pytest.mark.parametrize("layout", ["lane_major", "env_major"])
@pytest.mark.parametrize("semantics", ["step", "ahead"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("env_name", ENV_NAMES)
def test_sim_ahead_fused_rew_trunc_term_matches_oracle(env_name, dtype, semantics, layout):
    """vmap_sim_ahead(..., return_rew_trunc_term=True): reward / truncated / terminated trajectories out of the trajectory
    launch itself (core_env.py:490-531) vs the oracle's literal restatement evaluated on the returned states, and vs the
    stand-alone launch behind vmap_generate_rew_trunc_term_ahead (same device function: identical bits)."""
    B, K = 700, 24
    cs = CONTROL[env_name]
    env, props, spec, st, refs, _ = _problem(env_name, B, dtype, cs, seed=231)
    env.sim_ahead_semantics, env.traj_layout = semantics, layout
    acts = np.random.default_rng(232).uniform(-1, 1, (B, K, env.action_dim)).astype(NP_DTYPE[dtype])
    state = to_state(env, st, reference=refs)
    a_dev = torch.as_tensor(acts, device=env.device)
    obs, states, last, reward, truncated, terminated = env.vmap_sim_ahead(state, a_dev, env.tau, env.tau, return_rew_trunc_term=True)
    obs2, states2, last2 = env.vmap_sim_ahead(state, a_dev, env.tau, env.tau)
    assert torch.equal(obs, obs2)  # asking for the extra outputs does not change the trajectory
    TW = 1 if env_name in ("pmsm", "fluid_tank") else obs.shape[-1]
    assert reward.shape == (B, K, 1) and terminated.shape == (B, K, 1) and truncated.shape == (B, K + 1, TW)
    if layout == "lane_major":  # ABI 7 (include/excenv.h): reward / terminated [row][B], truncated [row][B][flag] — views of exactly that
        assert tuple(truncated.stride()) == (TW, B * TW, 1) and tuple(reward.stride())[:2] == (1, B) and tuple(terminated.stride())[:2] == (1, B)
    else:
        assert truncated.is_contiguous() and reward.is_contiguous() and terminated.is_contiguous()
    st_np = [getattr(states.physical_state, n).cpu().numpy() for n in env.STATE_FIELDS]
    control = [(n, refs[n]) for n in cs]
    r_ref, tr_ref, te_ref = oracle.rew_trunc_term_ahead(env_name, st_np, props, control=control)
    tol = 0.0 if env_name in TRIG_FREE else (1e-9 if dtype == torch.float64 else 2e-5)
    if tol == 0.0:
        assert np.array_equal(reward.cpu().numpy(), r_ref)
    else:
        assert np.allclose(reward.cpu().numpy(), r_ref, rtol=tol, atol=tol)
    assert np.array_equal(truncated.cpu().numpy(), tr_ref) and np.array_equal(terminated.cpu().numpy(), te_ref)
    if env_name != "fluid_tank":
        assert bool(truncated.any()) and not bool(truncated.all())
    r2, tr2, te2 = env.vmap_generate_rew_trunc_term_ahead(states, a_dev)
    assert torch.equal(r2, reward) and torch.equal(tr2, truncated) and torch.equal(te2, terminated)
    # contiguous copies of the state leaves (the reference's row-major arrays) take the other index order in the kernel
    import dataclasses
    states_c = dataclasses.replace(states, physical_state=env.PhysicalState(
        **{n: getattr(states.physical_state, n).contiguous() for n in env.STATE_FIELDS}))
    r3, tr3, te3 = env.vmap_generate_rew_trunc_term_ahead(states_c, a_dev)
    assert torch.equal(r3, reward) and torch.equal(tr3, truncated) and torch.equal(te3, terminated)


def test_rew_trunc_term_ahead_without_control_and_per_env_properties():
    B, K = 300, 9
    spec = spec_of("pendulum")
    spec["params"]["l"] = np.random.default_rng(3).uniform(1.0, 3.0, B)
    spec["phys_norm"]["omega"] = (-np.random.default_rng(4).uniform(5, 12, B), np.random.default_rng(4).uniform(5, 12, B))
    env, props, keep, spec = make_env("pendulum", B, torch.float32, spec=spec)
    st = random_state("pendulum", B, np.float32, spec_of("pendulum"), seed=5)
    st[1][::2] *= 3.0
    acts = torch.as_tensor(np.random.default_rng(6).uniform(-1, 1, (B, K, 1)).astype(np.float32), device=env.device)
    obs, states, last, reward, truncated, terminated = env.vmap_sim_ahead(to_state(env, st), acts, env.tau, env.tau,
                                                                        return_rew_trunc_term=True)
    assert bool((reward == 0).all()) and bool(terminated.all())  # empty control_state: reward 0, terminated = (reward == 0)
    assert torch.equal(truncated, obs.abs() > 1) and bool(truncated.any())
    st_np = [getattr(states.physical_state, n).cpu().numpy() for n in env.STATE_FIELDS]
    r_ref, tr_ref, te_ref = oracle.rew_trunc_term_ahead("pendulum", st_np, props)
    assert np.array_equal(truncated.cpu().numpy(), tr_ref) and np.array_equal(terminated.cpu().numpy(), te_ref)


# ------------------------------------------------------------ generate_state_from_observation on the device
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("env_name", ENV_NAMES)
def test_state_from_observation_kernel_round_trip_and_oracle(env_name, dtype):
    """vmap_generate_state_from_observation as one HIP launch: the reference's obs -> state -> obs round trip
    (tests/envs/test_core_functions.py:55-77) and the oracle's restatement (pendulum_env.py:331-364, pmsm_env.py:921-970)."""
    B = 1037
    cs = CONTROL[env_name][:2]
    env, props, spec, st, refs, act = _problem(env_name, B, dtype, cs, seed=241)
    state = to_state(env, st, reference=refs)
    obs = env.generate_observation(state, env.env_properties)
    assert obs.is_cuda and obs.shape == (B, len(env.obs_description))
    back = env.vmap_generate_state_from_observation(obs)
    assert type(back) == env.State and bool(torch.isnan(back.PRNGKey).all()) and not bool(back.additions.active_solver_state.any())
    want = oracle.state_from_observation(env_name, obs.cpu().numpy(), spec["phys_norm"])
    for j, n in enumerate(env.STATE_FIELDS):
        got = getattr(back.physical_state, n).cpu().numpy()
        if env_name == "pmsm" and n == "epsilon":  # atan2 of (sin, cos): device library vs libm
            tol = 1e-12 if dtype == torch.float64 else 2e-6
            assert np.allclose(got, want[j], rtol=0, atol=tol)
            assert np.allclose(got, st[j], rtol=0, atol=1e-9 if dtype == torch.float64 else 5e-6)
        else:
            assert np.array_equal(got, want[j]), n
    for n in env.STATE_FIELDS:
        r = getattr(back.reference, n)
        if n in cs:
            lo, hi = spec["phys_norm"][n]
            pos = len(env.STATE_FIELDS) + cs.index(n) if env_name != "pmsm" else 8 + cs.index(n)
            assert np.array_equal(r.cpu().numpy(), oracle.denormalize(obs[:, pos].cpu().numpy(), NP_DTYPE[dtype](lo), NP_DTYPE[dtype](hi))), n
        else:
            assert bool(torch.isnan(r).all())
    obs2 = env.generate_observation(back, env.env_properties)
    tol = 1e-12 if dtype == torch.float64 else 2e-6
    assert torch.allclose(obs2, obs, rtol=0, atol=tol)
    # the torch mirror (single observation, CPU) agrees with the kernel
    one = env.generate_state_from_observation(obs[5], env.env_properties)
    for n in env.STATE_FIELDS:
        assert torch.allclose(getattr(one.physical_state, n), getattr(back.physical_state, n)[5], rtol=0, atol=tol * 1e3)


@pytest.mark.parametrize("key_mode", [True, False])
@pytest.mark.parametrize("env_name", ["pendulum", "pmsm", "cartpole"])
def test_gym_wrapper_ref_generation_fast_path_equals_literal_path(env_name, key_mode):
    """With the reference generator armed the wrapper has three executions of a step: (key mode, device) excenv_update_ref +
    the fused gym launch; the fused launch alone while the host mirror of the smallest hold counter says nobody is due; the
    literal vmap_step -> update_ref -> reward path. Same outputs, states, references, keys and counters over several redraws
    (hold steps 2..4, so a redraw is due every few steps). Rewards (fused kernel vs torch mirror) and, for PMSM, everything
    downstream of jax.random.ball (device libm vs torch erfinv / log) are compared with a tolerance."""
    from exciting_environments_amd import GymWrapper
    from exciting_environments_amd import random as jr

    B, steps = 64, 40
    variants = [(True, True), (False, True), (False, False)] if key_mode else [(False, True), (False, False)]
    runs = []
    for device_refgen, mirror in variants:
        env, props, keep, spec = make_env(env_name, B, torch.float32)
        gw = GymWrapper(env=env, control_state=CONTROL[env_name], ref_params={"hold_steps_min": 2, "hold_steps_max": 5})
        gw.host_hold_mirror, gw.device_update_ref = mirror, device_refgen
        rng_ref = jr.PRNGKey(3, device=env.device) if key_mode else 3
        rng_env = jr.split(jr.PRNGKey(4, device=env.device), B) if key_mode else 4
        obs, _ = gw.reset(rng_env=rng_env, rng_ref=rng_ref)
        acts = torch.as_tensor(np.random.default_rng(5).uniform(-1, 1, (steps, B, env.action_dim)).astype(np.float32), device=env.device)
        out = [("obs", obs)]
        for k in range(steps):
            o, r, te, tr = gw.step(acts[k])
            out += [("obs", o), ("reward", r), ("flag", te.to(torch.float32)), ("flag", tr.to(torch.float32)),
                    ("hold", gw.reference_hold_steps.to(torch.float32).clone())]
        out += [("ref", getattr(gw.state.reference, n).clone()) for n in CONTROL[env_name]]
        out += [("state", getattr(gw.state.physical_state, n).clone()) for n in env.STATE_FIELDS]
        if key_mode:
            out.append(("key", gw.state.PRNGKey.to(torch.float64)))
        runs.append(out)
    ball = env_name == "pmsm" and key_mode  # references drawn through the gamma-rejection ball: libm-level differences
    for run in runs[:-1]:
        assert len(run) == len(runs[-1])
        for i, ((kind, a), (_, b)) in enumerate(zip(run, runs[-1])):
            if kind in ("hold", "key", "state") or (kind in ("obs", "ref") and not ball):
                assert torch.equal(a, b), (i, kind)
            elif kind == "flag":
                assert float((a != b).float().mean()) <= (0.02 if ball else 0.0), (i, kind)
            else:
                assert torch.allclose(a, b, rtol=2e-5, atol=2e-5), (i, kind, float((a - b).abs().max()))
    holds = torch.stack([t for kind, t in runs[0] if kind == "hold"])  # the counters really ran down and were redrawn
    assert int(holds.min()) >= 0 and float(holds.max()) <= 4 and bool((holds[1:] > holds[:-1]).any())


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("env_name", ENV_NAMES)
def test_update_ref_kernel_matches_the_host_twin(env_name, dtype):
    """excenv_update_ref (threefry split / uniform / randint / ball in one launch) against the torch restatement in
    random.py + GymWrapper.generate_new_ref on the same keys: keys and hold counters bit for bit, references bit for bit for
    the uniform-drawn environments and to libm accuracy for PMSM's ball-drawn currents."""
    from exciting_environments_amd import GymWrapper, _native
    from exciting_environments_amd import random as jr

    B = 4096
    env, props, keep, spec = make_env(env_name, B, dtype)
    cs = CONTROL[env_name]
    gw = GymWrapper(env=env, control_state=cs, ref_params={"hold_steps_min": 10, "hold_steps_max": 1000})
    gw.device_update_ref = False
    keys = jr.split(jr.PRNGKey(77, device=env.device), B)
    _, state = env.vmap_reset()
    import dataclasses
    refs0 = {n: torch.full((B,), 0.25, dtype=dtype, device=env.device) for n in cs}
    state = dataclasses.replace(state, PRNGKey=keys, reference=env.PhysicalState(
        **{n: refs0.get(n, getattr(state.reference, n)) for n in env.STATE_FIELDS}))
    hold = torch.as_tensor(np.random.default_rng(1).integers(0, 3, (B, 1)), device=env.device)  # about a third are due
    want_state, want_hold = gw.update_ref(state, hold)
    new_refs = [refs0[n].clone() for n in cs]
    k2, h2 = keys.clone(), hold.reshape(B).clone()
    p, _k = env._props_for(env.env_properties, B)
    _native.update_ref(env.ENV_ID, dtype, B, p, [env.STATE_FIELDS.index(n) for n in cs], new_refs, k2, h2, 10, 1000)
    assert torch.equal(h2, want_hold.reshape(B)) and torch.equal(k2, want_state.PRNGKey)
    due = hold.reshape(B) == 0
    assert 0.2 < float(due.float().mean()) < 0.5 and int(h2[due].min()) >= 9 and int(h2[due].max()) < 999
    for n, got in zip(cs, new_refs):
        want = getattr(want_state.reference, n)
        assert torch.equal(got[~due], refs0[n][~due])
        if env_name == "pmsm":
            tol = 1e-9 if dtype == torch.float64 else 2e-5
            close = torch.isclose(got, want, rtol=tol, atol=tol * 250)
            assert float(close.float().mean()) > 0.999, n  # a rejection decided differently by one ulp changes the sample
        else:
            assert torch.equal(got, want), n


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("reg", list(excenvs.EnvironmentRegistry), ids=lambda r: r.name.lower())
def test_random_state_kernel_equals_the_host_twin(reg, dtype):
    """excenv_random_state (vmap_init_state with one key per environment) against random.py on the CPU: the key leaf is
    integer work and exact; uniform-drawn states go through the same three roundings and are exact too; the PMSM currents
    come out of erf_inv / log / pow whose device versions differ in the last place."""
    from exciting_environments_amd import random as jr

    B = 1003  # ragged
    keys = jr.split(jr.PRNGKey(77), B)
    cpu = reg.make(batch_size=B, device="cpu", dtype=dtype)
    env = reg.make(batch_size=B, device="cuda", dtype=dtype)
    s_c = cpu.vmap_init_state(keys)
    s_g = env.vmap_init_state(keys.cuda())
    assert torch.equal(s_g.PRNGKey.cpu(), s_c.PRNGKey) and s_g.PRNGKey.dtype == s_c.PRNGKey.dtype
    for n in env.STATE_FIELDS:
        g, c = getattr(s_g.physical_state, n).cpu(), getattr(s_c.physical_state, n)
        assert g.shape == c.shape and g.dtype == c.dtype
        if reg is excenvs.EnvironmentRegistry.PMSM:
            tol = 2e-5 if dtype == torch.float32 else 1e-12
            assert torch.allclose(g, c, rtol=tol, atol=tol * 250.0), (n, float((g - c).abs().max()))
        else:
            assert torch.equal(g, c), (n, float((g - c).abs().max()))
        assert bool(torch.isnan(getattr(s_g.reference, n)).all())
    assert not bool(s_g.additions.active_solver_state.any())
    # the state is usable by the hot path as is (leaf alignment, contiguity)
    obs, _ = env.vmap_step(s_g, torch.zeros(B, env.action_dim, device="cuda", dtype=dtype))
    assert obs.shape[0] == B and bool(torch.isfinite(obs[:, : len(env.STATE_FIELDS)]).all())
    # CPU keys on a device environment take the same kernel
    s_g2 = env.vmap_init_state(keys)
    assert torch.equal(s_g2.PRNGKey, s_g.PRNGKey) and torch.equal(getattr(s_g2.physical_state, env.STATE_FIELDS[0]), getattr(s_g.physical_state, env.STATE_FIELDS[0]))


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("reg", list(excenvs.EnvironmentRegistry), ids=lambda r: r.name.lower())
def test_generate_observation_kernel_equals_the_torch_twin(reg, dtype):
    """excenv_observe (generate_observation of a [B] batch on the device, one launch) against the elementwise torch twin the
    CPU environments use: identical arithmetic for the normalised fields (bit-equal), sin / cos of the PMSM angle to 2 ulp;
    controlled-reference columns incl. NaN references; a per-environment property array."""
    from dataclasses import replace

    B = 777
    ctl = {"PMSM": ["i_d", "torque"], "PENDULUM": ["theta"], "ACROBOT": ["theta_2", "omega_1"]}.get(reg.name, [])
    pn = None
    if reg.name == "MASS_SPRING_DAMPER":  # a batched normalisation bound
        pn = {"deflection": excenvs.MinMaxNormalization(min=-10 * np.ones(B), max=np.linspace(5, 15, B)),
              "velocity": excenvs.MinMaxNormalization(min=-10, max=10)}
    kw = dict(batch_size=B, dtype=dtype, control_state=ctl)
    if pn is not None:
        kw["physical_normalizations"] = pn
    env = reg.make(device="cuda", **kw)
    cpu = reg.make(device="cpu", **kw)
    g = torch.Generator().manual_seed(21)
    _, st = cpu.vmap_reset()
    phys = {n: (torch.rand(B, generator=g, dtype=dtype) - 0.5) * 6 for n in cpu.STATE_FIELDS}
    ref = {n: torch.full((B,), float("nan"), dtype=dtype) for n in cpu.STATE_FIELDS}
    for n in ctl:
        ref[n] = (torch.rand(B, generator=g, dtype=dtype) - 0.5) * 4
        ref[n][::7] = float("nan")
    st = replace(st, physical_state=cpu.PhysicalState(**phys), reference=cpu.PhysicalState(**ref))
    want = cpu.generate_observation(st, cpu.env_properties)
    st_g = replace(st, physical_state=env.PhysicalState(**{n: v.cuda() for n, v in phys.items()}),
                   reference=env.PhysicalState(**{n: v.cuda() for n, v in ref.items()}))
    got = env.generate_observation(st_g, env.env_properties).cpu()
    assert got.shape == want.shape == (B, len(env.obs_description)) and got.dtype == want.dtype
    if reg.name == "PMSM":
        tol = 3e-7 if dtype == torch.float32 else 1e-15
        # fp32: the twin folds (max - min) of the Python-float bounds in double (like the reference's static PMSM properties),
        # the kernels subtract the fp32 bounds: one ulp of the denominator for bounds such as 2 * 400 / 3
        assert torch.allclose(got, want, rtol=0, atol=tol, equal_nan=True), float((got - want).abs().nan_to_num().max())
    else:
        assert torch.equal(got.nan_to_num(7.0), want.nan_to_num(7.0)), float((got - want).abs().nan_to_num().max())
    # vmap_reset goes through the same launch
    obs_r, st_r = env.vmap_reset()
    obs_c, _ = cpu.vmap_reset()
    assert torch.allclose(obs_r.cpu(), obs_c, rtol=0, atol=3e-7, equal_nan=True)


@pytest.mark.parametrize("kind", ["float64_leaf", "cpu_leaf"])
def test_reference_leaf_that_needs_conversion_is_reread_after_in_place_update(kind):
    """A reference leaf that is not already a device tensor of the env dtype is converted to a copy for the launch. The copy
    must not be kept under the identity of the original: an in-place update of the original between two calls keeps its id."""
    B = 512
    env, props, spec, st, refs, act = _problem("mass_spring_damper", B, torch.float32, ["velocity"], seed=77)
    state = to_state(env, st, reference=refs)
    if kind == "float64_leaf":
        leaf = torch.as_tensor(refs["velocity"], dtype=torch.float64, device=env.device)
    else:
        leaf = torch.as_tensor(refs["velocity"], dtype=torch.float32, device="cpu")
    state.reference.velocity = leaf
    a = torch.as_tensor(act, device=env.device)
    obs1, rew1, _, _, _ = env.vmap_gym_step(state, a)
    obs1, rew1 = obs1.clone(), rew1.clone()
    leaf.mul_(0.25)  # same object, same id, new values
    obs2, rew2, _, _, _ = env.vmap_gym_step(state, a)
    fresh = to_state(env, st, reference={"velocity": leaf.cpu().numpy().astype(np.float32)})
    obs3, rew3, _, _, _ = env.vmap_gym_step(fresh, a)
    assert torch.equal(obs2, obs3) and torch.equal(rew2, rew3)
    assert not torch.equal(rew1, rew2) and not torch.equal(obs1[:, -1], obs2[:, -1])
    # the trajectory path's broadcast reference views: same rule
    acts = torch.zeros((B, 3, env.action_dim), device=env.device)
    _, states1, _ = env.vmap_sim_ahead(state, acts, env.tau, env.tau)
    r1 = states1.reference.velocity.clone()
    leaf.mul_(2.0)
    _, states2, _ = env.vmap_sim_ahead(state, acts, env.tau, env.tau)
    assert torch.equal(states2.reference.velocity[:, 0].cpu(), leaf.to(torch.float32).cpu())
    assert not torch.equal(r1, states2.reference.velocity)


@pytest.mark.parametrize("control", [[], ["i_d", "i_q"], ["torque"], ["i_q", "torque", "i_d"]])
@pytest.mark.parametrize("semantics", ["ahead", "step"])
@pytest.mark.parametrize("dtype,solver", [(torch.float32, "euler"), (torch.float32, "tsit5"), (torch.float64, "euler"), (torch.float64, "rk4")])
def test_pmsm_gym_trajectories_from_the_wide_lean_kernel_equal_the_general_kernel(dtype, solver, semantics, control):
    """Round 4 (kernels.hpp, LGYM): PMSM's reward / terminated / truncated trajectories written by the four-environments-per-lane
    kernel (packed flag stores, pmsm_reward shared with the general instantiation) hold the bits of the one-environment general
    kernel — observations (incl. control columns), states, last state, reward, flags — and the launch that ran is the lean one."""
    from exciting_environments_amd import _native

    B, K = 2048, 37
    vmax = 4 if dtype is torch.float32 else 2
    env, props, keep, spec = make_env("pmsm", B, dtype, solver=solver, control_state=list(control))
    env.sim_ahead_semantics = semantics
    st = random_state("pmsm", B, NP_DTYPE[dtype], spec, seed=91)
    st[3] = (st[3] * 1.4).astype(NP_DTYPE[dtype])  # some environments outside the current circle: the flags are not all zero
    rng = np.random.default_rng(92)
    refs = {n: rng.uniform(-150, 150, B).astype(NP_DTYPE[dtype]) for n in control}
    acts = env.new_actions_buffer(K)
    acts.copy_(torch.as_tensor(rng.uniform(-1, 1, (B, K, 2)).astype(NP_DTYPE[dtype]), device=env.device))
    outs = {}
    for vec in (vmax, 1):
        env.launch_opts = _native.launch_opts(envs_per_lane=vec)
        outs[vec] = env.vmap_sim_ahead(to_state(env, st, reference=refs), acts, env.tau, env.tau, return_rew_trunc_term=True)
        torch.cuda.synchronize()
        assert _native.last_launch() == ("sim_ahead_kernel (lean, gym outputs)" if vec == vmax else "sim_ahead_kernel (general)")
    a, b = outs[vmax], outs[1]
    assert torch.equal(a[0], b[0])
    for n in env.STATE_FIELDS:
        assert torch.equal(getattr(a[1].physical_state, n), getattr(b[1].physical_state, n)), n
        assert torch.equal(getattr(a[2].physical_state, n), getattr(b[2].physical_state, n)), n
    for k, name in ((3, "reward"), (4, "truncated"), (5, "terminated")):
        assert a[k].shape == b[k].shape and a[k].dtype == b[k].dtype and torch.equal(a[k], b[k]), name
    assert a[4].any() and not a[4].all()
    if control:
        assert float(a[3].abs().max()) > 0


@pytest.mark.parametrize("env_name,control", [("pendulum", ["theta"]), ("pendulum", ["omega", "theta"]), ("mass_spring_damper", ["deflection"]),
                                              ("cartpole", ["theta", "velocity", "deflection"]), ("acrobot", ["theta_2", "omega_1", "theta_1", "omega_2"]),
                                              ("fluid_tank", ["height"]), ("fluid_tank", []), ("cartpole", [])])
@pytest.mark.parametrize("semantics", ["ahead", "step"])
@pytest.mark.parametrize("dtype,solver", [(torch.float32, "euler"), (torch.float32, "rk4"), (torch.float64, "euler"), (torch.float64, "tsit5")])
def test_gym_trajectories_from_the_wide_lean_kernel_equal_the_general_kernel(env_name, control, dtype, solver, semantics):
    """The other five models' reward / terminated / truncated trajectories out of the widest lean kernel (LGYM: references' sin / cos
    and normalised values computed once per trajectory, packed flag stores per observation column) hold the bits of the general
    one-environment kernel."""
    from exciting_environments_amd import _native

    B, K = 1024, 29
    vmax = 4 if dtype is torch.float32 else 2
    env, props, keep, spec = make_env(env_name, B, dtype, solver=solver, control_state=list(control))
    env.sim_ahead_semantics = semantics
    st = random_state(env_name, B, NP_DTYPE[dtype], spec, seed=93)
    st[0] = (st[0] * 1.3).astype(NP_DTYPE[dtype])  # some states outside the normalisation box: truncated flags of both kinds
    rng = np.random.default_rng(94)
    refs = {}
    for n in control:
        lo, hi = spec["phys_norm"][n]
        refs[n] = rng.uniform(1.2 * lo if lo < 0 else lo, 1.2 * hi, B).astype(NP_DTYPE[dtype])  # some references outside it too
    acts = env.new_actions_buffer(K)
    acts.copy_(torch.as_tensor(rng.uniform(-1, 1, (B, K, 1)).astype(NP_DTYPE[dtype]), device=env.device))
    outs = {}
    for vec in (vmax, 1):
        env.launch_opts = _native.launch_opts(envs_per_lane=vec)
        outs[vec] = env.vmap_sim_ahead(to_state(env, st, reference=refs), acts, env.tau, env.tau, return_rew_trunc_term=True)
        torch.cuda.synchronize()
        wide = vec == vmax and not (dtype is torch.float64 and solver != "euler" and env_name in ("cartpole", "acrobot"))
        assert _native.last_launch() == ("sim_ahead_kernel (lean, gym outputs)" if wide else "sim_ahead_kernel (general)")
    a, b = outs[vmax], outs[1]
    assert torch.equal(a[0], b[0])
    for n in env.STATE_FIELDS:
        assert torch.equal(getattr(a[1].physical_state, n), getattr(b[1].physical_state, n)), n
    for k, name in ((3, "reward"), (4, "truncated"), (5, "terminated")):
        assert a[k].shape == b[k].shape and a[k].dtype == b[k].dtype and torch.equal(a[k], b[k]), name
    if env_name != "fluid_tank":
        assert a[4].any()


def test_default_options_acrobot_rk4_gym_trajectories_at_2pow18():
    """acrobot RK4 / Tsit5 fp32 with gym outputs, lane-major actions, DEFAULT launch options and a batch that takes
    the widest form (2^18) returned "internal error: lean gym outputs need 4 environments per lane" — the lane-width cap for acrobot's
    RK kernels applied after the lean gym form had been chosen. Such a call takes the general kernel now; compared with the forced
    lean form of the same call (bit-equal) on a batch slice against the oracle's flags."""
    from exciting_environments_amd import _native

    B, K = 1 << 18, 12
    env, props, keep, spec = make_env("acrobot", B, torch.float32, solver="rk4", control_state=["theta_1", "omega_2"])
    st = random_state("acrobot", B, np.float32, spec, seed=95)
    rng = np.random.default_rng(96)
    refs = {"theta_1": rng.uniform(-3, 3, B).astype(np.float32), "omega_2": rng.uniform(-20, 20, B).astype(np.float32)}
    acts = env.new_actions_buffer(K)
    acts.copy_(torch.as_tensor(rng.uniform(-1, 1, (B, K, 1)).astype(np.float32), device=env.device))
    outs = {}
    for vec in (0, 4):
        env.launch_opts = _native.launch_opts(envs_per_lane=vec)
        outs[vec] = env.vmap_sim_ahead(to_state(env, st, reference=refs), acts, env.tau, env.tau, return_rew_trunc_term=True)
        torch.cuda.synchronize()
        assert _native.last_launch() == ("sim_ahead_kernel (general)" if vec == 0 else "sim_ahead_kernel (lean, gym outputs)")
    a, b = outs[0], outs[4]
    assert torch.equal(a[0], b[0])
    for k in (3, 4, 5):
        assert torch.equal(a[k], b[k])
    # Euler at the same batch keeps the lean form with default options
    env2, _, _, _ = make_env("acrobot", B, torch.float32, solver="euler", control_state=["theta_1", "omega_2"])
    env2.vmap_sim_ahead(to_state(env2, st, reference=refs), acts, env2.tau, env2.tau, return_rew_trunc_term=True)
    assert _native.last_launch() == "sim_ahead_kernel (lean, gym outputs)"


@pytest.mark.parametrize("which", ["state_in", "state_traj", "obs", "reference"])
def test_gym_trajectories_with_a_misaligned_array_take_the_general_kernel(which):
    """ADVICE r04 (high), case 2: a state / trajectory / observation / reference array that is only 4-byte aligned cannot take the
    16-byte accesses of the lean gym form; the raw ABI call used to fail with EINVAL, now it runs the general kernel and gives the
    aligned call's bits."""
    from exciting_environments_amd import _native

    B, K = 1 << 17, 5
    env, props_o, keep, spec = make_env("pendulum", B, torch.float32, control_state=["theta"])
    dev = env.device
    rng = np.random.default_rng(97)
    props, pkeep = env._props_for(env.env_properties, B)

    def buf(n, off):  # a float32 array of n elements that starts `off` elements into a 16-byte aligned allocation
        return torch.zeros(n + 4, dtype=torch.float32, device=dev)[off:off + n]

    results = []
    for off in (0, 1):
        o = {k: (off if k == which else 0) for k in ("state_in", "state_traj", "obs", "reference")}
        st_in = [buf(B, o["state_in"]) for _ in range(2)]
        g = torch.Generator(device="cpu").manual_seed(5)
        st_in[0].copy_((torch.rand(B, generator=g) * 6 - 3).to(dev))
        st_in[1].copy_((torch.rand(B, generator=g) * 10 - 5).to(dev))
        ref = buf(B, o["reference"])
        ref.copy_((torch.rand(B, generator=g) * 6 - 3).to(dev))
        acts = torch.as_tensor(rng.uniform(-1, 1, (K, 1, B)).astype(np.float32), device=dev) if not results else results[0]["acts"]
        obs = buf((K + 1) * 4 * B, o["obs"])  # O = 3 observation columns + the control column
        straj = [buf((K + 1) * B, o["state_traj"]) for _ in range(2)]
        last = [torch.zeros(B, dtype=torch.float32, device=dev) for _ in range(2)]
        rew = torch.zeros((K, B), dtype=torch.float32, device=dev)
        term = torch.zeros((K, B), dtype=torch.bool, device=dev)
        trunc = torch.zeros((K + 1, B, 4), dtype=torch.bool, device=dev)  # excenv_truncated_width(pendulum, 1) = 4 flags per environment
        control = _native.make_control([0], [ref])
        _native.sim_ahead(env.ENV_ID, env._solver.id, torch.float32, B, K, 1, props, control, env.tau, env.tau, st_in, acts,
                          _native.LAYOUT_LANE_MAJOR, obs, straj, _native.LAYOUT_LANE_MAJOR, last, _native.SEM_AHEAD, None,
                          _native.launch_opts(envs_per_lane=4), (rew, term, trunc))  # the widest form asked for explicitly
        torch.cuda.synchronize()
        assert _native.last_launch() == ("sim_ahead_kernel (general)" if off else "sim_ahead_kernel (lean, gym outputs)")
        results.append(dict(acts=acts, obs=obs.clone(), straj=[s.clone() for s in straj], rew=rew, term=term, trunc=trunc))
    a, b = results
    assert torch.equal(a["obs"], b["obs"]) and torch.equal(a["rew"], b["rew"]) and torch.equal(a["term"], b["term"])
    assert torch.equal(a["trunc"], b["trunc"]) and all(torch.equal(x, y) for x, y in zip(a["straj"], b["straj"]))
