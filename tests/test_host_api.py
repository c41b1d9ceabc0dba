"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol include/excenv.h declares,
and the Python mirror keeps the reference's constructor / attribute / error behaviour
(reference tests: tests/envs/test_core_functions.py, tests/envs/*/test_*.py default/custom initialisation).
No kernel is launched here."""
import ctypes
import math
import os
import re

import numpy as np
import pytest
import torch

import exciting_environments_amd as excenvs
from exciting_environments_amd import EnvironmentRegistry, MinMaxNormalization, _native
from exciting_environments_amd.tree import tree_structure

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
envs_to_test = list(EnvironmentRegistry)


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "excenv.h")).read()
    declared = set(re.findall(r"\b(excenv_[a-z_0-9]+)\s*\(", hdr))
    assert {"excenv_step", "excenv_sim_ahead", "excenv_last_error", "excenv_env_dims", "excenv_abi_version"} <= declared
    lib = ctypes.CDLL(_native.library_path())
    for sym in sorted(declared):
        assert hasattr(lib, sym), f"libexcenv_hip.so does not export {sym}"
    assert _native.lib().excenv_abi_version() == 7


def test_env_dims_and_algorithmic_bytes():
    dims = {0: (2, 1, 2, 3), 1: (2, 1, 2, 3), 2: (4, 1, 4, 6), 3: (4, 1, 4, 9), 4: (1, 1, 1, 4), 5: (7, 2, 8, 7)}
    for e, d in dims.items():
        assert _native.env_dims(e) == d
    # PMSM fp32 96 B (step) / 68 B (sim_ahead, full outputs) / 40 B (obs only)
    assert _native.step_bytes(5, torch.float32) == 96
    assert _native.sim_ahead_bytes(5, torch.float32, True) == 68
    assert _native.sim_ahead_bytes(5, torch.float32, False) == 40
    assert _native.sim_ahead_bytes(0, torch.float32, True) == 20
    assert _native.sim_ahead_bytes(1, torch.float64, True) == 40
    with pytest.raises(RuntimeError, match="bad env id"):
        _native.env_dims(17)


def test_abi_argument_validation_without_gpu():
    """Bad enums / NULL pointers are rejected by the C ABI before any HIP call."""
    lib = _native.lib()
    p = _native.Props()
    rc = lib.excenv_step(99, 0, 0, ctypes.c_int64(4), ctypes.byref(p), None, ctypes.c_double(1e-4), None, None, None, None, None, None)
    assert rc == -1 and b"bad env id" in lib.excenv_last_error()
    rc = lib.excenv_step(0, 7, 0, ctypes.c_int64(4), ctypes.byref(p), None, ctypes.c_double(1e-4), None, None, None, None, None, None)
    assert rc == -1 and b"bad solver id" in lib.excenv_last_error()
    rc = lib.excenv_step(0, 0, 0, ctypes.c_int64(4), ctypes.byref(p), None, ctypes.c_double(1e-4), None, None, None, None, None, None)
    assert rc == -2
    rc = lib.excenv_sim_ahead(0, 0, 0, ctypes.c_int64(4), ctypes.c_int64(-1), 1, ctypes.byref(p), None, ctypes.c_double(1e-4),
                              ctypes.c_double(1e-4), None, None, 0, None, None, 0, None, 0, None, None, None)
    assert rc == -1
    # per-call launch options are validated too (and there is no process-wide tuning entry point any more)
    bad = _native.LaunchOpts(3, 0, 0, 0)
    one = (ctypes.c_void_p * 8)(*([1] * 8))
    rc = lib.excenv_step(0, 0, 0, ctypes.c_int64(4), ctypes.byref(p), None, ctypes.c_double(1e-4), one, ctypes.c_void_p(16),
                         one, ctypes.c_void_p(16), ctypes.byref(bad), None)
    assert rc == -1 and b"envs_per_lane" in lib.excenv_last_error()
    assert not hasattr(lib, "excenv_set_tuning")
    # the entry points added with ABI v4
    i64, i32, vp = ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p
    assert lib.excenv_observe(99, 0, i64(4), ctypes.byref(p), None, one, vp(16), None) == -1
    assert lib.excenv_observe(0, 0, i64(4), ctypes.byref(p), None, None, vp(16), None) == -2 and b"NULL" in lib.excenv_last_error()
    ctl = _native.Control()
    ctl.n_control = 1
    ctl.control_idx[0] = 7  # the pendulum has two fields
    assert lib.excenv_observe(0, 0, i64(4), ctypes.byref(p), ctypes.byref(ctl), one, vp(16), None) == -1
    assert b"control_idx" in lib.excenv_last_error()
    assert lib.excenv_random_state(0, 0, i64(4), ctypes.byref(p), None, one, vp(16), None) == -2
    assert lib.excenv_random_state(0, 5, i64(4), ctypes.byref(p), vp(16), one, vp(16), None) == -1  # bad dtype
    # ABI v6: the collective wrapper validates before it touches RCCL; the launch-form query answers 0 for anything not fused
    assert lib.excenv_allgather(None, 0, vp(16), vp(16), i64(4), None) == -2 and b"NULL" in lib.excenv_last_error()
    assert lib.excenv_allgather(vp(16), 3, vp(16), vp(16), i64(4), None) == -1 and b"dtype" in lib.excenv_last_error()
    assert lib.excenv_allgather(vp(16), 0, vp(16), vp(16), i64(-1), None) == -1
    assert lib.excenv_allgather(vp(16), 0, None, None, i64(0), None) == 0  # nothing to gather
    assert lib.excenv_sim_ahead_fuses_actions(0, 0, 0, i64(1 << 20), i64(100), ctypes.byref(p), 0, 0, 0, 1, vp(16), None) == 1
    assert lib.excenv_sim_ahead_fuses_actions(0, 0, 0, i64(1 << 20), i64(101), ctypes.byref(p), 0, 0, 0, 1, vp(16), None) == 0  # 404-byte rows
    assert lib.excenv_sim_ahead_fuses_actions(0, 0, 0, i64(1 << 20), i64(100), ctypes.byref(p), 1, 0, 0, 1, vp(16), None) == 1  # control columns: filled behind the lean kernel
    assert lib.excenv_sim_ahead_fuses_actions(0, 0, 0, i64(1 << 20), i64(100), ctypes.byref(p), 0, 1, 0, 1, vp(16), None) == 0  # gym trajectories
    assert lib.excenv_sim_ahead_fuses_actions(0, 0, 0, i64(1 << 12), i64(100), ctypes.byref(p), 0, 0, 0, 1, vp(16), None) == 0  # small batch
    assert lib.excenv_sim_ahead_fuses_actions(0, 0, 0, i64(1 << 20), i64(100), ctypes.byref(p), 0, 0, 0, 0, vp(16), None) == 0  # row-major outputs
    off = _native.launch_opts(flags=_native.OPT_NO_FUSED_ACTIONS)
    assert lib.excenv_sim_ahead_fuses_actions(0, 0, 0, i64(1 << 20), i64(100), ctypes.byref(p), 0, 0, 0, 1, vp(16), ctypes.byref(off)) == 0
    assert lib.excenv_last_launch() in (b"", lib.excenv_last_launch())  # a string, never NULL
    assert lib.excenv_update_ref_to(0, 0, i64(4), ctypes.byref(p), i32(0), None, None, vp(16), vp(32), None, vp(16), vp(32),
                                    i32(1), i32(5), None) == -1 and b"alias" in lib.excenv_last_error()
    assert lib.excenv_update_ref_to(0, 0, i64(4), ctypes.byref(p), i32(9), None, None, vp(16), vp(32), None, vp(48), vp(64),
                                    i32(1), i32(5), None) == -1 and b"n_control" in lib.excenv_last_error()
    assert lib.excenv_update_ref_to(0, 0, i64(4), ctypes.byref(p), i32(0), None, None, None, vp(32), None, vp(48), vp(64),
                                    i32(1), i32(5), None) == -2


@pytest.mark.parametrize("env_type", envs_to_test)
@pytest.mark.parametrize("tau", [1e-4, 1e-5])
def test_tau(env_type, tau):
    assert env_type.make(tau=tau).tau == tau


def test_default_parameters_match_reference():
    """Literal defaults of the reference constructors (e.g. pendulum_env.py:84-97, motor_parameters.py:124-149)."""
    pend = EnvironmentRegistry.PENDULUM.make()
    assert (pend.batch_size, pend.tau) == (8, 1e-4)
    sp = pend.env_properties.static_params
    assert (sp.g, sp.l, sp.m) == (9.81, 2, 1)
    pn = pend.env_properties.physical_normalizations
    assert (pn.theta.min, pn.theta.max, pn.omega.min, pn.omega.max) == (-math.pi, math.pi, -10, 10)
    assert (pend.env_properties.action_normalizations.torque.min, pend.env_properties.action_normalizations.torque.max) == (-20, 20)
    msd = EnvironmentRegistry.MASS_SPRING_DAMPER.make()
    assert (msd.env_properties.static_params.k, msd.env_properties.static_params.d, msd.env_properties.static_params.m) == (100, 1, 1)
    cp = EnvironmentRegistry.CART_POLE.make()
    assert cp.tau == 2e-2 and cp.env_properties.static_params.mu_p == 0.000002 and cp.env_properties.physical_normalizations.deflection.max == 2.4
    ac = EnvironmentRegistry.ACROBOT.make()
    assert ac.tau == 1e-3 and ac.env_properties.static_params.I_1 == 1.3 and len(ac.PARAM_FIELDS) == 9
    ft = EnvironmentRegistry.FLUID_TANK.make()
    assert (ft.batch_size, ft.tau) == (1, 1e-3) and ft.env_properties.static_params.orifice_area == math.pi * 0.1**2
    pm = EnvironmentRegistry.PMSM.make()
    sp = pm.env_properties.static_params
    assert (sp.p, sp.r_s, sp.l_d, sp.l_q, sp.psi_p, sp.u_dc, sp.deadtime) == (3, 15e-3, 0.37e-3, 1.2e-3, 65.6e-3, 400, 1)
    assert pm.env_properties.physical_normalizations.omega_el.max == 3 * 11000 * 2 * math.pi / 60
    assert pm.env_properties.physical_normalizations.i_d.min == -250 and pm.env_properties.saturated is False
    for variant in excenvs.MotorVariant:
        env = EnvironmentRegistry.PMSM.make(motor_variant=variant)
        want = variant.get_params().static_params
        assert all(getattr(env.env_properties.static_params, k) == v for k, v in want.items())
    with pytest.raises(ValueError, match="not allowed for saturated"):
        EnvironmentRegistry.PMSM.make(saturated=True)


def test_custom_initialization_keeps_arrays_and_marks_axes():
    """reference tests/envs/pendulum/test_pendulum.py:72-129: per-env arrays are stored as given; in_axes 0 / None."""
    B = 4
    phys = {"theta": MinMaxNormalization(min=np.repeat(-np.pi / 2, B), max=np.pi / 2), "omega": MinMaxNormalization(min=-5, max=3)}
    act = {"torque": MinMaxNormalization(min=-10, max=10)}
    params = {"l": torch.ones(B), "g": 9.81, "m": 1}
    env = EnvironmentRegistry.PENDULUM.make(batch_size=B, static_params=params, physical_normalizations=phys, action_normalizations=act)
    assert env.env_properties.static_params.l is params["l"]
    assert env.env_properties.physical_normalizations.theta.min is phys["theta"].min
    ax = env.in_axes_env_properties
    assert ax.static_params.l == 0 and ax.static_params.g is None
    assert ax.physical_normalizations.theta.min == 0 and ax.physical_normalizations.theta.max is None
    assert ax.action_normalizations.torque.min is None
    # an array whose leading dimension is not batch_size is a broadcast leaf (core_env.py:268-272)
    env2 = EnvironmentRegistry.PENDULUM.make(batch_size=B, static_params={"l": np.ones(3), "g": 9.81, "m": 1})
    assert env2.in_axes_env_properties.static_params.l is None


def test_property_type_errors():
    with pytest.raises(ValueError, match="but list is given"):
        EnvironmentRegistry.PENDULUM.make(static_params={"l": [1.0, 2.0], "g": 9.81, "m": 1})
    with pytest.raises(ValueError, match="needs to be a scalar"):
        EnvironmentRegistry.PENDULUM.make(static_params={"l": "long", "g": 9.81, "m": 1})
    with pytest.raises(TypeError):
        EnvironmentRegistry.PENDULUM.make(solver="euler")
    with pytest.raises(TypeError):
        EnvironmentRegistry.PENDULUM.make(dtype=torch.float16)


@pytest.mark.parametrize("env_type", envs_to_test)
def test_reset(env_type):
    """reference tests/envs/test_core_functions.py:25-52."""
    B = 4
    env = env_type.make(batch_size=B, device="cpu")
    obs, state = env.reset(env.env_properties, 1234)
    assert obs.shape == env.obs_description.shape and type(state) == env.State
    obs, state = env.reset(env.env_properties)
    assert obs.shape == env.obs_description.shape and type(state) == env.State
    obs, state = env.vmap_reset(1234)
    assert obs.shape == (B, len(env.obs_description)) and type(state) == env.State
    assert float(obs.abs().max()) <= 1.0 + 1e-6
    obs, state = env.vmap_reset()
    assert obs.shape == (B, len(env.obs_description)) and type(state) == env.State
    assert not bool(state.additions.active_solver_state.any())
    assert bool(torch.isnan(state.PRNGKey).all()) and bool(torch.isnan(getattr(state.reference, env.STATE_FIELDS[0])).all())
    obs2, state2 = env.vmap_reset(initial_state=state)
    assert torch.equal(obs, obs2)
    with pytest.raises(AssertionError, match="same dataclass structure"):
        env.vmap_reset(initial_state=state.physical_state)


@pytest.mark.parametrize("env_type", envs_to_test)
def test_solver_state_leaf_has_the_reference_structure(env_type):
    """Additions.solver_state (e.g. pendulum_env.py:177-192, 289-290): None for Euler (diffrax.Euler keeps no solver state) and
    the RK4 extension, a NaN-filled (first_step, f0) pair for Tsit5 with f0 shaped like the ODE state the reference integrates
    (PMSM: (i_d, i_q, eps), pmsm_env.py:555) — so that the pytree structure of a State does not depend on where it came from
    (reset, single-env reset, initial_state round trip)."""
    import exciting_environments_amd as ex
    from exciting_environments_amd.tree import tree_flatten, tree_structure

    B = 3
    n_ode = {"PMSM": 3}.get(env_type.name, None)
    for solver, fsal in ((ex.Euler(), False), (ex.RK4(), False), (ex.Tsit5(), True)):
        env = env_type.make(batch_size=B, device="cpu", solver=solver)
        _, state = env.vmap_reset()
        ss = state.additions.solver_state
        if not fsal:
            assert ss is None
            continue
        n = n_ode or len(env.STATE_FIELDS)
        assert isinstance(ss, tuple) and len(ss) == 2 and isinstance(ss[1], tuple) and len(ss[1]) == n
        leaves, _ = tree_flatten(ss)
        assert len(leaves) == n + 1 and all(tuple(l.shape) == (B,) and bool(torch.isnan(l).all()) for l in leaves)
        _, single = env.reset(env.env_properties)
        assert tree_structure(single) == tree_structure(state)
        assert all(tuple(l.shape) == () for l in tree_flatten(single.additions.solver_state)[0])
        _, again = env.vmap_reset(initial_state=state)
        assert tree_structure(again) == tree_structure(state)


def test_default_reset_states_in_physical_units():
    """SURVEY.md §8 a11."""
    val = lambda env, n: float(getattr(env.vmap_reset()[1].physical_state, n)[0])
    pend = EnvironmentRegistry.PENDULUM.make(device="cpu", dtype=torch.float64)
    assert val(pend, "theta") == math.pi and val(pend, "omega") == 0.0
    cp = EnvironmentRegistry.CART_POLE.make(device="cpu", dtype=torch.float64)
    assert val(cp, "theta") == math.pi and val(cp, "deflection") == 0.0
    ft = EnvironmentRegistry.FLUID_TANK.make(device="cpu", dtype=torch.float64)
    assert val(ft, "height") == 1.5
    pm = EnvironmentRegistry.PMSM.make(device="cpu", dtype=torch.float64)
    assert val(pm, "i_d") == -125.0 and abs(val(pm, "omega_el") - 3 * 11000 * 2 * math.pi / 60 / 2) < 1e-9
    obs, _ = pm.vmap_reset()
    assert obs[0].tolist() == [0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0]


@pytest.mark.parametrize("env_type", envs_to_test)
def test_gen_observation_gen_state(env_type):
    """reference tests/envs/test_core_functions.py:55-77: obs -> state -> obs round trip is exact."""
    env = env_type.make(batch_size=4, device="cpu", dtype=torch.float64)
    obs, state = env.reset(env.env_properties)
    assert torch.equal(obs, env.generate_observation(state, env.env_properties))
    assert torch.equal(obs, env.generate_observation(env.generate_state_from_observation(obs, env.env_properties), env.env_properties))
    obs, state = env.vmap_reset()
    assert torch.equal(obs, env.generate_observation(env.vmap_generate_state_from_observation(obs), env.env_properties))


def test_control_state_extends_observation():
    env = EnvironmentRegistry.PENDULUM.make(batch_size=3, control_state=["theta"], device="cpu")
    assert list(env.obs_description) == ["theta", "omega", "theta_ref"]
    obs, state = env.vmap_reset()
    assert obs.shape == (3, 3) and bool(torch.isnan(obs[:, 2]).all())
    pm = EnvironmentRegistry.PMSM.make(batch_size=3, control_state=["i_d", "i_q"], device="cpu")
    assert pm.vmap_reset()[0].shape == (3, 10) and list(pm.obs_description[-2:]) == ["i_d_ref", "i_q_ref"]


@pytest.mark.parametrize("env_type", envs_to_test)
def test_shape_assertions_fire_before_any_kernel(env_type):
    """Messages of core_env.py:546-563, 591-609."""
    B = 4
    env = env_type.make(batch_size=B, device="cpu")
    _, state = env.vmap_reset()
    with pytest.raises(AssertionError, match=r"The action needs to be of shape \(batch_size, action_dim\)"):
        env.vmap_step(state, torch.ones(B + 1, env.action_dim))
    with pytest.raises(AssertionError, match="three dimensions"):
        env.vmap_sim_ahead(state, torch.ones(B, env.action_dim), env.tau, env.tau)
    with pytest.raises(AssertionError, match="does not correspond to the batch size"):
        env.vmap_sim_ahead(state, torch.ones(B + 1, 5, env.action_dim), env.tau, env.tau)
    with pytest.raises(AssertionError, match="does not correspond to the action dim"):
        env.vmap_sim_ahead(state, torch.ones(B, 5, env.action_dim + 1), env.tau, env.tau)
    with pytest.raises(AssertionError, match="greater or equal to the observation stepsize"):
        env.vmap_sim_ahead(state, torch.ones(B, 5, env.action_dim), 2 * env.tau, env.tau)
    bad = env.State(physical_state=env.PhysicalState(**{n: torch.zeros(B + 2) for n in env.STATE_FIELDS}),
                    PRNGKey=state.PRNGKey, additions=state.additions, reference=state.reference)
    with pytest.raises(AssertionError, match="physical state needs to be of shape"):
        env.vmap_step(bad, torch.ones(B, env.action_dim))
    _, s1 = env.reset(env.env_properties)
    with pytest.raises(AssertionError, match=r"shape \(action_dim,\)"):
        env.step(s1, torch.ones(env.action_dim + 1), env.env_properties)


def test_no_cpu_fallback():
    """The product path fails loudly without a HIP device instead of computing on the CPU."""
    env = EnvironmentRegistry.PENDULUM.make(batch_size=2, device="cpu")
    _, state = env.vmap_reset()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        env.vmap_step(state, torch.ones(2, 1))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        env.vmap_sim_ahead(state, torch.ones(2, 3, 1), env.tau, env.tau)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "exciting-environments_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hpp", ".hip", ".h")):
                txt = open(os.path.join(d, f)).read()
                assert "import oracle" not in txt and "liboracle" not in txt and "oracle/" not in txt.replace("oracle/oracle_body.inc rk_step", ""), f


def test_tree_structure_helpers():
    env = EnvironmentRegistry.ACROBOT.make(batch_size=2, device="cpu")
    _, a = env.vmap_reset()
    _, b = env.vmap_reset(3)
    assert tree_structure(a) == tree_structure(b)
    assert tree_structure(a) != tree_structure(a.physical_state)


def test_sim_properties_json_round_trip(tmp_path):
    params = {"g": 9.81, "l": 2, "m": 1}
    an = {"torque": MinMaxNormalization(-20, 20)}
    pn = {"theta": MinMaxNormalization(-math.pi, math.pi), "omega": MinMaxNormalization(-10, 10)}
    f = tmp_path / "p.json"
    excenvs.dump_sim_properties_to_json(params, an, pn, 1e-4, f)
    p2, a2, n2, tau = excenvs.load_sim_properties_from_json(f)
    assert p2 == params and a2 == an and n2 == pn and tau == 1e-4
    g = excenvs.load_sim_properties_from_json(os.path.join(ROOT, "tests", "golden", "pmsm", "sim_properties.json"))
    assert g[0]["deadtime"] == 1 and g[3] == 1e-4


def test_gym_wrapper_host_logic():
    """reference gym_wrapper.py:16-59 constructor checks; reset with / without a reference generator."""
    from exciting_environments_amd import GymWrapper

    env = EnvironmentRegistry.CART_POLE.make(batch_size=3, device="cpu")
    with pytest.raises(AssertionError, match="no valid physical state"):
        GymWrapper(env, control_state=["nope"])
    with pytest.raises(AssertionError, match="has to be a list"):
        GymWrapper(env, control_state="theta")
    g = GymWrapper(env, control_state=["theta", "omega"])
    assert env.control_state == ["theta", "omega"] and len(env.obs_description) == 6
    obs, info = g.reset()
    assert g.ref_gen is False and info == {} and obs.shape == (3, 6)
    obs, _ = g.reset(rng_env=3, rng_ref=4)
    assert g.ref_gen is True and g.reference_hold_steps.shape == (3, 1)
    assert int(g.reference_hold_steps.min()) >= 10 and int(g.reference_hold_steps.max()) < 1000
    assert bool(torch.isfinite(g.state.reference.theta).all()) and bool(torch.isnan(g.state.reference.velocity).all())
    g2 = GymWrapper.from_env(EnvironmentRegistry.PENDULUM, batch_size=2, device="cpu")
    assert g2.env.batch_size == 2


def test_reward_truncated_terminated_torch_mirrors_cpu():
    """Reference semantics on CPU tensors: angles through sin/cos, others squared normalised error; PMSM flags."""
    env = EnvironmentRegistry.PENDULUM.make(batch_size=2, control_state=["theta", "omega"], device="cpu", dtype=torch.float64)
    _, st = env.vmap_reset()
    st.physical_state.theta = torch.tensor([0.1, 3.0], dtype=torch.float64)
    st.physical_state.omega = torch.tensor([0.0, 11.0], dtype=torch.float64)
    st.reference.theta = torch.tensor([0.2, 0.2], dtype=torch.float64)
    st.reference.omega = torch.tensor([0.0, 0.0], dtype=torch.float64)
    r = env.generate_reward(st, None, env.env_properties)
    want0 = -((math.sin(0.1) - math.sin(0.2)) ** 2 + (math.cos(0.1) - math.cos(0.2)) ** 2)
    want1 = -((math.sin(3.0) - math.sin(0.2)) ** 2 + (math.cos(3.0) - math.cos(0.2)) ** 2) - (1.1 - 0.0) ** 2
    assert r.shape == (2, 1) and abs(float(r[0, 0]) - want0) < 1e-15 and abs(float(r[1, 0]) - want1) < 1e-12
    tr = env.generate_truncated(st, env.env_properties)
    assert tr.tolist() == [[False, False, False, False], [False, True, False, False]]
    assert env.generate_terminated(st, r, env.env_properties).tolist() == [[False], [False]]
    pm = EnvironmentRegistry.PMSM.make(batch_size=2, device="cpu", dtype=torch.float64)
    _, ps = pm.vmap_reset()
    ps.physical_state.i_q = torch.tensor([0.0, 249.0], dtype=torch.float64)
    ps.physical_state.i_d = torch.tensor([-125.0, -10.0], dtype=torch.float64)
    assert pm.generate_truncated(ps, pm.env_properties).tolist() == [[False], [True]]
    assert pm.generate_reward(ps, None, pm.env_properties).tolist() == [[0.0], [0.0]]


def test_output_slot_liveness_test_on_cpu_tensors():
    """core_env.py _slot_is_free (the guard of the recycled vmap_step outputs) without a GPU: a fresh pool's slot is free; a
    reference to any tensor of it, to its PhysicalState, a view, a detached alias or a DLPack capsule makes it busy; so does
    another stream handle; dropping the holder frees it again."""
    import gc

    import torch
    from exciting_environments_amd import EnvironmentRegistry

    from exciting_environments_amd.core_env import CoreEnvironment

    env = EnvironmentRegistry.CART_POLE.make(batch_size=64, device="cpu")
    from exciting_environments_amd import _placement

    if not _placement.liveness_available():
        pytest.skip("this torch build exposes no use counts: slots are never recycled")
    sl = env._step_pool.new_slots(4, False)
    assert not env._step_pool.is_free(sl, 0, 0)  # not armed yet
    env._step_pool.arm(sl, 0)
    assert all(env._step_pool.is_free(sl, i, 0) for i in range(4))
    assert not env._step_pool.is_free(sl, 1, 12345)  # another stream
    holders = {
        "leaf": lambda: sl.leaves[2][1],
        "obs": lambda: sl.obs[2],
        "physical_state": lambda: sl.phys[2],
        "view": lambda: sl.leaves[2][0][3:9],
        "detach": lambda: sl.obs[2].detach(),
        "dlpack": lambda: torch.utils.dlpack.to_dlpack(sl.leaves[2][3]),
        "numpy": lambda: sl.obs[2].numpy(),
    }
    for name, make in holders.items():
        h = make()
        assert not env._step_pool.is_free(sl, 2, 0), name
        if name in ("leaf", "obs", "physical_state"):  # holders of OUR objects leave the pool's other slots reusable
            assert env._step_pool.is_free(sl, 1, 0), name
        del h
        gc.collect()
        assert env._step_pool.is_free(sl, 2, 0), name
    sg = env._step_pool.new_slots(3, True)
    env._step_pool.arm(sg, 0)
    rew = sg.gym[1][0]
    assert not env._step_pool.is_free(sg, 1, 0) and env._step_pool.is_free(sg, 0, 0)
    del rew
    assert env._step_pool.is_free(sg, 1, 0)
    assert env._step_pool.per_alloc(False) >= 3


def test_single_environment_rew_trunc_term_ahead_and_repeat_values():
    """core_env.py:490-531 (single-environment form) agrees with row b of the vmapped form; repeat_values (core_env.py:279-290)."""
    import torch
    from dataclasses import replace
    from exciting_environments_amd import EnvironmentRegistry

    B, K = 3, 6
    env = EnvironmentRegistry.PENDULUM.make(batch_size=B, device="cpu", control_state=["theta"])
    g = torch.Generator().manual_seed(3)
    phys = env.PhysicalState(theta=(torch.rand((B, K + 1), generator=g) - 0.5) * 6, omega=(torch.rand((B, K + 1), generator=g) - 0.5) * 30)
    ref = env.PhysicalState(theta=(torch.rand((B, K + 1), generator=g) - 0.5) * 6, omega=torch.full((B, K + 1), float("nan")))
    _, st = env.vmap_reset()
    states = replace(st, physical_state=phys, reference=ref)
    acts = torch.zeros((B, K, 1))
    rew, trunc, term = env.vmap_generate_rew_trunc_term_ahead(states, acts)
    assert rew.shape == (B, K, 1) and trunc.shape[:2] == (B, K + 1) and term.shape == (B, K, 1)
    for b in range(B):
        one = replace(st, physical_state=env.PhysicalState(theta=phys.theta[b], omega=phys.omega[b]),
                      reference=env.PhysicalState(theta=ref.theta[b], omega=ref.omega[b]))
        r1, tr1, te1 = env.generate_rew_trunc_term_ahead(one, acts[b], env.env_properties)
        assert torch.equal(r1, rew[b]) and torch.equal(tr1, trunc[b]) and torch.equal(te1, term[b])
    with pytest.raises(AssertionError, match="two dimensions"):
        env.generate_rew_trunc_term_ahead(states, acts, env.env_properties)
    assert env.repeat_values(None, 4) is None
    assert env.repeat_values(2.5, 3).tolist() == [2.5, 2.5, 2.5] and env.repeat_values(True, 2).tolist() == [True, True]
    t = env.repeat_values((torch.tensor(1.0), torch.tensor([1.0, 2.0])), 2)
    assert t[0].tolist() == [1.0, 1.0] and t[1].tolist() == [[1.0, 2.0], [1.0, 2.0]]
    with pytest.raises(ValueError):
        env.repeat_values("x", 2)


def test_reference_import_paths():
    """from exciting_environments.pmsm import PMSM, MotorVariant etc. (the reference's sub-packages) work with the package name swapped."""
    from exciting_environments_amd.acrobot import Acrobot
    from exciting_environments_amd.cart_pole import CartPole
    from exciting_environments_amd.fluid_tank import FluidTank
    from exciting_environments_amd.mass_spring_damper import MassSpringDamper
    from exciting_environments_amd.pendulum import Pendulum
    from exciting_environments_amd.pmsm import PMSM, MotorVariant
    import exciting_environments_amd as ex

    assert (Acrobot, CartPole, FluidTank, MassSpringDamper, Pendulum, PMSM, MotorVariant) == (
        ex.Acrobot, ex.CartPole, ex.FluidTank, ex.MassSpringDamper, ex.Pendulum, ex.PMSM, ex.MotorVariant)


def test_headline_loop_stays_inside_the_instruction_cache():
    """Guard rail (tools/loop_code_size.py): the K loop of the headline kernel — PMSM Euler fp32, V = 4, two unrolled solver
    steps x 4 environments per lane — is the largest piece of code that must stay resident in the 64 KB instruction cache for
    the headline number; anything that grows it past 60 KB fails here, at build time, instead of showing up as a slow GPU run."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("loop_code_size", os.path.join(ROOT, "tools", "loop_code_size.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if not os.path.exists(mod.OBJDUMP):
        pytest.skip("llvm-objdump not available")
    spans = mod.loop_spans()
    assert mod.HEADLINE in spans, "headline instantiation missing from the library"
    loop, size = spans[mod.HEADLINE]
    assert 8 * 1024 < loop <= 60 * 1024, f"headline K loop is {loop} bytes"
    # the C2 / C4 kernels (pendulum Euler fp32, mass-spring-damper Tsit5 fp64) are far below it
    small = [v[0] for k, v in spans.items() if ("Pendulum" in k or "MassSpringDamper" in k) and "sim_ahead_kernel" in k]
    assert small and max(small) <= 60 * 1024


def test_trajectory_kernels_do_not_spill():
    """Guard rail (tools/loop_code_size.py kernel_resources): a trajectory kernel that spills reloads its registers behind
    `s_waitcnt vmcnt(0)`, i.e. behind every outstanding trajectory store (the register-ring kernel lost 0.6 of 7.7 ms to 18 spilled
    registers; a non-inlined lambda costs a stack frame the same way). The headline kernels use no scratch memory at all; the
    register-ring kernels at most a few bytes that only cold paths touch."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("loop_code_size", os.path.join(ROOT, "tools", "loop_code_size.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if not (os.path.exists(mod.OBJDUMP) and os.path.exists(mod.READELF)):
        pytest.skip("llvm-objdump / llvm-readelf not available")
    res = mod.kernel_resources()
    assert res[mod.HEADLINE]["scratch"] == 0 and res[mod.HEADLINE]["vgpr"] <= 256
    ring = {k: v for k, v in res.items() if "sim_ahead_emr_kernel" in k}
    assert len(ring) >= 60
    worst = max(ring.items(), key=lambda kv: kv[1]["scratch"])
    # fp64 PMSM (five ring leaves x 8 doubles next to a double-precision integration): up to eleven register pairs, none of them
    # reloaded inside the step loop's common path — the launch measures 5.5 ms / 0.65 of the roof (round 3, 44 bytes: 5.7 ms)
    assert worst[1]["scratch"] <= 96, worst
    assert all(v["scratch"] == 0 for k, v in ring.items() if "PmsmIfEE" in k)
    # the acrobot instantiations with gym outputs are capped at 256 registers on purpose (kernels.hpp sim_min_waves: two waves per
    # SIMD with a few spilled registers measured 4.66 ms where 263 registers and one wave measured 6.0): bounded, not zero
    capped = {k: v for k, v in res.items() if "sim_ahead_kernel" in k and "AcrobotIfEEf" in k and "ELb0ELb0ELb1ELi256EEE" in k}
    assert capped and all(v["vgpr"] <= 256 for v in capped.values())
    # ... and the Euler instantiations of cart-pole and acrobot with gym outputs at 168 (a third wave per SIMD, measured 3 % faster
    # with 48 ... 128 spilled bytes than two waves without): what default options run
    capped.update({k: v for k, v in res.items() if "sim_ahead_kernel" in k and "CartPoleIfEEfLi0E" in k and "ELb0ELb0ELb1ELi256EEE" in k})
    euler = {k: v for k, v in capped.items() if "AcrobotIfEEfLi0E" in k or "CartPoleIfEEfLi0E" in k}
    assert len(euler) == 8 and all(v["vgpr"] <= 168 and v["scratch"] <= 160 for v in euler.values()), euler
    lean32 = {k: v for k, v in res.items() if "sim_ahead_kernel" in k and "IfEEf" in k and "PmsmSat" not in k and k not in capped}
    assert lean32 and all(v["scratch"] == 0 for v in lean32.values()), [k for k, v in lean32.items() if v["scratch"]]


def test_placement_search_memory_budget_for_the_c5_shard():
    """SURVEY.md §8e / BASELINE C5: every rank of the 8-GPU run holds 2^22 PMSM environments, 100-step chunks, fp32. The pooled +
    placed output sets must fit one MI355X (288 GB) with room for the actions, the gathered observations and torch itself — also
    at the peak of a placement search (candidates + spacers)."""
    from exciting_environments_amd.core_env import CoreEnvironment

    B, rows, OW, S, isz = 1 << 22, 101, 8, 7, 4
    hbm = 288 * 10**9
    b = CoreEnvironment.placement_memory_budget(B, rows, OW, S, isz, free_bytes=hbm - 8 * 10**9)
    assert abs(b["set"] - 25.53e9) < 0.05e9
    assert b["steady"] <= 52e9                       # two sets
    assert b["search_peak"] <= 0.65 * hbm, b         # the live set, the set being replaced and a search in progress
    actions = B * 100 * 2 * isz
    gathered = 8 * B * OW * isz                      # the all-gathered final observation row of eight ranks
    assert b["steady"] + b["search_peak"] - b["set"] + actions + gathered < 0.75 * hbm
    assert b["searches_at_most"] == 4
    # a device that is already full leaves no room for spacers: the bound follows the free memory, not the constant
    tight = CoreEnvironment.placement_memory_budget(B, rows, OW, S, isz, free_bytes=30 * 10**9)
    assert tight["search_peak"] < b["search_peak"] and tight["search_peak"] <= 2 * tight["set"] + 4 * 12e9 + 14e9 + 4 * 10e9 + 1e9


def test_placement_never_judges_a_set_on_its_first_launch():
    """VERDICT r04 item 4 / BENCH_r04: the driver's run timed 10.388 ms for the first launch into a pooled set of a fresh process
    (clock ramp-up, first-touch mapping), then 4.88 / 4.89 — round 4 took min() from the first launch on, judged the set 2x slower
    than its sibling and ran a replacement search (seconds of allocator churn) before keeping the old set. The decision logic with
    scripted timings, no GPU: a first launch is recorded apart and never compared; sets the absolute criterion accepted are final;
    a set whose STEADY launches are slow is still replaced, and only early."""
    from types import SimpleNamespace

    import torch
    from exciting_environments_amd._placement import TrajectoryPlacement, TrajSet

    env = SimpleNamespace(trajectory_placement="auto", trajectory_pool=True, dtype=torch.float32, device=torch.device("cpu"))
    key = (1 << 22, 101, 8, 7, True, torch.float32)

    def pair(pattern):
        pl = TrajectoryPlacement(env)
        sets = [TrajSet(key), TrajSet(key)]
        for t in sets:
            t.placement = {"pattern_over_fill": 0.8394, "accept_at": 0.81} if pattern else {"chosen_ms": 4.9}
        pl.sets = sets
        return pl, sets

    # exactly the BENCH_r04 sequence, sets judged by real launches: 10.4 (cold), then 4.9s
    pl, (a, b) = pair(pattern=False)
    a.record_ms(10.388)
    assert a.first_ms == 10.388 and a.steady_ms is None and a.uses == 0
    assert not pl.replacement_due(a, [b]) and not pl.settled  # nothing to judge yet
    b.record_ms(4.95)
    a.record_ms(4.8837)
    assert not pl.replacement_due(a, [b]) and not pl.replacement_due(b, [a])
    b.record_ms(4.8874)
    assert a.steady_ms == 4.8837 and b.steady_ms == 4.8874 and not pl.replacement_due(a, [b]) and not pl.replacement_due(b, [a])
    assert not pl.settled  # still inside the decision window: a later launch could move the comparison
    for _ in range(3):
        a.record_ms(4.89)
        b.record_ms(4.88)
        assert not pl.replacement_due(a, [b]) and not pl.replacement_due(b, [a])
    assert pl.settled and pl.replaced == {}
    # sets accepted by the absolute criterion are final from the start: settled before any launch has been timed
    pl, (a, b) = pair(pattern=True)
    assert pl.settled
    a.record_ms(10.388)
    a.record_ms(9.0)
    b.record_ms(4.9)
    b.record_ms(4.9)
    assert not pl.replacement_due(a, [b]) and pl.settled
    # a set that is slow in its steady launches is up for replacement — while young, and at most REPLACEMENTS times per shape
    pl, (a, b) = pair(pattern=False)
    for ms in (10.4, 5.6, 5.61):
        a.record_ms(ms)
    for ms in (5.0, 4.9, 4.91):
        b.record_ms(ms)
    assert pl.replacement_due(a, [b]) and not pl.replacement_due(b, [a]) and not pl.settled
    for ms in (5.6, 5.6):
        a.record_ms(ms)
        b.record_ms(4.9)
    assert a.uses == 4 and b.uses == 4 and not pl.replacement_due(a, [b]) and pl.settled  # past the decision window: it stays
    pl.replaced[key] = pl.REPLACEMENTS
    a2 = TrajSet(key)
    a2.placement = {"chosen_ms": 5.7}
    a2.record_ms(9.0)
    a2.record_ms(5.7)
    assert not pl.replacement_due(a2, [b])
    # switched off: nothing is ever due, always settled
    env.trajectory_placement = "off"
    assert pl.settled and not pl.replacement_due(a, [b])
    env.trajectory_placement = "auto"
