"""One environment per lane at large batches (csrc/kernels.hpp row_sync, csrc/launch.hpp): per-environment property arrays (the
reference's test_custom_initialization shape, e.g. tests/envs/pendulum/test_pendulum.py:72-129), control_state columns with gym
outputs, or a caller who asks for one environment per lane — from 2^17 environments on the four waves of a workgroup store every row
together, and with whole workgroups and aligned arrays the row leaves through LDS as 16-byte stores. The arithmetic is untouched:
results must match the oracle, and the lean V = 1 form must have the bits of the four-environments-per-lane kernel. ``-m gpu``."""
import numpy as np
import pytest
import torch

import oracle
from exciting_environments_amd import _native
from helpers import ANGLE_STATES, NP_DTYPE, circ_close, make_env, max_err, random_state, spec_of, to_state

pytestmark = pytest.mark.gpu


def _check_slices(env, env_name, spec, st, acts_np, props_fn, obs, states, dtype, semantics="ahead"):
    B = obs.shape[0]
    n = 700
    sem = oracle.SEM_STEP if semantics == "step" else oracle.SEM_AHEAD
    for sl in (slice(0, n), slice(B // 2 - 5, B // 2 - 5 + n), slice(B - n, B)):
        sub_props, _keepalive = props_fn(sl, n)  # the Props struct points into the keep-alive arrays
        o_ref, s_ref, _l = oracle.sim_ahead(env_name, "euler", [s[sl] for s in st], acts_np[sl], sub_props, spec["tau"], semantics=sem)
        rtol, atol = (1e-5, 2e-5) if dtype is torch.float32 else (1e-10, 1e-10)
        got = obs[sl].cpu().numpy()[..., :o_ref.shape[-1]]
        assert circ_close(got, o_ref, {"pendulum": [0]}.get(env_name, []), rtol, atol), max_err(got, o_ref)
        for j, name_j in enumerate(env.STATE_FIELDS):
            g = getattr(states.physical_state, name_j)[sl].cpu().numpy()
            scale = max(1.0, float(np.nanmax(np.abs(s_ref[j]))))
            if j in ANGLE_STATES.get(env_name, []):
                assert circ_close(g[..., None], s_ref[j][..., None], [0], rtol, atol * scale, period=2 * np.pi), name_j
            else:
                assert np.allclose(g, s_ref[j], rtol=rtol, atol=atol * scale), (name_j, max_err(g, s_ref[j]))


@pytest.mark.parametrize("env_name,dtype", [("pmsm", torch.float32), ("pmsm", torch.float64), ("pendulum", torch.float32), ("cartpole", torch.float64)])
@pytest.mark.parametrize("extra", [0, 100])  # whole workgroups (rows through LDS) / a ragged last workgroup (barrier only)
def test_per_environment_property_arrays_at_a_large_batch_match_the_oracle(env_name, dtype, extra):
    B, K = (1 << 17) + extra, 12
    spec = spec_of(env_name)
    rng = np.random.default_rng(81)
    if env_name == "pendulum":
        spec["params"]["l"] = rng.uniform(0.5, 2.5, B)
        spec["act_norm"]["torque"] = (-20, rng.uniform(15, 25, B))
    elif env_name == "pmsm":
        spec["params"]["r_s"] = rng.uniform(10e-3, 20e-3, B)
        spec["phys_norm"]["i_q"] = (-250, rng.uniform(200, 300, B))
    else:
        spec["params"]["m_p"] = rng.uniform(0.05, 0.2, B)
    env, props, keep, _ = make_env(env_name, B, dtype, spec=spec)
    env.trajectory_pool = False
    npdt = NP_DTYPE[dtype]
    st = random_state(env_name, B, npdt, spec, seed=82)
    acts_np = rng.uniform(-1, 1, (B, K, env.action_dim)).astype(npdt)
    acts = env.new_actions_buffer(K)
    acts.copy_(torch.as_tensor(acts_np, device=env.device))
    obs, states, last = env.vmap_sim_ahead(to_state(env, st), acts, env.tau, env.tau)
    torch.cuda.synchronize()
    assert _native.last_launch() == "sim_ahead_kernel (general)"

    def props_fn(sl, n):
        sub = {k: ({kk: (vv[sl] if isinstance(vv, np.ndarray) else vv) for kk, vv in v.items()} if k == "params" else
                   {kk: tuple((x[sl] if isinstance(x, np.ndarray) else x) for x in vv) for kk, vv in v.items()})
               for k, v in spec.items() if k in ("params", "phys_norm", "act_norm")}
        return oracle.make_props(env_name, sub["params"], sub["phys_norm"], sub["act_norm"], npdt, n)

    _check_slices(env, env_name, spec, st, acts_np, props_fn, obs, states, dtype)


@pytest.mark.parametrize("env_name,dtype,solver", [("pmsm", torch.float32, "euler"), ("pendulum", torch.float64, "tsit5"),
                                                    ("acrobot", torch.float32, "euler"), ("fluid_tank", torch.float32, "rk4")])
@pytest.mark.parametrize("semantics", ["ahead", "step"])
@pytest.mark.parametrize("with_states", [True, False])
def test_one_environment_per_lane_has_the_bits_of_four(env_name, dtype, solver, semantics, with_states):
    B, K = 1 << 17, 10
    env, props, keep, spec = make_env(env_name, B, dtype, solver=solver)
    env.sim_ahead_semantics = semantics
    env.trajectory_pool = False
    env.store_state_trajectory = with_states
    st = random_state(env_name, B, NP_DTYPE[dtype], spec, seed=83)
    acts = env.new_actions_buffer(K)
    acts.copy_(torch.as_tensor(np.random.default_rng(84).uniform(-1, 1, (B, K, env.action_dim)).astype(NP_DTYPE[dtype]), device=env.device))
    # the widest form asked for explicitly: the default rules may pick one environment per lane themselves (PMSM Euler observations
    # only, since round 5)
    env.launch_opts = _native.launch_opts(envs_per_lane=4 if dtype is torch.float32 else 2)
    wide = env.vmap_sim_ahead(to_state(env, st), acts, env.tau, env.tau)
    torch.cuda.synchronize()
    assert "V=1" not in _native.last_launch()
    env.launch_opts = _native.launch_opts(envs_per_lane=1)
    one = env.vmap_sim_ahead(to_state(env, st), acts, env.tau, env.tau)
    torch.cuda.synchronize()
    assert _native.last_launch() == "sim_ahead_kernel (V=1)"
    assert torch.equal(wide[0], one[0])
    for n in env.STATE_FIELDS:
        assert torch.equal(getattr(wide[2].physical_state, n), getattr(one[2].physical_state, n)), n
        if with_states:
            assert torch.equal(getattr(wide[1].physical_state, n), getattr(one[1].physical_state, n)), n


def test_control_columns_and_gym_outputs_through_the_general_kernel_at_a_large_batch():
    """control_state + per-environment properties + fused gym trajectories: O + 2 observation streams and the state leaves through
    LDS, reward / flags by direct stores — against the same call at a batch below the row_sync threshold (same environments)."""
    B, Bs, K = 1 << 17, 4096, 8
    spec = spec_of("cartpole")
    rng = np.random.default_rng(85)
    spec["params"]["m_p"] = rng.uniform(0.05, 0.2, B)
    env, props, keep, _ = make_env("cartpole", B, torch.float32, spec=spec, control_state=["theta", "deflection"])
    env.trajectory_pool = False
    st = random_state("cartpole", B, np.float32, spec, seed=86)
    refs = {"theta": rng.uniform(-3, 3, B).astype(np.float32), "deflection": rng.uniform(-2, 2, B).astype(np.float32)}
    acts_np = rng.uniform(-1, 1, (B, K, 1)).astype(np.float32)
    acts = env.new_actions_buffer(K)
    acts.copy_(torch.as_tensor(acts_np, device=env.device))
    big = env.vmap_sim_ahead(to_state(env, st, reference=refs), acts, env.tau, env.tau)
    torch.cuda.synchronize()
    assert _native.last_launch() == "sim_ahead_kernel (general)"
    spec_s = spec_of("cartpole")
    spec_s["params"]["m_p"] = spec["params"]["m_p"][:Bs]
    env_s, *_ = make_env("cartpole", Bs, torch.float32, spec=spec_s, control_state=["theta", "deflection"])
    env_s.trajectory_pool = False
    acts_s = env_s.new_actions_buffer(K)
    acts_s.copy_(torch.as_tensor(acts_np[:Bs], device=env.device))
    small = env_s.vmap_sim_ahead(to_state(env_s, [s[:Bs] for s in st], reference={k: v[:Bs] for k, v in refs.items()}), acts_s, env_s.tau, env_s.tau)
    assert torch.equal(big[0][:Bs], small[0])
    for n in env.STATE_FIELDS:
        assert torch.equal(getattr(big[1].physical_state, n)[:Bs], getattr(small[1].physical_state, n)), n
