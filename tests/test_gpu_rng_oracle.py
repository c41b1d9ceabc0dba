"""The device key-stream kernels against the ORACLE's restatement of jax.random (oracle/oracle_rng.inc, pinned on the Random123
known-answer vectors and on values published in JAX's documentation, tests/test_oracle_rng.py) — not against the product's
own torch twin: ``excenv_random_state`` (``vmap_init_state(keys)``, core_env.py:649-662; init_state pendulum_env.py:270-276,
pmsm_env.py:402-456) and ``excenv_update_ref_to`` / ``excenv_update_ref`` (``GymWrapper.update_ref``, gym_wrapper.py:170-192).
Keys, key leaves and hold counters are integer work: exact. States drawn through ``uniform`` pass the same three roundings on
both sides: exact. PMSM's currents come out of erf_inv / log / pow / sqrt (``jax.random.ball``): device math library against
the oracle's double-precision erf_inv — a few ulp, and a rejection decided differently by one ulp changes a sample (a fraction
well below one in a thousand). ``-m gpu``."""
import numpy as np
import pytest
import torch

import oracle
from exciting_environments_amd import _native
from helpers import NP_DTYPE, make_env

pytestmark = pytest.mark.gpu
ENV_NAMES = ["pendulum", "mass_spring_damper", "cartpole", "acrobot", "fluid_tank", "pmsm"]
CONTROL = {"pendulum": ["theta"], "mass_spring_damper": ["deflection", "velocity"], "cartpole": ["theta", "deflection"],
           "acrobot": ["theta_1", "omega_2"], "fluid_tank": ["height"], "pmsm": ["i_d", "i_q", "torque"]}


def _ball_tolerance(dtype):
    eps = np.finfo(NP_DTYPE[dtype]).eps
    return 256 * eps  # relative to the current scale (i_max = 250 A, torque ~ 200 Nm)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("env_name", ENV_NAMES)
def test_random_state_kernel_matches_the_oracle(env_name, dtype):
    B = 4099  # ragged
    env, props, keep, spec = make_env(env_name, B, dtype)
    keys = oracle.split(oracle.prng_key(77), B)
    want, want_leaf = oracle.random_state(env_name, keys, props, NP_DTYPE[dtype])
    state = env.vmap_init_state(torch.as_tensor(keys, device=env.device))
    assert state.PRNGKey.dtype == torch.int64 and np.array_equal(state.PRNGKey.cpu().numpy(), want_leaf)
    for j, n in enumerate(env.STATE_FIELDS):
        got = getattr(state.physical_state, n).cpu().numpy()
        if env_name == "pmsm" and n in ("i_d", "i_q", "torque"):
            scale = 250.0
            close = np.abs(got - want[j]) <= _ball_tolerance(dtype) * scale
            assert close.mean() > 0.999, (n, float(np.abs(got - want[j]).max()))
        else:
            assert np.array_equal(got, want[j]), (n, float(np.abs(got - want[j]).max()))


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("env_name", ENV_NAMES)
def test_update_ref_kernels_match_the_oracle(env_name, dtype):
    """Out-of-place (excenv_update_ref_to) and in-place (excenv_update_ref) launches over a batch of which about a third is due."""
    B = 8192
    env, props, keep, spec = make_env(env_name, B, dtype)
    cs = CONTROL[env_name]
    idx = [env.STATE_FIELDS.index(n) for n in cs]
    rng = np.random.default_rng(3)
    keys = oracle.split(oracle.prng_key(123), B)
    hold = rng.integers(0, 3, B)
    refs = [rng.uniform(-1, 1, B).astype(NP_DTYPE[dtype]) for _ in cs]
    want_refs, want_keys, want_hold = oracle.update_ref(env_name, idx, refs, keys, hold, props, NP_DTYPE[dtype], 10, 1000)
    p, _k = env._props_for(env.env_properties, B)
    dev = env.device
    k_in, h_in = torch.as_tensor(keys, device=dev), torch.as_tensor(hold, device=dev)
    r_in = [torch.as_tensor(r, device=dev) for r in refs]
    # out of place
    r_out = [torch.empty_like(r) for r in r_in]
    k_out, h_out = torch.empty_like(k_in), torch.empty_like(h_in)
    _native.update_ref_to(env.ENV_ID, dtype, B, p, idx, r_in, k_in, h_in, r_out, k_out, h_out, 10, 1000)
    # in place
    r_ip, k_ip, h_ip = [r.clone() for r in r_in], k_in.clone(), h_in.clone()
    _native.update_ref(env.ENV_ID, dtype, B, p, idx, r_ip, k_ip, h_ip, 10, 1000)
    torch.cuda.synchronize()
    assert np.array_equal(k_in.cpu().numpy(), keys) and np.array_equal(h_in.cpu().numpy(), hold)  # inputs untouched
    due = hold == 0
    assert 0.2 < due.mean() < 0.5
    for rr, kk, hh in ((r_out, k_out, h_out), (r_ip, k_ip, h_ip)):
        assert np.array_equal(kk.cpu().numpy(), want_keys) and np.array_equal(hh.cpu().numpy(), want_hold)
        for n, got_t, want in zip(cs, rr, want_refs):
            got = got_t.cpu().numpy()
            assert np.array_equal(got[~due], want[~due]), n
            if env_name == "pmsm":
                close = np.abs(got - want) <= _ball_tolerance(dtype) * 250.0
                assert close.mean() > 0.999, (n, float(np.abs(got - want).max()))
            else:
                assert np.array_equal(got, want), n


def test_gym_wrapper_reference_stream_follows_the_oracle_over_many_steps():
    """GymWrapper.reset(rng_ref=key) + 300 steps on the device (one update_ref launch per step): the key of every environment, its
    hold counter and its references equal the oracle's update_ref applied 300 times to the same start."""
    from exciting_environments_amd import GymWrapper
    from exciting_environments_amd import random as jr

    B, steps = 512, 300
    env, props, keep, spec = make_env("pendulum", B, torch.float64)
    gw = GymWrapper(env=env, control_state=["theta"], ref_params={"hold_steps_min": 3, "hold_steps_max": 40})
    key = jr.PRNGKey(9, device=env.device)
    gw.reset(rng_ref=key)
    # the wrapper's own state after the reset: keys [B, 2], hold counters [B, 1], reference theta [B]
    keys_np = gw.state.PRNGKey.cpu().numpy().copy()
    hold_np = gw.reference_hold_steps.reshape(B).cpu().numpy().copy()
    ref_np = gw.state.reference.theta.cpu().numpy().copy()
    act = torch.zeros((B, 1), dtype=torch.float64, device=env.device)
    for _ in range(steps):
        gw.step(act)
    for _ in range(steps):
        (ref_np,), keys_np, hold_np = oracle.update_ref("pendulum", [0], [ref_np], keys_np, hold_np, props, np.float64, 3, 40)
    assert np.array_equal(gw.state.PRNGKey.cpu().numpy(), keys_np)
    assert np.array_equal(gw.reference_hold_steps.reshape(B).cpu().numpy(), hold_np)
    assert np.array_equal(gw.state.reference.theta.cpu().numpy(), ref_np)
    assert len(np.unique(hold_np)) > 5 and not np.array_equal(keys_np, oracle.split(oracle.prng_key(9), B))  # the stream moved
