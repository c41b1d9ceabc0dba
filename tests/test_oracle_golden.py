"""Pins the CPU oracle to the reference's golden fixtures.

Mirrors the reference's test_step_results (e.g. tests/envs/pendulum/test_pendulum.py:132-156,
tests/envs/pmsm/test_pmsm.py:150-174): state from stored_observations[0], explicit Euler, fp64,
10 000 (PMSM 1 000) single-env steps, jnp.allclose(rtol=1e-16 | 1e-8, atol=1e-8).
"""
import os

import numpy as np
import pytest

import oracle
from conftest import ENV_NAMES, golden_rtol


def _setup(env, g, B=1, dtype=np.float64):
    props, keep = oracle.make_props(env, g["params"], g["phys_norm"], g["act_norm"], dtype, B)
    st = [np.full(B, float(s), dtype=dtype) for s in oracle.state_from_observation(env, g["observations"][0], g["phys_norm"])]
    return props, keep, st


@pytest.mark.parametrize("env", ENV_NAMES)
def test_step_path_reproduces_fixture(env, golden):
    g = golden[env]
    props, keep, st = _setup(env, g)
    gen = [g["observations"][0]]
    for a in g["actions"]:
        obs, st = oracle.step(env, "euler", st, a[None, :], props, g["tau"])
        gen.append(obs[0])
    gen = np.array(gen)
    assert np.allclose(gen, g["observations"], rtol=golden_rtol(env), atol=1e-8)


@pytest.mark.parametrize("env", ENV_NAMES)
def test_sim_ahead_step_semantics_is_bitwise_k_steps(env, golden):
    """SEM_STEP sim_ahead == K x step, bit for bit, and therefore also reproduces the fixture; B-replication
    pins the vmapped form (SURVEY.md §8c last row)."""
    g = golden[env]
    B = 3
    props, keep, st = _setup(env, g, B=B)
    acts = np.repeat(g["actions"][None], B, axis=0)
    obs, straj, last = oracle.sim_ahead(env, "euler", st, acts, props, g["tau"], semantics=oracle.SEM_STEP)
    assert obs.shape == (B,) + g["observations"].shape
    for b in range(B):
        assert np.allclose(obs[b], g["observations"], rtol=golden_rtol(env), atol=1e-8)
        assert np.array_equal(obs[b], obs[0])
    s1 = [s[:1].copy() for s in st]
    for k in range(50):
        o, s1 = oracle.step(env, "euler", s1, g["actions"][k][None, :], props, g["tau"])
        assert np.array_equal(o[0], obs[0, k + 1])
    for j in range(len(st)):
        assert np.array_equal(last[j], straj[j][:, -1])


@pytest.mark.parametrize("env", ENV_NAMES)
def test_sim_ahead_reference_structure_matches_fixture_modulo_wrap(env, golden):
    """SEM_AHEAD (post-processing on saved rows only) stays within the reference's fixture tolerance;
    angles are compared modulo 2*pi because wrap(pi) == -pi at the +-pi seam."""
    g = golden[env]
    props, keep, st = _setup(env, g)
    obs, _, _ = oracle.sim_ahead(env, "euler", st, g["actions"][None], props, g["tau"], semantics=oracle.SEM_AHEAD)
    got, want = obs[0].copy(), g["observations"].copy()
    angle_cols = {"pendulum": [0], "cartpole": [2], "acrobot": [0, 1]}.get(env, [])
    for c in angle_cols:  # normalised angle in [-1, 1] <-> compare on the circle
        d = np.abs(got[:, c] - want[:, c])
        d = np.minimum(d, 2.0 - d)
        assert d.max() < 1e-8
        got[:, c] = want[:, c]
    assert np.allclose(got, want, rtol=golden_rtol(env), atol=1e-8)


def test_step_equals_ahead_property_reference_inputs(golden):
    """Reference property test tests/envs/test_core_functions.py:134-155: Euler, 10 steps of `ones`
    actions from the default reset, last obs of sim_ahead allclose(rtol 1e-16, atol 1e-8) to the 10th stepped obs."""
    defaults = {
        "pendulum": [np.pi, 0.0],
        "mass_spring_damper": [0.0, 0.0],
        "cartpole": [0.0, 0.0, np.pi, 0.0],
        "acrobot": [np.pi, 0.0, 0.0, 0.0],
        "fluid_tank": [1.5],
        "pmsm": [0.0, 0.0, 0.0, -125.0, 0.0, 0.0, 3 * 11000 * 2 * np.pi / 60 / 2],
    }
    for env in ENV_NAMES:
        g = golden[env]
        props, keep, _ = _setup(env, g)
        st = [np.array([v]) for v in defaults[env]]
        A = len(oracle.ACTION_FIELDS[env])
        acts = np.ones((1, 10, A))
        obs_a, _, last = oracle.sim_ahead(env, "euler", st, acts, props, g["tau"], semantics=oracle.SEM_AHEAD)
        s = st
        for _ in range(10):
            o, s = oracle.step(env, "euler", s, np.ones((1, A)), props, g["tau"])
        got, want = obs_a[0, -1].copy(), o[0]
        assert np.allclose(got, want, rtol=1e-16, atol=1e-8), env


def test_layouts_agree(golden):
    """lane-major ([K][C][B]) and env-major ([B][K][C]) trajectories hold the same numbers."""
    import ctypes
    env = "pmsm"
    g = golden[env]
    B, K = 5, 17
    rng = np.random.default_rng(0)
    props, keep, st = _setup(env, g, B=B)
    acts = rng.uniform(-1, 1, (B, K, 2))
    obs, straj, last = oracle.sim_ahead(env, "euler", st, acts, props, g["tau"])
    acts_lm = np.ascontiguousarray(acts.transpose(1, 2, 0))
    obs_lm = np.empty((K + 1, 8, B))
    straj_lm = [np.empty((K + 1, B)) for _ in range(7)]
    last_lm = [np.empty(B) for _ in range(7)]
    rc = oracle.lib().oracle_sim_ahead(
        ctypes.c_int(5), ctypes.c_int(0), ctypes.c_int(1), ctypes.c_int64(B), ctypes.c_int64(K), ctypes.c_int32(1),
        ctypes.byref(props), None, ctypes.c_double(g["tau"]), ctypes.c_double(g["tau"]), oracle._ptr_array(st),
        ctypes.c_void_p(acts_lm.ctypes.data), ctypes.c_int(oracle.LAYOUT_LANE_MAJOR),
        ctypes.c_void_p(obs_lm.ctypes.data), oracle._ptr_array(straj_lm), ctypes.c_int(oracle.LAYOUT_LANE_MAJOR),
        oracle._ptr_array(last_lm), ctypes.c_int(oracle.SEM_STEP))
    assert rc == 0
    assert np.array_equal(obs_lm.transpose(2, 0, 1), obs)
    for j in range(7):
        assert np.array_equal(straj_lm[j].T, straj[j])
        assert np.array_equal(last_lm[j], last[j])


def test_sqrt_free_form_of_the_pmsm_flag_predicate():
    """devmath.hpp sqrt_exceeds_one: the PMSM kernels test `s > nextafter(1)` where the reference (and the oracle) test `sqrt(s) > 1`
    (pmsm_env.py:972-983). With a correctly rounded square root the two are the same predicate for every s — exhaustively over the
    2^21 representable values around 1 in both precisions, over random values and the special ones."""
    import re

    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "exciting-environments_amd", "csrc", "devmath.hpp")).read()
    thr = dict(re.findall(r"sqrt_exceeds_one\((float|double) s\) \{ return s > ([0-9.]+)f?;", src))
    for np_t, bits_t, key in ((np.float32, np.uint32, "float"), (np.float64, np.uint64, "double")):
        one = np_t(1.0)
        nxt = np.nextafter(one, np_t(2.0))
        assert np_t(float(thr[key])) == nxt, (key, thr[key])  # the constant in the source is exactly nextafter(1)
        b1 = int(one.view(bits_t))
        x = np.arange(b1 - (1 << 20), b1 + (1 << 20), dtype=bits_t).view(np_t)
        assert ((np.sqrt(x) > one) == (x > nxt)).all()
        y = np.concatenate([np.random.default_rng(3).uniform(0, 4, 2_000_000).astype(np_t),
                            np.array([0.0, np.inf, np.nan, np.finfo(np_t).tiny, np.finfo(np_t).max, 1.0], dtype=np_t)])
        assert ((np.sqrt(y) > one) == (y > nxt)).all()
