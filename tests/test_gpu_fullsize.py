"""BASELINE.json configurations at their full sizes, checked through size-independent properties:
replica consistency (a small oracle-verified set of environments tiled across the whole batch must give the same
bits in every tile), chunk chaining (last_state -> init_state, the reference's continuation mechanism,
core_env.py:484-486), and a closed-form solution for the linear system. Needs an MI355X (``-m gpu``)."""
import numpy as np
import pytest
import torch
from scipy.linalg import expm

import oracle
from helpers import make_env, random_state, spec_of, to_state

pytestmark = pytest.mark.gpu


def _tile_state(env, st_small, reps):
    _, state = env.vmap_reset()
    for n, v in zip(env.STATE_FIELDS, st_small):
        setattr(state.physical_state, n, torch.as_tensor(v, dtype=env.dtype, device=env.device).repeat(reps))
    return state


def _tile_actions(env, acts_small, reps):
    K = acts_small.shape[1]
    buf = env.new_actions_buffer(K)  # [B, K, A] view over lane-major memory
    small = torch.as_tensor(acts_small, dtype=env.dtype, device=env.device)  # [T, K, A]
    T = small.shape[0]
    base = buf.permute(1, 2, 0)  # [K, A, B] contiguous
    base.view(K, env.action_dim, reps, T).copy_(small.permute(1, 2, 0)[:, :, None, :].expand(K, env.action_dim, reps, T))
    return buf


def _all_tiles_equal(x, T):
    """x: [B, ...] with B = reps*T -> every tile of T consecutive envs equals the first one, bit for bit."""
    tiles = x.reshape((-1, T) + tuple(x.shape[1:]))
    return bool((tiles == tiles[:1]).all()) or bool(((tiles == tiles[:1]) | (tiles.isnan() & tiles[:1].isnan())).all())


def test_readme_plumbing_config_c1():
    """configs[0] (README.md:15-33): Pendulum, Euler, tau=2e-2, batch 5, 1000 steps -> five identical trajectories."""
    spec = spec_of("pendulum")
    spec["tau"] = 2e-2
    spec["act_norm"]["torque"] = (-15, 15)
    env, props, keep, _ = make_env("pendulum", 5, torch.float32, spec=spec)
    obs, state = env.vmap_reset()
    actions = torch.linspace(-1, 1, 1000, device=env.device)[None, :, None].repeat(5, 1, 1)
    stepped = [obs]
    for k in range(1000):
        obs, state = env.vmap_step(state, actions[:, k, :])
        stepped.append(obs)
    stepped = torch.stack(stepped, dim=1)
    assert stepped.shape == (5, 1001, 2)
    assert bool((stepped == stepped[:1]).all())
    st0 = [np.full(5, np.pi, dtype=np.float32), np.zeros(5, dtype=np.float32)]
    o_ref, _, _ = oracle.sim_ahead("pendulum", "euler", st0, actions.cpu().numpy(), props, 2e-2, semantics=oracle.SEM_STEP)
    d = np.abs(stepped.cpu().numpy() - o_ref)
    d[..., 0] = np.minimum(d[..., 0], 2 - d[..., 0])
    assert d.max() < 2e-4  # fp32, 1000 steps at tau = 2e-2: sin() ulps amplified along the swing
    _, s0 = env.vmap_reset()
    env.sim_ahead_semantics = "step"
    o_sa, _, _ = env.vmap_sim_ahead(s0, actions, env.tau, env.tau)
    assert torch.equal(o_sa[:, 1:], stepped[:, 1:])


def test_pmsm_euler_fp32_batch_2pow22_config_c3():
    """configs[2]: PMSM Euler fp32, B = 2^22, one 100-step chunk with full outputs."""
    B, K, T = 1 << 22, 100, 1024
    env, props_small, keep, spec = make_env("pmsm", B, torch.float32)
    _, props_T, keepT, _ = make_env("pmsm", T, torch.float32, device="cpu")
    st_small = random_state("pmsm", T, np.float32, spec, seed=101)
    acts_small = np.random.default_rng(102).uniform(-1, 1, (T, K, 2)).astype(np.float32)
    state = _tile_state(env, st_small, B // T)
    actions = _tile_actions(env, acts_small, B // T)
    assert torch.equal(actions[:T].cpu(), torch.as_tensor(acts_small)) and torch.equal(actions[B - T:].cpu(), torch.as_tensor(acts_small))

    env.sim_ahead_semantics = "ahead"
    obs, states, last = env.vmap_sim_ahead(state, actions, env.tau, env.tau)
    assert obs.shape == (B, K + 1, 8) and states.physical_state.i_d.shape == (B, K + 1)
    assert _all_tiles_equal(obs, T)
    for n in env.STATE_FIELDS:
        assert _all_tiles_equal(getattr(states.physical_state, n), T), n
        assert torch.equal(getattr(last.physical_state, n), getattr(states.physical_state, n)[:, -1])
    o_ref, s_ref, _ = oracle.sim_ahead("pmsm", "euler", st_small, acts_small, props_T, spec["tau"], semantics=oracle.SEM_AHEAD)
    assert np.allclose(obs[:T].cpu().numpy(), o_ref, rtol=1e-5, atol=1e-5)
    assert bool(torch.isfinite(obs).all())
    del obs, states

    # chunk chaining, SEM_STEP: 100 steps == 50 + 50 through last_state, bit for bit
    env.sim_ahead_semantics = "step"
    o_full, _, l_full = env.vmap_sim_ahead(state, actions, env.tau, env.tau)
    tail = o_full[:, 50:].clone()
    del o_full
    a1, a2 = env.new_actions_buffer(50), env.new_actions_buffer(50)
    a1.copy_(actions[:, :50])
    a2.copy_(actions[:, 50:])
    o1, _, l1 = env.vmap_sim_ahead(state, a1, env.tau, env.tau)
    del o1
    o2, _, l2 = env.vmap_sim_ahead(l1, a2, env.tau, env.tau)
    assert torch.equal(o2, tail)
    for n in env.STATE_FIELDS:
        assert torch.equal(getattr(l2.physical_state, n), getattr(l_full.physical_state, n))


def test_pendulum_euler_fp32_batch_2pow20_config_c2():
    """configs[1]: Pendulum Euler fp32, B = 2^20, tau = 2e-2, one 1000-step chunk."""
    B, K, T = 1 << 20, 1000, 512
    spec = spec_of("pendulum")
    spec["tau"] = 2e-2
    env, _, _, _ = make_env("pendulum", B, torch.float32, spec=spec)
    _, props_T, keepT, _ = make_env("pendulum", T, torch.float32, spec=spec, device="cpu")
    rng = np.random.default_rng(111)
    st_small = [rng.uniform(-np.pi, np.pi, T).astype(np.float32), rng.uniform(-1, 1, T).astype(np.float32)]
    acts_small = rng.uniform(-1, 1, (T, K, 1)).astype(np.float32)
    state = _tile_state(env, st_small, B // T)
    actions = _tile_actions(env, acts_small, B // T)
    env.sim_ahead_semantics = "step"
    obs, states, last = env.vmap_sim_ahead(state, actions, env.tau, env.tau)
    assert obs.shape == (B, K + 1, 2)
    assert _all_tiles_equal(obs, T) and _all_tiles_equal(states.physical_state.theta, T)
    theta = states.physical_state.theta
    assert float(theta.min()) >= -np.pi - 1e-6 and float(theta.max()) <= np.pi + 1e-6  # wrapped every step
    o_ref, _, _ = oracle.sim_ahead("pendulum", "euler", st_small, acts_small[:, :50], props_T, 2e-2, semantics=oracle.SEM_STEP)
    d = np.abs(obs[:T, :51].cpu().numpy() - o_ref)
    d[..., 0] = np.minimum(d[..., 0], 2 - d[..., 0])
    assert d.max() < 1e-5


def test_msd_tsit5_fp64_batch_2pow20_config_c4():
    """configs[3]: MassSpringDamper Tsit5 fp64, B = 2^20, 500-step chunk — tolerance check against the closed-form
    (matrix-exponential) solution for piecewise-constant force, and bit-exactness against the CPU oracle."""
    B, K, T = 1 << 20, 500, 256
    env, _, _, spec = make_env("mass_spring_damper", B, torch.float64, "tsit5")
    _, props_T, keepT, _ = make_env("mass_spring_damper", T, torch.float64, "tsit5", device="cpu")
    rng = np.random.default_rng(121)
    st_small = [rng.uniform(-1, 1, T), rng.uniform(-1, 1, T)]
    acts_small = rng.uniform(-1, 1, (T, K, 1))
    state = _tile_state(env, st_small, B // T)
    actions = _tile_actions(env, acts_small, B // T)
    env.sim_ahead_semantics = "step"
    obs, states, last = env.vmap_sim_ahead(state, actions, env.tau, env.tau)
    assert obs.shape == (B, K + 1, 2)
    assert _all_tiles_equal(obs, T)
    o_ref, s_ref, _ = oracle.sim_ahead("mass_spring_damper", "tsit5", st_small, acts_small, props_T, spec["tau"],
                                       semantics=oracle.SEM_STEP)
    assert np.array_equal(obs[:T].cpu().numpy(), o_ref)  # trig-free: bit-exact
    # closed form: x_{k+1} = Ad x_k + Bd u_k
    d, k, m, tau = 1.0, 100.0, 1.0, spec["tau"]
    M = np.array([[0, 1, 0], [-k / m, -d / m, 1 / m], [0, 0, 0]])
    E = expm(M * tau)
    Ad, Bd = E[:2, :2], E[:2, 2]
    x = np.stack(st_small, axis=1)
    u = (acts_small[..., 0] + 1) / 2 * 40 - 20
    exact = [x]
    for j in range(K):
        x = x @ Ad.T + u[:, j:j + 1] * Bd[None, :]
        exact.append(x)
    exact = np.stack(exact, axis=1)
    got = np.stack([states.physical_state.deflection[:T].cpu().numpy(), states.physical_state.velocity[:T].cpu().numpy()], axis=-1)
    assert np.abs(got - exact).max() < 1e-11


# ---------------------------------------------------------------------------------------- full-length runs
def _circ(env_name, d):
    for col in {"pendulum": [0], "cartpole": [2], "acrobot": [0, 1]}.get(env_name, []):
        d[..., col] = np.minimum(d[..., col], np.abs(2 - d[..., col]))
    return d


def _full_length_run(env_name, solver, dtype, log2_batch, n_chunks, Kc, T, tau=None, semantics="ahead", seed=0, collect=0):
    """A BASELINE configuration at its full batch AND full step count, as `n_chunks` chained vmap_sim_ahead launches of
    `Kc` steps (last_state -> init_state: the reference's continuation mechanism, core_env.py:484-486; chunking is
    forced by its max_steps = 4096 and by HBM capacity). A tile of T oracle-checkable environments is replicated across the
    batch; every chunk draws fresh actions. Per chunk:
      * every tile of the observation trajectory (and of last_state) equals the first one bit for bit,
      * last_state equals the final trajectory row and everything is finite,
      * the first tile matches the CPU oracle started from the same chunk-initial state: bit-exact for the trig-free
        systems over the whole chunk, |d| <= 1e-5 * (1 + |ref|) (normalised units: the allclose(rtol=1e-5, atol=1e-5) of the
        other parity tests) over the first 64 rows otherwise.
    collect > 0: the observation rows of the first `collect` environments over the WHOLE chained horizon are kept, and the oracle
    runs the same horizon chained on its OWN states (never restarted from the kernel's) in fp32 and fp64 — see _drift().
    Returns (env-steps simulated, None | dict(kernel=[collect, n_chunks*Kc+1, O], oracle32=..., oracle64=...))."""
    B = 1 << log2_batch
    np_dt = np.float32 if dtype == torch.float32 else np.float64
    spec = spec_of(env_name)
    if tau is not None:
        spec["tau"] = tau
    env, _, _, _ = make_env(env_name, B, dtype, solver, spec=spec)
    _, props_T, keepT, _ = make_env(env_name, T, dtype, solver, spec=spec, device="cpu")
    env.sim_ahead_semantics = semantics
    sem = oracle.SEM_AHEAD if semantics == "ahead" else oracle.SEM_STEP
    rng = np.random.default_rng(seed)
    st_small = random_state(env_name, T, np_dt, spec, seed=seed + 1)
    state = _tile_state(env, st_small, B // T)
    trig_free = env_name in ("mass_spring_damper", "fluid_tank")
    steps = 0
    prev = None
    chain = None
    if collect:
        assert collect <= T
        chain = {"kernel": [], "oracle32": [], "oracle64": []}
        o_state = {}
        o_props = {}
        for key, odt in (("oracle32", np.float32), ("oracle64", np.float64)):
            if np.dtype(odt).itemsize < np.dtype(np_dt).itemsize:
                continue  # an fp64 kernel has no fp32 floor to compare with
            o_props[key] = oracle.make_props(env_name, spec["params"], spec["phys_norm"], spec["act_norm"], odt, collect)
            o_state[key] = [np.asarray(v[:collect], dtype=odt) for v in st_small]
    for c in range(n_chunks):
        acts_small = rng.uniform(-1, 1, (T, Kc, env.action_dim)).astype(np_dt)
        actions = _tile_actions(env, acts_small, B // T)
        st0 = [getattr(state.physical_state, n)[:T].cpu().numpy() for n in env.STATE_FIELDS]
        # odd chunks write the previous chunk's buffers again (out=, with the initial state aliasing its last_state), even
        # chunks allocate: both forms of the continuation at the full size
        reuse = prev if (c % 2 == 1 and prev is not None) else None
        prev = env.vmap_sim_ahead(state, actions, env.tau, env.tau, out=reuse)
        obs, states, last = prev
        assert obs.shape == (B, Kc + 1, len(env.obs_description))
        assert _all_tiles_equal(obs, T), f"chunk {c}: observation tiles differ"
        assert bool(torch.isfinite(obs[:, -1]).all()), f"chunk {c}: non-finite observations"
        for n in env.STATE_FIELDS:
            lv = getattr(last.physical_state, n)
            assert torch.equal(lv, getattr(states.physical_state, n)[:, -1]), (c, n)
            assert _all_tiles_equal(lv, T) and bool(torch.isfinite(lv).all()), (c, n)
        o_ref, _, _ = oracle.sim_ahead(env_name, solver, st0, acts_small, props_T, spec["tau"], semantics=sem)
        got = obs[:T].cpu().numpy()
        if trig_free:
            assert np.array_equal(got, o_ref), f"chunk {c}: not bit-exact vs the oracle"
        else:
            d = _circ(env_name, np.abs(got[:, :65].astype(np.float64) - o_ref[:, :65]))
            excess = d - 1e-5 * (1.0 + np.abs(o_ref[:, :65]))
            assert excess.max() <= 0, f"chunk {c}: max |d| {d.max()}, max over tolerance {excess.max()}"
        if collect:
            first = 0 if c == 0 else 1  # row 0 of a later chunk repeats the previous chunk's last row
            chain["kernel"].append(got[:collect, first:].copy())
            for key in o_state:
                odt = o_state[key][0].dtype
                oo, _, ol = oracle.sim_ahead(env_name, solver, o_state[key], acts_small[:collect].astype(odt), o_props[key][0],
                                             spec["tau"], semantics=sem)
                chain[key].append(oo[:, first:].copy())
                o_state[key] = ol
        state = last
        steps += B * Kc
        del obs, states, actions
        if c % 2 == 1:
            prev = None  # the next (even) chunk allocates afresh; `state` keeps last_state alive
    if collect:
        chain = {k: np.concatenate(v, axis=1) for k, v in chain.items() if v}
    return steps, chain


DRIFT_ROWS = (1, 10, 100, 1000, 2000, 5000, 10000)


def _drift(name, env_name, chain):
    """Error-vs-step curves of a chained full-horizon run, in units of each observation's full scale (angles on the circle):
      a = kernel vs the oracle in the kernel's precision (what differs: device sin / cos, nothing else),
      b = kernel (fp32) vs the fp64 oracle, c = fp32 oracle vs fp64 oracle — the floor ANY fp32 implementation of the reference has.
    Per curve: the running maximum over environments and columns up to row n, and the 99th / 50th percentile over environments of
    each environment's running maximum (chaotic systems: a few environments near a separatrix carry the maximum). Written to
    gpurun_out/full_horizon_<name>.json when that directory is writable (DESIGN.md §5 tabulates it)."""
    import json
    import os

    k = chain["kernel"]
    pairs = {}
    if k.dtype == np.float32:
        pairs = {"a": (k, chain["oracle32"]), "b": (k, chain["oracle64"]), "c": (chain["oracle32"], chain["oracle64"])}
    else:
        pairs = {"a": (k, chain["oracle64"])}
    out = {"config": name, "envs": int(k.shape[0]), "rows": int(k.shape[1]), "at_rows": [r for r in DRIFT_ROWS if r < k.shape[1]]}
    for key, (x, y) in list(pairs.items()):
        dd = _circ(env_name, np.abs(x.astype(np.float64) - y.astype(np.float64)))
        for name_, d in ((key, dd.max(axis=2)), (key + "_rel", (dd / np.maximum(1.0, np.abs(y.astype(np.float64)))).max(axis=2))):  # [envs, rows]
            run = np.maximum.accumulate(d, axis=1)
            out[name_] = {"max": [float(run[:, r].max()) for r in out["at_rows"]],
                          "p99": [float(np.percentile(run[:, r], 99)) for r in out["at_rows"]],
                          "p50": [float(np.percentile(run[:, r], 50)) for r in out["at_rows"]]}
    try:
        d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, f"full_horizon_{name}.json"), "w") as f:
            json.dump(out, f, indent=1)
    except OSError:
        pass
    return out


def test_full_length_config_c2_pendulum_euler_fp32_10000_steps():
    """configs[1] end to end: Pendulum Euler fp32, B = 2^20, tau = 2e-2, 10 000 steps (10 chained 1000-step launches) — and the
    chained kernel against the chained oracle over all 10 000 rows, no restarts."""
    steps, chain = _full_length_run("pendulum", "euler", torch.float32, 20, 10, 1000, 512, tau=2e-2, seed=210, collect=256)
    assert steps == (1 << 20) * 10000
    dr = _drift("c2_pendulum_euler_f32", "pendulum", chain)
    _assert_drift(dr, C2_BOUNDS)


def test_full_length_config_c3_pmsm_euler_fp32_10000_steps():
    """configs[2] end to end: PMSM Euler fp32, B = 2^22, 10 000 steps (100 chained 100-step launches, full outputs) — and the
    chained kernel against the chained oracle over all 10 000 rows, no restarts."""
    steps, chain = _full_length_run("pmsm", "euler", torch.float32, 22, 100, 100, 1024, seed=220, collect=256)
    assert steps == (1 << 22) * 10000
    dr = _drift("c3_pmsm_euler_f32", "pmsm", chain)
    _assert_drift(dr, C3_BOUNDS)


def test_full_length_config_c4_msd_tsit5_fp64_5000_steps():
    """configs[3] end to end: MassSpringDamper Tsit5 fp64, B = 2^20, 5 000 steps (10 chained 500-step launches): the chained kernel
    equals the chained oracle bit for bit over the whole horizon."""
    steps, chain = _full_length_run("mass_spring_damper", "tsit5", torch.float64, 20, 10, 500, 256, seed=230, collect=256)
    assert steps == (1 << 20) * 5000
    assert chain["kernel"].shape == (256, 5001, 2)
    assert np.array_equal(chain["kernel"], chain["oracle64"])
    _drift("c4_msd_tsit5_f64", "mass_spring_damper", chain)


def test_full_length_pmsm_tsit5_fp32_10000_steps():
    """PMSM Tsit5 fp32 (parity-unpinned solver: kernel vs oracle only), B = 2^20, 10 000 chained steps, no restarts."""
    steps, chain = _full_length_run("pmsm", "tsit5", torch.float32, 20, 100, 100, 1024, seed=240, collect=256)
    assert steps == (1 << 20) * 10000
    dr = _drift("pmsm_tsit5_f32", "pmsm", chain)
    _assert_drift(dr, C3_BOUNDS)


# Bounds on the chained full-horizon curves (units: full scale of each normalised observation, angles on the circle).
# Measured in round 5 (256 environments, one MI355X; DESIGN.md §5 has the table):
#   C3 PMSM Euler fp32, 10 000 chained steps: a (kernel vs fp32 oracle) 2.4e-6 at row 100, 2.9e-5 at row 1 000, then a plateau —
#      5.0e-5 at row 10 000, median environment 7.6e-6 (the current dynamics forget a perturbation within ~250 steps; what
#      accumulates over that window are the <= 2 ulp differences between the device's and libm's sin / cos in the Park rotation);
#      c (fp32 oracle vs fp64 oracle, the floor of ANY fp32 implementation) grows to 4.3e-3 (median 7.6e-4): the kernel's own
#      share of the distance to the fp64 reference arithmetic is 1 %. PMSM Tsit5: a 3.2e-5, c 3.3e-3.
#   C2 pendulum Euler fp32 (tau = 2e-2, random torque, no damping: chaotic): a <= 1.9e-6 over the first 100 rows, then a and c both
#      grow to O(1) on the same curve (median environment at row 10 000: a 1.6e-2, c 2.2e-2) — no fp32 implementation can hold a fixed
#      bound there, so the assertion is that the kernel stays BELOW the fp32 floor's curve.
C3_BOUNDS = {"a_first_100": 1e-5, "a_max": 1e-4, "a_p50": 2e-5, "a_over_c_last": 0.05}
C2_BOUNDS = {"a_first_100": 1e-5, "a_over_c_p50": 1.5, "a_over_c_p99": 1.5}


def _assert_drift(dr, bounds):
    last = -1
    if "a_max" in bounds:
        assert dr["a"]["max"][last] <= bounds["a_max"], dr["a"]
    if "a_p50" in bounds:
        assert dr["a"]["p50"][last] <= bounds["a_p50"], dr["a"]
    if "a_over_c_last" in bounds:
        assert dr["a"]["max"][last] <= bounds["a_over_c_last"] * dr["c"]["max"][last], (dr["a"]["max"], dr["c"]["max"])
    if "a_first_100" in bounds:
        assert dr["a"]["max"][dr["at_rows"].index(100)] <= bounds["a_first_100"], dr["a"]
    for q in ("max", "p99", "p50"):
        if f"a_over_c_{q}" in bounds:
            for i, r in enumerate(dr["at_rows"]):
                floor = max(dr["c"][q][i], 1e-6)
                assert dr["a"][q][i] <= bounds[f"a_over_c_{q}"] * floor, (q, r, dr["a"][q][i], floor)
