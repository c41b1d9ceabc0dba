"""Randomised cross-configuration parity: every draw picks an environment, solver, dtype, batch size (ragged sizes, 1),
trajectory length (0 included), substeps, semantics, trajectory layout, a random subset of properties turned into
per-env arrays and a random control_state, runs vmap_step and vmap_sim_ahead through the C ABI and compares with the
CPU oracle. Seeds are fixed: failures reproduce. ``-m gpu``."""
import numpy as np
import pytest
import torch

import oracle
from conftest import ENV_NAMES
from helpers import ANGLE_OBS, NP_DTYPE, TRIG_FREE, circ_close, make_env, random_state, spec_of, to_state

pytestmark = pytest.mark.gpu


def _draw(rng):
    env_name = ENV_NAMES[rng.integers(len(ENV_NAMES))]
    cfg = dict(
        env_name=env_name,
        solver=["euler", "rk4", "tsit5"][rng.integers(3)],
        dtype=[torch.float32, torch.float64][rng.integers(2)],
        B=int(rng.choice([1, 2, 3, 63, 64, 65, 255, 257, 1000, 1024, 4099])),
        K=int(rng.choice([0, 1, 2, 7, 16, 33])),
        substeps=1 if env_name == "pmsm" else int(rng.choice([1, 1, 2, 3])),
        semantics=["step", "ahead"][rng.integers(2)],
        layout=["lane_major", "env_major", "env_major_ws", "env_major_strided"][rng.integers(4)],
        lane_actions=bool(rng.integers(2)),
    )
    spec = spec_of(env_name)
    B = cfg["B"]
    batched = []
    for group, keys in (("params", list(spec["params"])), ("phys_norm", list(spec["phys_norm"])), ("act_norm", list(spec["act_norm"]))):
        for k in keys:
            if rng.random() < 0.15 and not (env_name == "pmsm" and k == "deadtime"):
                if group == "params":
                    v = float(spec["params"][k])
                    spec["params"][k] = v * rng.uniform(0.8, 1.2, B) if v != 0 else rng.uniform(0.0, 1e-3, B)
                else:
                    lo, hi = spec[group][k]
                    spec[group][k] = (lo - abs(hi - lo) * rng.uniform(0, 0.1, B), hi)
                batched.append(f"{group}.{k}")
    cfg["batched"] = batched
    fields = oracle.STATE_FIELDS[env_name]
    n_ctl = int(rng.choice([0, 0, 1, 2])) if len(fields) > 1 else int(rng.choice([0, 1]))
    cfg["control"] = [fields[i] for i in rng.choice(len(fields), size=min(n_ctl, len(fields)), replace=False)]
    return cfg, spec


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("EXCENV_FUZZ_SEEDS", "60"))))
def test_random_configuration_matches_oracle(seed):
    rng = np.random.default_rng(1000 + seed)
    cfg, spec = _draw(rng)
    env_name, dtype, B, K, sub = cfg["env_name"], cfg["dtype"], cfg["B"], cfg["K"], cfg["substeps"]
    npdt = NP_DTYPE[dtype]
    env, props, keep, _ = make_env(env_name, B, dtype, cfg["solver"], spec=spec, control_state=cfg["control"] or None)
    env.sim_ahead_semantics = cfg["semantics"]
    env.traj_layout = "lane_major" if cfg["layout"] == "lane_major" else "env_major"
    env.env_major_fused = cfg["layout"] == "env_major"
    env.env_major_workspace = cfg["layout"] != "env_major_strided"
    st = random_state(env_name, B, npdt, spec, seed=2000 + seed)
    refs = {}
    for n in cfg["control"]:
        lo, hi = spec["phys_norm"][n]
        refs[n] = ((rng.uniform(-0.9, 0.9, B) + 1) / 2 * (np.asarray(hi) - np.asarray(lo)) + np.asarray(lo)).astype(npdt)
    control = [(n, refs[n]) for n in cfg["control"]]
    state = to_state(env, st, reference=refs)
    tol = 0.0 if env_name in TRIG_FREE else (1e-9 if dtype == torch.float64 else 2e-5)

    def close(got, want):
        if tol == 0.0:
            return np.array_equal(got, want, equal_nan=True)
        return circ_close(got, want, ANGLE_OBS.get(env_name, []), tol, tol)

    act = rng.uniform(-1.1, 1.1, (B, env.action_dim)).astype(npdt)
    obs, new = env.vmap_step(state, torch.as_tensor(act, device=env.device))
    o_ref, s_ref = oracle.step(env_name, cfg["solver"], st, act, props, spec["tau"], control=control)
    assert obs.shape == o_ref.shape and close(obs.cpu().numpy(), o_ref), (cfg, "step")

    acts = rng.uniform(-1, 1, (B, K, env.action_dim)).astype(npdt)
    a_dev = torch.as_tensor(acts, device=env.device)
    if cfg["lane_actions"] and K > 0:
        buf = env.new_actions_buffer(K)
        buf.copy_(a_dev)
        a_dev = buf
    o, s, l = env.vmap_sim_ahead(state, a_dev, env.tau / sub, env.tau)
    sem = oracle.SEM_STEP if cfg["semantics"] == "step" else oracle.SEM_AHEAD
    o_ref, s_ref, l_ref = oracle.sim_ahead(env_name, cfg["solver"], st, acts, props, spec["tau"] / sub, env_tau=spec["tau"],
                                           substeps=sub, semantics=sem, control=control)
    assert tuple(o.shape) == o_ref.shape, cfg
    assert close(o.cpu().numpy(), o_ref), (cfg, "sim_ahead", float(np.nanmax(np.abs(o.cpu().numpy() - o_ref))) if o_ref.size else 0)
    for j, n in enumerate(env.STATE_FIELDS):
        got_last = getattr(l.physical_state, n).cpu().numpy()
        assert np.array_equal(getattr(s.physical_state, n)[:, -1].cpu().numpy(), got_last, equal_nan=True), (cfg, n)
        if tol == 0.0:
            assert np.array_equal(got_last, l_ref[j]), (cfg, n)
