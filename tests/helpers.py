"""Shared helpers for the parity tests: build the same environment twice — through the product package
(HIP kernels) and as oracle properties — from one specification, and draw seeded states / actions."""
import numpy as np
import torch

import oracle
from conftest import load_golden

REG_NAME = {
    "pendulum": "PENDULUM", "mass_spring_damper": "MASS_SPRING_DAMPER", "cartpole": "CART_POLE",
    "acrobot": "ACROBOT", "fluid_tank": "FLUID_TANK", "pmsm": "PMSM",
}
NP_DTYPE = {torch.float32: np.float32, torch.float64: np.float64}
ANGLE_STATES = {"pendulum": [0], "cartpole": [2], "acrobot": [0, 1], "pmsm": [2]}
# columns of the observation that hold a normalised wrapped angle (pmsm exposes cos/sin instead)
ANGLE_OBS = {"pendulum": [0], "cartpole": [2], "acrobot": [0, 1]}
TRIG_FREE = ("mass_spring_damper", "fluid_tank")


def spec_of(env_name):
    """Default-like specification (the fixture's sim_properties.json == the reference defaults, except tau)."""
    g = load_golden(env_name)
    return dict(params=dict(g["params"]), phys_norm=dict(g["phys_norm"]), act_norm=dict(g["act_norm"]), tau=g["tau"])


def make_env(env_name, B, dtype, solver="euler", spec=None, control_state=None, device="cuda"):
    import exciting_environments_amd as ex
    from exciting_environments_amd import EnvironmentRegistry, MinMaxNormalization

    spec = spec or spec_of(env_name)
    solv = {"euler": ex.Euler(), "rk4": ex.RK4(), "tsit5": ex.Tsit5()}[solver]

    def dev(v):
        if isinstance(v, np.ndarray):
            return torch.as_tensor(v, dtype=dtype, device=device)
        return v

    env = getattr(EnvironmentRegistry, REG_NAME[env_name]).make(
        batch_size=B, tau=spec["tau"], solver=solv, dtype=dtype, device=device,
        static_params={k: dev(v) for k, v in spec["params"].items()},
        physical_normalizations={k: MinMaxNormalization(dev(lo), dev(hi)) for k, (lo, hi) in spec["phys_norm"].items()},
        action_normalizations={k: MinMaxNormalization(dev(lo), dev(hi)) for k, (lo, hi) in spec["act_norm"].items()},
        control_state=control_state,
    )
    props, keep = oracle.make_props(env_name, spec["params"], spec["phys_norm"], spec["act_norm"], NP_DTYPE[dtype], B)
    return env, props, keep, spec


def random_state(env_name, B, np_dtype, spec, seed):
    """Seeded physical states inside the normalisation box (PMSM: stable speeds <= 600 rad/s)."""
    rng = np.random.default_rng(seed)
    out = []
    for name in oracle.STATE_FIELDS[env_name]:
        lo, hi = spec["phys_norm"][name]
        lo, hi = np.broadcast_to(np.asarray(lo, dtype=np.float64), (B,)), np.broadcast_to(np.asarray(hi, dtype=np.float64), (B,))
        x = rng.uniform(-0.9, 0.9, B)
        v = (x + 1) / 2 * (hi - lo) + lo
        if env_name == "pmsm":
            if name == "omega_el":
                v = rng.uniform(0, 600, B)
            elif name in ("u_d_buffer", "u_q_buffer"):
                v = rng.uniform(-100, 100, B)
            elif name == "i_d":
                v = rng.uniform(-200, -50, B)
            elif name == "i_q":
                v = rng.uniform(-100, 100, B)
        out.append(v.astype(np_dtype))
    return out


def to_state(env, st_np, reference=None):
    """Build the package's State pytree from host arrays."""
    _, state = env.vmap_reset()
    for n, v in zip(env.STATE_FIELDS, st_np):
        setattr(state.physical_state, n, torch.as_tensor(v, dtype=env.dtype, device=env.device))
    if reference:
        for n, v in reference.items():
            setattr(state.reference, n, torch.as_tensor(v, dtype=env.dtype, device=env.device))
    return state


def phys_np(env, state):
    return [getattr(state.physical_state, n).cpu().numpy() for n in env.STATE_FIELDS]


def circ_close(got, want, cols, rtol, atol, period=2.0):
    """allclose where the listed columns (last axis) are compared on a circle of the given period."""
    got, want = np.array(got, dtype=np.float64), np.array(want, dtype=np.float64)
    for c in cols:
        d = np.abs(got[..., c] - want[..., c])
        d = np.minimum(d, np.abs(period - d))
        if not np.all(d <= atol + rtol * np.abs(want[..., c])):
            return False
        got[..., c] = want[..., c]
    return np.allclose(got, want, rtol=rtol, atol=atol, equal_nan=True)


def max_err(got, want):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    return float(np.nanmax(np.abs(got - want))) if got.size else 0.0
