"""ORACLE — test infrastructure only (see excenv_oracle.c header).

ctypes/numpy front end of liboracle.so, the plain-C CPU restatement of the reference's
vmap_step / vmap_sim_ahead arithmetic. Importable only from tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg; the product package never imports it.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

MAX_STATE, MAX_ACTION, MAX_STATIC, MAX_CONTROL = 8, 2, 9, 8

ENV_IDS = {
    "pendulum": 0,
    "mass_spring_damper": 1,
    "cartpole": 2,
    "acrobot": 3,
    "fluid_tank": 4,
    "pmsm": 5,
}
SOLVER_IDS = {"euler": 0, "rk4": 1, "tsit5": 2}
DTYPE_IDS = {np.dtype(np.float32): 0, np.dtype(np.float64): 1}
LAYOUT_ENV_MAJOR, LAYOUT_LANE_MAJOR = 0, 1
SEM_STEP, SEM_AHEAD = 0, 1
SEM_AHEAD_ACCUMULATED_T = 2  # oracle-only experiment: diffrax's accumulated stage times decide the action index (oracle_body.inc)

# (S, A, O, P) and field orders, mirroring include/excenv.h
ENV_DIMS = {0: (2, 1, 2, 3), 1: (2, 1, 2, 3), 2: (4, 1, 4, 6), 3: (4, 1, 4, 9), 4: (1, 1, 1, 4), 5: (7, 2, 8, 7)}
STATE_FIELDS = {
    "pendulum": ["theta", "omega"],
    "mass_spring_damper": ["deflection", "velocity"],
    "cartpole": ["deflection", "velocity", "theta", "omega"],
    "acrobot": ["theta_1", "theta_2", "omega_1", "omega_2"],
    "fluid_tank": ["height"],
    "pmsm": ["u_d_buffer", "u_q_buffer", "epsilon", "i_d", "i_q", "torque", "omega_el"],
}
ACTION_FIELDS = {
    "pendulum": ["torque"],
    "mass_spring_damper": ["force"],
    "cartpole": ["force"],
    "acrobot": ["torque"],
    "fluid_tank": ["inflow"],
    "pmsm": ["u_d", "u_q"],
}
PARAM_FIELDS = {
    "pendulum": ["g", "l", "m"],
    "mass_spring_damper": ["d", "k", "m"],
    "cartpole": ["mu_p", "mu_c", "l", "m_p", "m_c", "g"],
    "acrobot": ["g", "l_1", "l_2", "m_1", "m_2", "l_c1", "l_c2", "I_1", "I_2"],
    "fluid_tank": ["base_area", "orifice_area", "c_d", "g"],
    "pmsm": ["p", "r_s", "l_d", "l_q", "psi_p", "u_dc", "deadtime"],
}


class Param(ctypes.Structure):
    _fields_ = [("value", ctypes.c_double), ("per_env", ctypes.c_void_p)]


class PmsmLut(ctypes.Structure):
    _fields_ = [("n_d", ctypes.c_int32), ("n_q", ctypes.c_int32), ("grid_d", ctypes.c_void_p),
                ("grid_q", ctypes.c_void_p), ("tables", ctypes.c_void_p)]


class Props(ctypes.Structure):
    _fields_ = [
        ("static_params", Param * MAX_STATIC),
        ("state_min", Param * MAX_STATE),
        ("state_max", Param * MAX_STATE),
        ("action_min", Param * MAX_ACTION),
        ("action_max", Param * MAX_ACTION),
        ("pmsm_lut", ctypes.POINTER(PmsmLut)),
    ]


class Control(ctypes.Structure):
    _fields_ = [
        ("n_control", ctypes.c_int32),
        ("control_idx", ctypes.c_int32 * MAX_CONTROL),
        ("reference", ctypes.c_void_p * MAX_CONTROL),
        ("obs_reference", ctypes.c_void_p * MAX_CONTROL),  # unused by the oracle (excenv_control_t layout)
    ]


def build(force: bool = False) -> str:
    """Compile liboracle.so with the committed Makefile (gcc)."""
    stale = not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
        for f in ("excenv_oracle.c", "oracle_body.inc", "oracle_rng.inc", "Makefile")
    )
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-B", "liboracle.so"], check=True, capture_output=True)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.oracle_step.restype = ctypes.c_int
        _lib.oracle_sim_ahead.restype = ctypes.c_int
        _lib.oracle_num_threads.restype = ctypes.c_int
    return _lib


def num_threads() -> int:
    return int(lib().oracle_num_threads())


def set_num_threads(n: int) -> None:
    lib().oracle_set_num_threads(ctypes.c_int(n))


def _fill(param: Param, v, dtype, B, keep):
    if isinstance(v, np.ndarray) and v.ndim >= 1:
        assert v.shape[0] == B, "per-env property must have length batch_size"
        arr = np.ascontiguousarray(v, dtype=dtype)
        keep.append(arr)
        param.value = float("nan")
        param.per_env = arr.ctypes.data
    else:
        param.value = float(v)
        param.per_env = None


def make_props(env: str, params: dict, phys_norm: dict, act_norm: dict, dtype, B: int, pmsm_lut=None):
    """params: name->scalar|[B]; phys_norm/act_norm: name->(min,max) each scalar|[B]. Returns (Props, keepalive).
    pmsm_lut: (grid_d [n_d], grid_q [n_q], tables [n_d, n_q, 8]) host arrays for the PMSM saturated model."""
    dtype = np.dtype(dtype)
    keep: list = []
    p = Props()
    if pmsm_lut is not None:
        gd, gq, tab = (np.ascontiguousarray(a, dtype=dtype) for a in pmsm_lut)
        assert tab.shape == (gd.shape[0], gq.shape[0], 8)
        lut = PmsmLut(gd.shape[0], gq.shape[0], gd.ctypes.data, gq.ctypes.data, tab.ctypes.data)
        keep += [gd, gq, tab, lut]
        p.pmsm_lut = ctypes.pointer(lut)
    for j, name in enumerate(PARAM_FIELDS[env]):
        _fill(p.static_params[j], params[name], dtype, B, keep)
    for j, name in enumerate(STATE_FIELDS[env]):
        lo, hi = phys_norm[name]
        _fill(p.state_min[j], lo, dtype, B, keep)
        _fill(p.state_max[j], hi, dtype, B, keep)
    for j, name in enumerate(ACTION_FIELDS[env]):
        lo, hi = act_norm[name]
        _fill(p.action_min[j], lo, dtype, B, keep)
        _fill(p.action_max[j], hi, dtype, B, keep)
    return p, keep


def _ptr_array(arrs: Sequence[np.ndarray]):
    return (ctypes.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])


def _make_control(env: str, control, dtype, B, keep):
    if not control:
        return None
    c = Control()
    c.n_control = len(control)
    for j, (name, ref) in enumerate(control):
        c.control_idx[j] = STATE_FIELDS[env].index(name)
        arr = np.ascontiguousarray(np.broadcast_to(np.asarray(ref, dtype=dtype), (B,)))
        keep.append(arr)
        c.reference[j] = arr.ctypes.data
    return c


def step(env: str, solver: str, state: Sequence[np.ndarray], action: np.ndarray, props: Props, tau: float,
         control=None):
    """One vmap_step. state: S arrays [B]; action [B,A]. control: list of (field_name, ref[B]).
    Returns (obs [B,O+nc], new_state list)."""
    dtype = np.dtype(state[0].dtype)
    B = state[0].shape[0]
    eid = ENV_IDS[env]
    S, A, O, _ = ENV_DIMS[eid]
    keep: list = []
    ctl = _make_control(env, control, dtype, B, keep)
    nc = len(control) if control else 0
    st_in = [np.ascontiguousarray(s, dtype=dtype) for s in state]
    act = np.ascontiguousarray(action, dtype=dtype).reshape(B, A)
    st_out = [np.empty(B, dtype=dtype) for _ in range(S)]
    obs = np.empty((B, O + nc), dtype=dtype)
    rc = lib().oracle_step(
        ctypes.c_int(eid), ctypes.c_int(SOLVER_IDS[solver]), ctypes.c_int(DTYPE_IDS[dtype]), ctypes.c_int64(B),
        ctypes.byref(props), ctypes.byref(ctl) if ctl else None, ctypes.c_double(tau),
        _ptr_array(st_in), ctypes.c_void_p(act.ctypes.data), _ptr_array(st_out), ctypes.c_void_p(obs.ctypes.data),
    )
    if rc != 0:
        raise RuntimeError(f"oracle_step failed rc={rc}")
    return obs, st_out


def gym_step(env: str, solver: str, state: Sequence[np.ndarray], action: np.ndarray, props: Props, tau: float,
             control=None):
    """vmap_step + generate_reward / generate_terminated / generate_truncated (GymWrapper.gym_step).
    Returns (obs, new_state, reward [B,1], terminated [B,1] bool, truncated [B,TW] bool)."""
    dtype = np.dtype(state[0].dtype)
    B = state[0].shape[0]
    eid = ENV_IDS[env]
    S, A, O, _ = ENV_DIMS[eid]
    keep: list = []
    ctl = _make_control(env, control, dtype, B, keep)
    nc = len(control) if control else 0
    TW = 1 if env in ("pmsm", "fluid_tank") else O + nc
    st_in = [np.ascontiguousarray(s, dtype=dtype) for s in state]
    act = np.ascontiguousarray(action, dtype=dtype).reshape(B, A)
    st_out = [np.empty(B, dtype=dtype) for _ in range(S)]
    obs = np.empty((B, O + nc), dtype=dtype)
    reward = np.empty(B, dtype=dtype)
    term = np.empty(B, dtype=np.uint8)
    trunc = np.empty((B, TW), dtype=np.uint8)
    lib().oracle_gym_step.restype = ctypes.c_int
    rc = lib().oracle_gym_step(
        ctypes.c_int(eid), ctypes.c_int(SOLVER_IDS[solver]), ctypes.c_int(DTYPE_IDS[dtype]), ctypes.c_int64(B),
        ctypes.byref(props), ctypes.byref(ctl) if ctl else None, ctypes.c_double(tau),
        _ptr_array(st_in), ctypes.c_void_p(act.ctypes.data), _ptr_array(st_out), ctypes.c_void_p(obs.ctypes.data),
        ctypes.c_void_p(reward.ctypes.data), ctypes.c_void_p(term.ctypes.data), ctypes.c_void_p(trunc.ctypes.data),
    )
    if rc != 0:
        raise RuntimeError(f"oracle_gym_step failed rc={rc}")
    return obs, st_out, reward[:, None], term.astype(bool)[:, None], trunc.astype(bool)


def sim_ahead(env: str, solver: str, state: Sequence[np.ndarray], actions: np.ndarray, props: Props,
              obs_stepsize: float, env_tau: Optional[float] = None, substeps: int = 1, semantics: int = SEM_STEP,
              control=None, want_states: bool = True, out=None):
    """vmap_sim_ahead on host arrays in the reference (env-major) layout.
    actions [B,K,A] -> obs [B,N+1,O+nc], states list of [B,N+1], last_state list of [B]."""
    dtype = np.dtype(state[0].dtype)
    B = state[0].shape[0]
    eid = ENV_IDS[env]
    S, A, O, _ = ENV_DIMS[eid]
    actions = np.ascontiguousarray(actions, dtype=dtype)
    assert actions.ndim == 3 and actions.shape[0] == B and actions.shape[2] == A
    K = actions.shape[1]
    N = K * substeps
    keep: list = []
    ctl = _make_control(env, control, dtype, B, keep)
    nc = len(control) if control else 0
    st_in = [np.ascontiguousarray(s, dtype=dtype) for s in state]
    if out is not None:  # caller-provided (obs, straj, last) buffers: no allocation / page faults per call
        obs, straj, last = out
        assert obs.shape == (B, N + 1, O + nc) and obs.dtype == dtype and obs.flags.c_contiguous
    else:
        obs = np.empty((B, N + 1, O + nc), dtype=dtype)
        straj = [np.empty((B, N + 1), dtype=dtype) for _ in range(S)] if want_states else None
        last = [np.empty(B, dtype=dtype) for _ in range(S)]
    rc = lib().oracle_sim_ahead(
        ctypes.c_int(eid), ctypes.c_int(SOLVER_IDS[solver]), ctypes.c_int(DTYPE_IDS[dtype]), ctypes.c_int64(B),
        ctypes.c_int64(K), ctypes.c_int32(substeps), ctypes.byref(props), ctypes.byref(ctl) if ctl else None,
        ctypes.c_double(obs_stepsize), ctypes.c_double(obs_stepsize if env_tau is None else env_tau),
        _ptr_array(st_in), ctypes.c_void_p(actions.ctypes.data), ctypes.c_int(LAYOUT_ENV_MAJOR),
        ctypes.c_void_p(obs.ctypes.data), _ptr_array(straj) if straj else None, ctypes.c_int(LAYOUT_ENV_MAJOR),
        _ptr_array(last), ctypes.c_int(semantics),
    )
    if rc != 0:
        raise RuntimeError(f"oracle_sim_ahead failed rc={rc}")
    return obs, straj, last


def rew_trunc_term_ahead(env: str, states: Sequence[np.ndarray], props: Props, control=None):
    """generate_rew_trunc_term_ahead (core_env.py:490-531) on trajectories states[j] [B, rows] (env-major):
    returns (reward [B, rows-1, 1], truncated [B, rows, TW] bool, terminated [B, rows-1, 1] bool)."""
    dtype = np.dtype(states[0].dtype)
    B, rows = states[0].shape
    eid = ENV_IDS[env]
    S, A, O, _ = ENV_DIMS[eid]
    keep: list = []
    ctl = _make_control(env, control, dtype, B, keep)
    nc = len(control) if control else 0
    TW = 1 if env in ("pmsm", "fluid_tank") else O + nc
    st = [np.ascontiguousarray(s, dtype=dtype) for s in states]
    reward = np.empty((B, rows), dtype=dtype)
    term = np.empty((B, rows), dtype=np.uint8)
    trunc = np.empty((B, rows, TW), dtype=np.uint8)
    lib().oracle_rew_trunc_term.restype = ctypes.c_int
    rc = lib().oracle_rew_trunc_term(
        ctypes.c_int(eid), ctypes.c_int(DTYPE_IDS[dtype]), ctypes.c_int64(B), ctypes.c_int64(rows), ctypes.byref(props),
        ctypes.byref(ctl) if ctl else None, _ptr_array(st), ctypes.c_void_p(reward.ctypes.data),
        ctypes.c_void_p(term.ctypes.data), ctypes.c_void_p(trunc.ctypes.data))
    if rc != 0:
        raise RuntimeError(f"oracle_rew_trunc_term failed rc={rc}")
    return reward[:, 1:, None], trunc.astype(bool), term.astype(bool)[:, 1:, None]


def denormalize(x, lo, hi):
    """utils.py:16-17 on host arrays (used to bootstrap a state from a stored observation)."""
    return (x + 1) / 2 * (hi - lo) + lo


def state_from_observation(env: str, obs0: np.ndarray, phys_norm: dict):
    """generate_state_from_observation (e.g. pendulum_env.py:331-364; pmsm_env.py:921-970) for one stored row."""
    f = STATE_FIELDS[env]
    obs0 = np.asarray(obs0)  # one row [O] or a batch [..., O]
    if env == "pmsm":
        normed = {
            "u_d_buffer": obs0[..., 6], "u_q_buffer": obs0[..., 7],
            "epsilon": np.arctan2(obs0[..., 5], obs0[..., 4]) / np.asarray(np.pi, dtype=obs0.dtype),
            "i_d": obs0[..., 0], "i_q": obs0[..., 1], "torque": obs0[..., 3], "omega_el": obs0[..., 2],
        }
    else:
        normed = {name: obs0[..., j] for j, name in enumerate(f)}
    return [np.asarray(denormalize(normed[name], *phys_norm[name])) for name in f]


# ---- jax.random restated (oracle_rng.inc): keys are int64 arrays [..., 2] holding uint32 words ---------------------------
def _keys(keys) -> np.ndarray:
    k = np.ascontiguousarray(np.asarray(keys, dtype=np.int64).reshape(-1, 2))
    return k


def threefry2x32(k0: int, k1: int, c0: int, c1: int):
    out = (ctypes.c_uint32 * 2)()
    lib().oracle_threefry2x32(ctypes.c_uint32(k0), ctypes.c_uint32(k1), ctypes.c_uint32(c0), ctypes.c_uint32(c1), out)
    return int(out[0]), int(out[1])


def prng_key(seed: int) -> np.ndarray:
    """jax.random.PRNGKey(seed) key data: (seed >> 32, seed & 0xffffffff) (jax/_src/prng.py threefry_seed)."""
    seed = int(seed)
    return np.array([(seed >> 32) & 0xFFFFFFFF, seed & 0xFFFFFFFF], dtype=np.int64)


def split(keys, num: int = 2) -> np.ndarray:
    """jax.random.split(key, num) for every key of a [..., 2] batch -> [..., num, 2]."""
    lead = np.asarray(keys).shape[:-1]
    k = _keys(keys)
    out = np.empty((k.shape[0], num, 2), dtype=np.int64)
    lib().oracle_split(ctypes.c_int64(k.shape[0]), k.ctypes.data_as(ctypes.c_void_p), ctypes.c_int32(num), out.ctypes.data_as(ctypes.c_void_p))
    return out.reshape(lead + (num, 2))


def random_bits(keys, m: int, bit_width: int = 32) -> np.ndarray:
    lead = np.asarray(keys).shape[:-1]
    k = _keys(keys)
    out = np.empty((k.shape[0], m), dtype=np.int64)
    lib().oracle_random_bits(ctypes.c_int64(k.shape[0]), k.ctypes.data_as(ctypes.c_void_p), ctypes.c_int32(m), ctypes.c_int32(bit_width),
                             out.ctypes.data_as(ctypes.c_void_p))
    return out.reshape(lead + (m,))


def randint(keys, m: int, minval: int, maxval: int) -> np.ndarray:
    lead = np.asarray(keys).shape[:-1]
    k = _keys(keys)
    out = np.empty((k.shape[0], m), dtype=np.int64)
    lib().oracle_randint(ctypes.c_int64(k.shape[0]), k.ctypes.data_as(ctypes.c_void_p), ctypes.c_int32(m), ctypes.c_int32(minval),
                         ctypes.c_int32(maxval), out.ctypes.data_as(ctypes.c_void_p))
    return out.reshape(lead + (m,))


def randint64(keys, m: int, minval: int, maxval: int) -> np.ndarray:
    """jax.random.randint under jax_enable_x64 (int64 form, two 64-bit draws)."""
    lead = np.asarray(keys).shape[:-1]
    k = _keys(keys)
    out = np.empty((k.shape[0], m), dtype=np.int64)
    lib().oracle_randint64(ctypes.c_int64(k.shape[0]), k.ctypes.data_as(ctypes.c_void_p), ctypes.c_int32(m), ctypes.c_int64(minval),
                           ctypes.c_int64(maxval), out.ctypes.data_as(ctypes.c_void_p))
    return out.reshape(lead + (m,))


def erfinv(y: float) -> float:
    f = lib().oracle_erfinv
    f.restype = ctypes.c_double
    return float(f(ctypes.c_double(y)))


def _sample(which, keys, dtype, m, lo, hi, width):
    lead = np.asarray(keys).shape[:-1]
    k = _keys(keys)
    dtype = np.dtype(dtype)
    out = np.empty((k.shape[0], width), dtype=dtype)
    rc = lib().oracle_rng_sample(ctypes.c_int(which), ctypes.c_int(DTYPE_IDS[dtype]), ctypes.c_int64(k.shape[0]),
                                 k.ctypes.data_as(ctypes.c_void_p), ctypes.c_int32(m), ctypes.c_double(lo), ctypes.c_double(hi),
                                 out.ctypes.data_as(ctypes.c_void_p))
    if rc != 0:
        raise RuntimeError(f"oracle_rng_sample failed rc={rc}")
    return out.reshape(lead + (width,))


def uniform(keys, m: int, dtype=np.float32, minval=0.0, maxval=1.0) -> np.ndarray:
    """jax.random.uniform(key, (m,), dtype, minval, maxval) -> [..., m]."""
    return _sample(0, keys, dtype, m, minval, maxval, m)


def normal(keys, dtype=np.float32) -> np.ndarray:
    """jax.random.normal(key, (), dtype) -> [...]."""
    return _sample(1, keys, dtype, 1, 0.0, 0.0, 1)[..., 0]


def exponential(keys, dtype=np.float32) -> np.ndarray:
    return _sample(2, keys, dtype, 1, 0.0, 0.0, 1)[..., 0]


def gamma(keys, alpha: float, m: int, dtype=np.float32) -> np.ndarray:
    """jax.random.gamma(key, alpha, (m,), dtype) -> [..., m]."""
    return _sample(3, keys, dtype, m, alpha, 0.0, m)


def ball2(keys, dtype=np.float32) -> np.ndarray:
    """jax.random.ball(key, 2, dtype=dtype) -> [..., 2]."""
    return _sample(4, keys, dtype, 1, 0.0, 0.0, 2)


def random_state(env: str, keys, props: Props, dtype):
    """vmap_init_state(keys) (core_env.py:649-662; init_state e.g. pendulum_env.py:270-276, PMSM pmsm_env.py:402-456):
    (list of S state arrays [B], key leaf [B, 2])."""
    dtype = np.dtype(dtype)
    k = _keys(keys)
    B = k.shape[0]
    S = ENV_DIMS[ENV_IDS[env]][0]
    st = [np.empty(B, dtype=dtype) for _ in range(S)]
    leaf = np.empty((B, 2), dtype=np.int64)
    rc = lib().oracle_random_state(ctypes.c_int(ENV_IDS[env]), ctypes.c_int(DTYPE_IDS[dtype]), ctypes.c_int64(B), ctypes.byref(props),
                                   k.ctypes.data_as(ctypes.c_void_p), _ptr_array(st), leaf.ctypes.data_as(ctypes.c_void_p))
    if rc != 0:
        raise RuntimeError(f"oracle_random_state failed rc={rc}")
    return st, leaf


def update_ref(env: str, control_idx: Sequence[int], references: Sequence[np.ndarray], keys, hold, props: Props, dtype,
               hold_min: int, hold_max: int):
    """GymWrapper.update_ref over the batch (gym_wrapper.py:170-192), out of place: (new references, new keys [B, 2], new hold [B])."""
    dtype = np.dtype(dtype)
    k = _keys(keys)
    B = k.shape[0]
    hold = np.ascontiguousarray(np.asarray(hold, dtype=np.int64).reshape(B))
    refs_in = [np.ascontiguousarray(r, dtype=dtype) for r in references]
    refs_out = [np.empty(B, dtype=dtype) for _ in references]
    k_out, h_out = np.empty((B, 2), dtype=np.int64), np.empty(B, dtype=np.int64)
    idx = (ctypes.c_int32 * max(1, len(control_idx)))(*control_idx)
    rc = lib().oracle_update_ref(ctypes.c_int(ENV_IDS[env]), ctypes.c_int(DTYPE_IDS[dtype]), ctypes.c_int64(B), ctypes.byref(props),
                                 ctypes.c_int32(len(control_idx)), idx, _ptr_array(refs_in) if refs_in else None,
                                 k.ctypes.data_as(ctypes.c_void_p), hold.ctypes.data_as(ctypes.c_void_p),
                                 _ptr_array(refs_out) if refs_out else None, k_out.ctypes.data_as(ctypes.c_void_p),
                                 h_out.ctypes.data_as(ctypes.c_void_p), ctypes.c_int32(hold_min), ctypes.c_int32(hold_max))
    if rc != 0:
        raise RuntimeError(f"oracle_update_ref failed rc={rc}")
    return refs_out, k_out, h_out
