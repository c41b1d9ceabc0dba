"""The six ODE environments — host-side mirrors of the reference classes
(pendulum/pendulum_env.py, mass_spring_damper/mass_spring_damper_env.py, cart_pole/cart_pole_env.py,
acrobot/acrobot_env.py, fluid_tank/fluid_tank_env.py, pmsm/pmsm_env.py): same constructor keywords,
defaults, dataclass field names and orders. The vector fields themselves are HIP device code
(csrc/models.hpp)."""
from __future__ import annotations

import math
from dataclasses import dataclass, make_dataclass, replace
from enum import Enum
from typing import Any, Callable

import ctypes

import numpy as np
import torch

from . import _native
from .core_env import CoreEnvironment
from .solvers import Euler
from .utils import MinMaxNormalization


def _dc(name, field_names, doc):
    cls = make_dataclass(name, [(f, Any) for f in field_names])
    cls.__doc__ = doc
    return cls


@dataclass
class _Additions:
    """Dataclass containing additional information for simulation (e.g. pendulum_env.py:123-128). `solver_state` has the
    reference's structure for the chosen solver (None for Euler; a NaN-filled (first_step, f0) pair for Tsit5 —
    CoreEnvironment._solver_state_leaf); the fixed-step kernels themselves carry no solver state."""

    solver_state: Any
    active_solver_state: Any


class _SimpleEnv(CoreEnvironment):
    """Shared constructor of the five single-action environments (e.g. pendulum_env.py:52-114)."""

    DEFAULT_PHYSICAL_NORMALIZATIONS: dict = {}
    DEFAULT_ACTION_NORMALIZATIONS: dict = {}
    DEFAULT_STATIC_PARAMS: dict = {}
    DEFAULT_BATCH_SIZE = 8
    DEFAULT_TAU = 1e-4
    Additions = _Additions

    def __init__(self, batch_size: int = None, physical_normalizations: dict = None,
                 action_normalizations: dict = None, soft_constraints: Callable = None, static_params: dict = None,
                 control_state: list = None, solver=Euler(), tau: float = None, dtype=torch.float32, device=None):
        if batch_size is None:
            batch_size = self.DEFAULT_BATCH_SIZE
        if tau is None:
            tau = self.DEFAULT_TAU
        if not physical_normalizations:
            physical_normalizations = {k: MinMaxNormalization(*v) for k, v in self.DEFAULT_PHYSICAL_NORMALIZATIONS.items()}
        if not action_normalizations:
            action_normalizations = {k: MinMaxNormalization(*v) for k, v in self.DEFAULT_ACTION_NORMALIZATIONS.items()}
        if not soft_constraints:
            soft_constraints = None
        if not static_params:
            static_params = dict(self.DEFAULT_STATIC_PARAMS)
        if not control_state:
            control_state = []
        self.control_state = control_state
        self.soft_constraints = soft_constraints
        env_properties = self.EnvProperties(
            physical_normalizations=self.PhysicalState(**physical_normalizations),
            action_normalizations=self.Action(**action_normalizations),
            static_params=self.StaticParams(**static_params),
        )
        super().__init__(batch_size, env_properties=env_properties, tau=tau, solver=solver, dtype=dtype, device=device)


class Pendulum(_SimpleEnv):
    """State ``['theta', 'omega']``, action ``['torque']``; default reset theta=pi, omega=0
    (pendulum_env.py:19-100)."""

    ENV_ID = 0
    STATE_FIELDS = ("theta", "omega")
    ACTION_FIELDS = ("torque",)
    PARAM_FIELDS = ("g", "l", "m")
    DEFAULT_NORM_STATE = (1.0, 0.0)
    ANGLE_FIELDS = ("theta",)
    PhysicalState = _dc("PhysicalState", STATE_FIELDS, "Physical state of the pendulum.")
    Action = _dc("Action", ACTION_FIELDS, "Action of the pendulum.")
    StaticParams = _dc("StaticParams", PARAM_FIELDS, "Static parameters of the pendulum.")
    DEFAULT_PHYSICAL_NORMALIZATIONS = {"theta": (-math.pi, math.pi), "omega": (-10, 10)}
    DEFAULT_ACTION_NORMALIZATIONS = {"torque": (-20, 20)}
    DEFAULT_STATIC_PARAMS = {"g": 9.81, "l": 2, "m": 1}


class MassSpringDamper(_SimpleEnv):
    """State ``['deflection', 'velocity']``, action ``['force']`` (mass_spring_damper_env.py:50-98)."""

    ENV_ID = 1
    STATE_FIELDS = ("deflection", "velocity")
    ACTION_FIELDS = ("force",)
    PARAM_FIELDS = ("d", "k", "m")
    DEFAULT_NORM_STATE = (0.0, 0.0)
    PhysicalState = _dc("PhysicalState", STATE_FIELDS, "Physical state of the mass-spring-damper.")
    Action = _dc("Action", ACTION_FIELDS, "Action of the mass-spring-damper.")
    StaticParams = _dc("StaticParams", PARAM_FIELDS, "Static parameters of the mass-spring-damper.")
    DEFAULT_PHYSICAL_NORMALIZATIONS = {"deflection": (-10, 10), "velocity": (-10, 10)}
    DEFAULT_ACTION_NORMALIZATIONS = {"force": (-20, 20)}
    DEFAULT_STATIC_PARAMS = {"k": 100, "d": 1, "m": 1}


class CartPole(_SimpleEnv):
    """State ``['deflection', 'velocity', 'theta', 'omega']``, action ``['force']``; default tau 2e-2
    (cart_pole_env.py:50-110)."""

    ENV_ID = 2
    STATE_FIELDS = ("deflection", "velocity", "theta", "omega")
    ACTION_FIELDS = ("force",)
    PARAM_FIELDS = ("mu_p", "mu_c", "l", "m_p", "m_c", "g")
    DEFAULT_NORM_STATE = (0.0, 0.0, 1.0, 0.0)
    ANGLE_FIELDS = ("theta",)
    DEFAULT_TAU = 2e-2
    PhysicalState = _dc("PhysicalState", STATE_FIELDS, "Physical state of the cart-pole.")
    Action = _dc("Action", ACTION_FIELDS, "Action of the cart-pole.")
    StaticParams = _dc("StaticParams", PARAM_FIELDS, "Static parameters of the cart-pole.")
    DEFAULT_PHYSICAL_NORMALIZATIONS = {"deflection": (-2.4, 2.4), "velocity": (-8, 8), "theta": (-math.pi, math.pi),
                                       "omega": (-8, 8)}
    DEFAULT_ACTION_NORMALIZATIONS = {"force": (-20, 20)}
    DEFAULT_STATIC_PARAMS = {"mu_p": 0.000002, "mu_c": 0.0005, "l": 0.5, "m_p": 0.1, "m_c": 1, "g": 9.81}


class Acrobot(_SimpleEnv):
    """State ``['theta_1', 'theta_2', 'omega_1', 'omega_2']``, action ``['torque']``; default tau 1e-3
    (acrobot_env.py:50-133)."""

    ENV_ID = 3
    STATE_FIELDS = ("theta_1", "theta_2", "omega_1", "omega_2")
    ACTION_FIELDS = ("torque",)
    PARAM_FIELDS = ("g", "l_1", "l_2", "m_1", "m_2", "l_c1", "l_c2", "I_1", "I_2")
    DEFAULT_NORM_STATE = (1.0, 0.0, 0.0, 0.0)
    ANGLE_FIELDS = ("theta_1", "theta_2")
    DEFAULT_TAU = 1e-3
    PhysicalState = _dc("PhysicalState", STATE_FIELDS, "Physical state of the acrobot.")
    Action = _dc("Action", ACTION_FIELDS, "Action of the acrobot.")
    StaticParams = _dc("StaticParams", PARAM_FIELDS, "Static parameters of the acrobot.")
    DEFAULT_PHYSICAL_NORMALIZATIONS = {"theta_1": (-math.pi, math.pi), "theta_2": (-math.pi, math.pi),
                                       "omega_1": (-10, 10), "omega_2": (-10, 10)}
    DEFAULT_ACTION_NORMALIZATIONS = {"torque": (-20, 20)}
    DEFAULT_STATIC_PARAMS = {"g": 9.81, "l_1": 2, "l_2": 2, "m_1": 1, "m_2": 1, "l_c1": 1, "l_c2": 1, "I_1": 1.3,
                             "I_2": 1.3}


class FluidTank(_SimpleEnv):
    """State ``['height']``, action ``['inflow']``; default batch 1, tau 1e-3, reset at normalised height 0.0
    i.e. h = 1.5 m (fluid_tank_env.py:23-68,218-224)."""

    ENV_ID = 4
    STATE_FIELDS = ("height",)
    ACTION_FIELDS = ("inflow",)
    PARAM_FIELDS = ("base_area", "orifice_area", "c_d", "g")
    DEFAULT_NORM_STATE = (0.0,)
    DEFAULT_BATCH_SIZE = 1
    DEFAULT_TAU = 1e-3
    PhysicalState = _dc("PhysicalState", STATE_FIELDS, "Physical state of the fluid tank.")
    Action = _dc("Action", ACTION_FIELDS, "Action of the fluid tank.")
    StaticParams = _dc("StaticParams", PARAM_FIELDS, "Static parameters of the fluid tank.")
    DEFAULT_PHYSICAL_NORMALIZATIONS = {"height": (0, 3)}
    DEFAULT_ACTION_NORMALIZATIONS = {"inflow": (0, 0.2)}
    DEFAULT_STATIC_PARAMS = {"base_area": math.pi, "orifice_area": math.pi * 0.1**2, "c_d": 0.6, "g": 9.81}

    def generate_truncated(self, state, env_properties):
        """fluid_tank_env.py:325-328: constant 0."""
        h = torch.as_tensor(state.physical_state.height)
        return torch.zeros(tuple(h.shape) + (1,), dtype=torch.bool, device=h.device)

    def generate_terminated(self, state, reward, env_properties):
        """fluid_tank_env.py:330-333: constant False."""
        return torch.zeros_like(reward, dtype=torch.bool)

    @property
    def states_description(self):
        return np.array(["fluid height"])

    @property
    def obs_description(self):
        return np.hstack([self.states_description, np.array([n + "_ref" for n in self.control_state])])


# ---------------------------------------------------------------------------------------------- PMSM
class MotorVariant(Enum):
    """pmsm/motor_parameters.py:152-163. The saturated (LUT) models of BRUSA / SEW are outside the hot path
    (SURVEY.md §2); their linear parameter sets are provided."""

    DEFAULT = "DEFAULT"
    BRUSA = "BRUSA"
    SEW = "SEW"

    def get_params(self):
        return _motor_params(self)


@dataclass
class MotorParams:
    physical_normalizations: dict
    action_normalizations: dict
    static_params: dict


def _motor_params(variant: MotorVariant) -> MotorParams:
    """pmsm/motor_parameters.py:70-149."""
    if variant is MotorVariant.SEW:
        u, i_d, i_q, om, tq = 2 * 550 / 3, (-16, 0), (-16, 16), 4 * 2000 / 60 * 2 * math.pi, 15
        sp = dict(p=4, r_s=208e-3, l_d=1.44e-3, l_q=1.44e-3, psi_p=122e-3, u_dc=550, deadtime=1)
    else:
        u, i_d, i_q, om, tq = 2 * 400 / 3, (-250, 0), (-250, 250), 3 * 11000 * 2 * math.pi / 60, 200
        if variant is MotorVariant.BRUSA:
            sp = dict(p=3, r_s=17.932e-3, l_d=0.37e-3, l_q=1.2e-3, psi_p=65.65e-3, u_dc=400, deadtime=1)
        else:
            sp = dict(p=3, r_s=15e-3, l_d=0.37e-3, l_q=1.2e-3, psi_p=65.6e-3, u_dc=400, deadtime=1)
    pn = {
        "u_d_buffer": MinMaxNormalization(-u, u), "u_q_buffer": MinMaxNormalization(-u, u),
        "epsilon": MinMaxNormalization(-math.pi, math.pi), "i_d": MinMaxNormalization(*i_d),
        "i_q": MinMaxNormalization(*i_q), "omega_el": MinMaxNormalization(0, om),
        "torque": MinMaxNormalization(-tq, tq),
    }
    an = {"u_d": MinMaxNormalization(-u, u), "u_q": MinMaxNormalization(-u, u)}
    return MotorParams(pn, an, sp)


SATURATED_QUANTS = ("L_dd", "L_dq", "L_qd", "L_qq", "Psi_d", "Psi_q")


def prepare_pmsm_lut(pmsm_lut: dict):
    """Mirror of PMSM.generate_interpolators_and_lut (pmsm_env.py:316-363): NaNs of every table are filled from the nearest
    valid node, every edge is repeated once (so that linear extrapolation is constant), and the grids are extended by one
    step on each side. `pmsm_lut` has the reference's keys: i_d_vec, i_q_vec (shape (1, n) or (n,)) and the six tables
    L_dd, L_dq, L_qd, L_qq, Psi_d, Psi_q of shape (n_iq, n_id). Returns (grid_d [n_d], grid_q [n_q], tables [n_d, n_q, 8])
    as float64 numpy arrays; tables[ix, iy, k] is quantity k at (grid_d[ix], grid_q[iy])."""
    from scipy.interpolate import griddata

    i_d_vec = np.asarray(pmsm_lut["i_d_vec"], dtype=np.float64).reshape(1, -1)
    i_q_vec = np.asarray(pmsm_lut["i_q_vec"], dtype=np.float64).reshape(1, -1)
    i_d_max, i_q_max, i_d_min, i_q_min = i_d_vec.max(), i_q_vec.max(), i_d_vec.min(), i_q_vec.min()
    i_d_stepsize = (i_d_max - i_d_min) / (i_d_vec.shape[1] - 1)
    i_q_stepsize = (i_q_max - i_q_min) / (i_q_vec.shape[1] - 1)
    padded = {}
    for q in SATURATED_QUANTS:
        qmap = np.array(pmsm_lut[q], dtype=np.float64, copy=True)
        x, y = np.indices(qmap.shape)
        nan_mask = np.isnan(qmap)
        if nan_mask.any():
            qmap[nan_mask] = griddata((x[~nan_mask], y[~nan_mask]), qmap[~nan_mask], (x[nan_mask], y[nan_mask]), method="nearest")
        a = np.vstack([qmap[0, :], qmap, qmap[-1, :]])
        padded[q] = np.hstack([a[:, :1], a, a[:, -1:]])
    n_y, n_x = padded[SATURATED_QUANTS[0]].shape
    grid_d = np.linspace(i_d_min - i_d_stepsize, i_d_max + i_d_stepsize, n_x)
    grid_q = np.linspace(i_q_min - i_q_stepsize, i_q_max + i_q_stepsize, n_y)
    tables = np.zeros((n_x, n_y, 8), dtype=np.float64)
    for k, q in enumerate(SATURATED_QUANTS):
        tables[:, :, k] = padded[q].T
    return grid_d, grid_q, tables


LUT_FILE_NAMES = {"BRUSA": "LUT_BRUSA_jax_grad.mat", "SEW": "LUT_SEW_jax_grad.mat"}


def load_pmsm_lut(motor_variant, path=None) -> dict:
    """The motor's table file as the reference loads it (pmsm/motor_parameters.py:94,121: ``loadmat(<package>/LUT_<motor>_
    jax_grad.mat)``): a dict with i_d_vec, i_q_vec and the six (n_iq, n_id) tables. `path` is the file itself or a directory
    that holds it; without it the directory named by the environment variable EXCENV_PMSM_LUT_DIR and then
    ``<this package>/data`` are searched — the package ships the reference's two table files there (data only; the same bytes as
    the fixtures under tests/golden/pmsm), so ``PMSM.make(saturated=True, motor_variant=BRUSA)`` works out of the box."""
    import os

    from scipy.io import loadmat

    name = LUT_FILE_NAMES[motor_variant.value if isinstance(motor_variant, MotorVariant) else str(motor_variant)]
    tried = []
    for cand in (path, os.environ.get("EXCENV_PMSM_LUT_DIR"), os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")):
        if not cand:
            continue
        f = os.path.join(cand, name) if os.path.isdir(cand) else cand
        tried.append(f)
        if os.path.isfile(f):
            d = loadmat(f)
            return {k: v for k, v in d.items() if not k.startswith("__")}
    raise ValueError(
        f"PMSM(saturated=True): the look-up table file {name} was not found (looked at: {tried or 'nowhere'}). Pass "
        "pmsm_lut_path=<file or directory>, set EXCENV_PMSM_LUT_DIR, or pass the tables themselves as pmsm_lut=<dict>.")


class PMSM(CoreEnvironment):
    """Permanent-magnet synchronous motor in dq coordinates with voltage-hexagon clip and one-step action dead time
    (pmsm/pmsm_env.py:115-267). ``saturated=False``: linear model (`linear_ode`). ``saturated=True``: flux linkages and
    differential inductances from look-up tables (`nonlinear_ode`): the motor variant's table file is loaded like the reference
    does (``load_pmsm_lut``; ``pmsm_lut_path=`` names the file or its directory), or the tables are passed as ``pmsm_lut=``
    (a dict with the reference's keys)."""

    ENV_ID = 5
    N_ODE = 3  # the reference integrates y = (i_d, i_q, eps) (pmsm_env.py:555)
    STATE_FIELDS = ("u_d_buffer", "u_q_buffer", "epsilon", "i_d", "i_q", "torque", "omega_el")
    ACTION_FIELDS = ("u_d", "u_q")
    PARAM_FIELDS = ("p", "r_s", "l_d", "l_q", "psi_p", "u_dc", "deadtime")
    PhysicalState = _dc("PhysicalState", STATE_FIELDS, "Physical state of the PMSM.")
    Action = _dc("Action", ACTION_FIELDS, "Action of the PMSM.")
    StaticParams = _dc("StaticParams", PARAM_FIELDS, "Static parameters of the PMSM.")
    Additions = _Additions

    @dataclass
    class EnvProperties:
        """pmsm_env.py:307-314."""

        saturated: Any
        physical_normalizations: Any
        action_normalizations: Any
        static_params: Any

    def __init__(self, batch_size: int = 8, saturated=False, motor_variant: MotorVariant = MotorVariant.DEFAULT,
                 physical_normalizations: dict = None, action_normalizations: dict = None,
                 soft_constraints: Callable = None, static_params: dict = None, control_state: list = None,
                 solver=Euler(), tau: float = 1e-4, dtype=torch.float32, device=None, pmsm_lut: dict = None,
                 pmsm_lut_path: str = None):
        self._lut_host = None
        motor_params = motor_variant.get_params()
        if saturated:
            if motor_variant == MotorVariant.DEFAULT:
                raise ValueError(
                    f"MotorVariant '{motor_variant.value}' is not allowed for saturated LUTs. "
                    "Use a specific motor variant. DEFAULT is only valid for saturated=False."
                )
            if pmsm_lut is None:  # pmsm/motor_parameters.py:94,121 + pmsm_env.py:164-175: the variant's own table file
                pmsm_lut = load_pmsm_lut(motor_variant, pmsm_lut_path)
            self._lut_host = prepare_pmsm_lut(pmsm_lut)
            self.pmsm_lut = pmsm_lut
            for k in ("l_d", "l_q", "psi_p"):  # pmsm_env.py:171-174
                motor_params.static_params[k] = float("nan")
        if not static_params:
            static_params = motor_params.static_params
        if not physical_normalizations:
            physical_normalizations = motor_params.physical_normalizations
        if not action_normalizations:
            action_normalizations = motor_params.action_normalizations
        if not control_state:
            control_state = []
        self.control_state = control_state
        self.soft_constraints = soft_constraints
        env_properties = self.EnvProperties(
            saturated=saturated,
            physical_normalizations=self.PhysicalState(**physical_normalizations),
            action_normalizations=self.Action(**action_normalizations),
            static_params=self.StaticParams(**static_params),
        )
        super().__init__(batch_size, env_properties=env_properties, tau=tau, solver=solver, dtype=dtype, device=device)
        self._action_description = ["u_d", "u_q"]
        self._obs_description = ["i_d", "i_q", "cos_eps", "sin_eps", "omega_el", "torque", "u_d_buffer", "u_q_buffer"]

    def _pack_props(self, env_properties, B: int):
        p, keep = super()._pack_props(env_properties, B)
        if getattr(env_properties, "saturated", False):
            gd, gq, tab = (torch.as_tensor(a, dtype=self.dtype, device=self.device).contiguous() for a in self._lut_host)
            lut = _native.PmsmLut(gd.shape[0], gq.shape[0], gd.data_ptr(), gq.data_ptr(), tab.data_ptr())
            keep += [gd, gq, tab, lut]
            p.pmsm_lut = ctypes.pointer(lut)
        return p, keep

    def _lut_torch(self, i_d, i_q):
        """Bilinear look-up of the six quantities with torch ops (random reset only; the kernels carry their own)."""
        gd, gq, tab = (torch.as_tensor(a, dtype=self.dtype, device=self.device) for a in self._lut_host)

        def find(g, x):
            i = (torch.searchsorted(g, x.contiguous()) - 1).clamp(0, g.numel() - 2)
            return i, (x - g[i]) / (g[i + 1] - g[i])

        ix, tx = find(gd, i_d)
        iy, ty = find(gq, i_q)
        w = lambda a, b: (a * b)[..., None]
        return (tab[ix, iy] * w(1 - tx, 1 - ty) + tab[ix, iy + 1] * w(1 - tx, ty) + tab[ix + 1, iy] * w(tx, 1 - ty)
                + tab[ix + 1, iy + 1] * w(tx, ty))

    def create_in_axes_dataclass(self, dataclass_obj):
        if isinstance(dataclass_obj, PMSM.EnvProperties):  # `saturated` is a plain bool leaf
            sub = super().create_in_axes_dataclass(replace(dataclass_obj, saturated=0.0))
            return replace(sub, saturated=None)
        return super().create_in_axes_dataclass(dataclass_obj)

    def _init_state(self, env_properties, rng, shape):
        """pmsm_env.py:383-485: physical units directly (no denormalisation pass)."""
        dev = self._init_state_device_keys(env_properties, rng, shape)
        if dev is not None:
            return dev
        pn = env_properties.physical_normalizations
        full = lambda v: torch.as_tensor(v, dtype=self.dtype, device=self.device).expand(shape).clone()
        lo_hi = lambda n: (self._norm_leaf(getattr(pn, n).min), self._norm_leaf(getattr(pn, n).max))
        if rng is None:
            i_lo, i_hi = lo_hi("i_d")
            o_lo, o_hi = lo_hi("omega_el")
            phys = dict(u_d_buffer=full(0.0), u_q_buffer=full(0.0), epsilon=full(0.0), i_d=full((i_lo + i_hi) / 2),
                        i_q=full(0.0), torque=full(0.0), omega_el=full((o_lo + o_hi) / 2))
        else:
            from . import random as _random

            key_leaf = None
            if _random.is_key(rng):
                # pmsm_env.py:403-406: rng, subkey = split(rng); uniform(subkey, (2,), -1, 1); rng, subkey = split(rng);
                # ball(subkey, 2) — the whole draw follows the key stream (random.ball restates jax.random.ball's
                # gamma-rejection construction; parity unpinned, see random.py).
                k = rng.to(self.device)
                assert tuple(k.shape[:-1]) == tuple(shape), f"rng keys must have shape {tuple(shape) + (2,)}"
                s1 = _random.split(k)
                sn = _random.uniform(s1[..., 1, :], 2, self.dtype, -1.0, 1.0)
                s2 = _random.split(s1[..., 0, :])
                state_norm = [sn[..., 0], sn[..., 1]]
                disc = _random.ball(s2[..., 1, :].reshape(-1, 2), 2, 2, self.dtype).reshape(tuple(shape) + (2,))
                key_leaf = s2[..., 0, :]
            else:
                gen = rng
                if not isinstance(rng, torch.Generator):
                    gen = torch.Generator(device=self.device)
                    gen.manual_seed(int(rng))
                u = lambda: torch.rand(shape, generator=gen, dtype=self.dtype, device=self.device)
                state_norm = [u() * 2 - 1, u() * 2 - 1]
                r, phi = torch.sqrt(u()), u() * (2 * math.pi)  # uniform in the unit disc (own stream)
                disc = torch.stack([r * torch.cos(phi), r * torch.sin(phi)], dim=-1)
            (d_lo, d_hi), (q_lo, q_hi) = lo_hi("i_d"), lo_hi("i_q")
            i_max = torch.as_tensor(max(abs(float(torch.as_tensor(v).max())) for v in (d_lo, d_hi, q_lo, q_hi)))
            i_d, i_q = disc[..., 0] * i_max, disc[..., 1] * i_max
            relu = torch.nn.functional.relu
            i_d = i_d - 2 * relu(i_d - d_hi) + 2 * relu(-i_d + d_lo)
            i_q = i_q - 2 * relu(i_q - q_hi) + 2 * relu(-i_q + q_lo)
            sp = env_properties.static_params
            leaf = self._norm_leaf
            if getattr(env_properties, "saturated", False):  # currents_to_torque_saturated (pmsm_env.py:377-381)
                q = self._lut_torch(i_d, i_q)
                torque = 1.5 * leaf(sp.p) * (q[..., 4] * i_q - q[..., 5] * i_d)
            else:
                torque = 1.5 * leaf(sp.p) * (leaf(sp.psi_p) + (leaf(sp.l_d) - leaf(sp.l_q)) * i_d) * i_q
            (e_lo, e_hi), (o_lo, o_hi) = lo_hi("epsilon"), lo_hi("omega_el")
            phys = dict(u_d_buffer=full(0.0), u_q_buffer=full(0.0),
                        epsilon=(state_norm[0] + 1) / 2 * (e_hi - e_lo) + e_lo, i_d=i_d, i_q=i_q, torque=torque,
                        omega_el=(state_norm[1] + 1) / 2 * (o_hi - o_lo) + o_lo)
        ref = {n: self._nan(shape) for n in self.STATE_FIELDS}
        prng = self._nan(shape)
        if rng is not None and key_leaf is not None:
            prng = key_leaf
        return self.State(physical_state=self.PhysicalState(**phys), PRNGKey=prng,
                          additions=self._additions(shape, False), reference=self.PhysicalState(**ref))

    def generate_observation(self, system_state, env_properties):
        """pmsm_env.py:898-919: [i_d, i_q, omega_el, torque, cos(eps), sin(eps), u_d_buffer, u_q_buffer] (+ refs)."""
        eps = system_state.physical_state.epsilon
        ns = self.normalize_state(system_state, env_properties)
        p = ns.physical_state
        cols = [p.i_d, p.i_q, p.omega_el, p.torque, torch.cos(eps), torch.sin(eps), p.u_d_buffer, p.u_q_buffer]
        cols += [getattr(ns.reference, n) for n in self.control_state]
        return torch.stack(torch.broadcast_tensors(*cols), dim=-1)

    def _state_from_obs_torch(self, obs, env_properties, key=None):
        """pmsm_env.py:921-970 (elementwise torch mirror; device batches go through the kernel, core_env.py)."""
        shape = tuple(obs.shape[:-1])
        phys = dict(u_d_buffer=obs[..., 6], u_q_buffer=obs[..., 7],
                    epsilon=torch.atan2(obs[..., 5], obs[..., 4]) / math.pi, i_d=obs[..., 0], i_q=obs[..., 1],
                    torque=obs[..., 3], omega_el=obs[..., 2])
        ref = {n: self._nan(shape) for n in self.STATE_FIELDS}
        for pos, n in enumerate(self.control_state):
            ref[n] = obs[..., 8 + pos]
        norm_state = self.State(physical_state=self.PhysicalState(**phys),
                                PRNGKey=self._nan(shape) if key is None else key,
                                additions=self._additions(shape, False), reference=self.PhysicalState(**ref))
        return self.denormalize_state(norm_state, env_properties)

    def generate_truncated(self, system_state, env_properties):
        """pmsm_env.py:972-979: normalised current magnitude > 1."""
        ns = self.normalize_state(system_state, env_properties)
        i_s = torch.sqrt(ns.physical_state.i_d ** 2 + ns.physical_state.i_q ** 2)
        return (i_s > 1)[..., None]

    def generate_terminated(self, system_state, reward, env_properties):
        """pmsm_env.py:981-983."""
        return self.generate_truncated(system_state, env_properties)

    def generate_reward(self, state, action, env_properties):
        """pmsm_env.py:985-1037 (current_reward_func / torque_reward_func with gamma = 0.85)."""
        ns = self.normalize_state(state, env_properties)
        p, r = ns.physical_state, ns.reference
        reward = torch.zeros_like(p.i_d)
        if "i_d" in self.control_state and "i_q" in self.control_state:
            mse = 0.5 * (p.i_d - r.i_d) ** 2 + 0.5 * (p.i_q - r.i_q) ** 2
            reward = reward + -1 * (mse * (1 - 0.85))
        if "torque" in self.control_state:
            i_s = torch.sqrt(p.i_d ** 2 + p.i_q ** 2)
            i_n, i_d_plus, tol = 1.0, 0.2, 0.01
            rew = torch.zeros_like(r.torque)
            rew = torch.where(i_s > 1, -1 * i_s.abs(), rew)
            rew = torch.where((i_s < i_n) & (p.i_d > i_d_plus), -0.5 * ((p.i_d - i_d_plus) / (i_n - i_d_plus)), rew)
            ad = (p.torque - r.torque).abs()
            rew = torch.where((i_s < i_n) & (p.i_d < i_d_plus) & (ad > tol), 0.5 * (1 - ((r.torque - p.torque) / 2).abs()), rew)
            rew = torch.where((i_s < i_n) & (p.i_d < i_d_plus) & (ad < tol), 1 - 0.5 * i_s, rew)
            reward = reward + rew * (1 - 0.85)
        return reward[..., None]

    @property
    def action_description(self):
        return self._action_description

    @property
    def obs_description(self):
        return np.hstack([np.array(self._obs_description), np.array([n + "_ref" for n in self.control_state])])
