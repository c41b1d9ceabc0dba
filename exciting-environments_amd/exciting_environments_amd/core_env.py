"""Core batched runtime — host-side mirror of reference exciting_environments/core_env.py.

Same class, method names, argument meaning, return shapes and error behaviour as the reference's
``CoreEnvironment`` for the batched ODE hot path; the arithmetic of ``vmap_step`` / ``vmap_sim_ahead``
(and their single-env forms) runs in hand-written HIP kernels behind the C ABI of ``include/excenv.h``.
Arrays are ``torch`` tensors on the HIP device (``jax`` is not part of this stack); pytrees are plain
dataclasses (helpers in ``tree.py``).

Differences from the reference that a caller can observe are listed in DESIGN.md ("Deviations").
"""
from __future__ import annotations

import ctypes
import math
import os
import sys
import warnings
from abc import ABC
from dataclasses import dataclass, fields, is_dataclass, replace
from operator import attrgetter
from typing import Any, Optional

import numpy as np
import torch

from . import _native
from . import random as _random
from ._placement import StepSlotPool, TrajectoryPlacement
from ._trajectory import TrajectoryLaunchMixin
from .solvers import Euler, _Solver
from .tree import tree_structure


def _is_array(x) -> bool:
    return isinstance(x, (torch.Tensor, np.ndarray))


def _is_scalar(x) -> bool:
    if isinstance(x, (bool, int, float, np.generic)):
        return True
    return _is_array(x) and x.ndim == 0


# private fast accessors of torch when this build has them, the public (slower) API otherwise
_cuda_get_device = getattr(torch._C, "_cuda_getDevice", None) or torch.cuda.current_device
_cuda_is_capturing = getattr(torch._C, "_cuda_isCurrentStreamCapturing", None) or torch.cuda.is_current_stream_capturing


class CoreEnvironment(TrajectoryLaunchMixin, ABC):
    """Core structure of the provided environments (reference core_env.py:15-57).

    The simulated systems are physical state-space models dx/dt = f(x(t), u(t)); outputs are
    discretised with a fixed-step ODE solver. Sub-classes define the field names of the physical
    state / action / static parameters; the vector field itself lives in the HIP kernels.
    """

    # set by sub-classes
    ENV_ID: int = -1
    STATE_FIELDS: tuple = ()
    ACTION_FIELDS: tuple = ()
    PARAM_FIELDS: tuple = ()
    DEFAULT_NORM_STATE: tuple = ()  # normalised default reset state (rng=None)
    PhysicalState: Any = None
    Action: Any = None
    StaticParams: Any = None
    Additions: Any = None

    @dataclass
    class State:
        """The state of the environment (core_env.py:236-243)."""

        physical_state: Any
        PRNGKey: Any
        additions: Any
        reference: Any

    @dataclass
    class EnvProperties:
        """The properties of the environment that stay constant during simulation (core_env.py:245-251)."""

        physical_normalizations: Any
        action_normalizations: Any
        static_params: Any

    def __init__(self, batch_size: int, env_properties, tau: float = 1e-4, solver=Euler(), dtype=torch.float32,
                 device=None):
        """core_env.py:36-57. ``dtype`` replaces the reference's process-global ``jax_enable_x64`` switch;
        ``device`` defaults to the current HIP device (CPU tensors are only good for construction / reset)."""
        if not isinstance(solver, _Solver):
            raise TypeError(
                f"solver must be one of exciting_environments_amd.Euler()/RK4()/Tsit5(), got {type(solver)}"
            )
        _native.dtype_id(dtype)
        self.batch_size = batch_size
        self.tau = tau
        self._solver = solver
        self.dtype = dtype
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        self.device = torch.device(device)
        if self.device.type == "cuda" and self.device.index is None and torch.cuda.is_available():
            # "cuda" -> "cuda:<current>": tensors report an indexed device, and the fast paths compare devices for equality
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.env_properties = env_properties
        self.in_axes_env_properties = self.create_in_axes_dataclass(env_properties)
        self.action_dim = len(fields(self.Action))
        self.physical_state_dim = len(fields(self.PhysicalState))
        # "lane_major": trajectories are [K+1, O, B] buffers returned as [B, K+1, O] views (coalesced kernel
        # accesses); "env_major": contiguous [B, K+1, O] like the reference's row-major jnp arrays.
        self.traj_layout = "lane_major"
        # "ahead": structure of the reference's _ode_solver_simulate_ahead; "step": K exact `step`s.
        self.sim_ahead_semantics = "ahead"
        # env-major (row-major) buffers, three bit-identical paths: fused LDS time-tile kernel (default when both actions and
        # trajectories are env-major), else transposition through a scratch workspace, else the generic-stride kernel
        self.env_major_fused = True
        self.env_major_workspace = True
        # False: vmap_sim_ahead skips the physical-state trajectories (`states` is None; 40 instead of 68 bytes per
        # PMSM env-step). The reference always returns them, so the default is True.
        self.store_state_trajectory = True
        # per-call launch options (excenv_launch_opts_t); None = library defaults. Tuning experiments only.
        self.launch_opts = None
        self._packed_props = None
        self._packed_for = None
        self._flag_cache = {}
        self._obs_dim_cache = None
        # Output memory of the hot path (_placement.py): vmap_step's recycled output slots; the pooled, placed output sets of large
        # vmap_sim_ahead calls. Both are invisible to callers (a buffer is handed out again only when nothing can observe it).
        self._step_pool = StepSlotPool(self)
        self.trajectory_pool = True          # False: every large call allocates its outputs (the behaviour up to round 2)
        self.trajectory_placement = "auto"   # "auto": ordered pair, search as fall-back; "search": the search only; "off": none
        self._placement = TrajectoryPlacement(self)
        self._traj_bcast_cache = None
        self._ws_bytes_cache = None
        self._fused_actions_cache = None
        self._last_out = None
        self._ctl_cache = None
        self._active_additions = None
        self._leaf_getter = None

    # ------------------------------------------------------------------ properties plumbing
    def create_in_axes_dataclass(self, dataclass_obj):
        """core_env.py:253-277: 0 for leaves batched over batch_size, None for broadcast leaves."""
        out = {}
        for f in fields(dataclass_obj):
            name = f.name
            value = getattr(dataclass_obj, name)
            if value is None:
                out[name] = None
            elif isinstance(value, list):
                raise ValueError(
                    f'Passed env property "{name}" needs to be a jnp.array to have different setting per batch, but list is given.'
                )
            elif is_dataclass(value):
                out[name] = self.create_in_axes_dataclass(value)
            elif _is_scalar(value):
                out[name] = None
            elif _is_array(value):
                out[name] = 0 if value.shape[0] == self.batch_size else None
            else:
                raise ValueError(
                    f'Passed env property "{name}" needs to be a scalar, jnp.array or jdc.pytree_dataclass, but {type(value)} is given.'
                )
        return replace(dataclass_obj, **out)

    def _leaf(self, x):
        """Property leaf -> Python float (broadcast) or [B] tensor of the working dtype on the device."""
        if isinstance(x, bool):
            return x
        if _is_array(x) and x.ndim >= 1:
            t = torch.as_tensor(x).to(device=self.device, dtype=self.dtype)
            return t
        if isinstance(x, torch.Tensor):
            return float(x.item())
        return float(x)

    def _pack_props(self, env_properties, B: int):
        """EnvProperties -> excenv_props_t (+ the device tensors it points into, kept alive by the caller)."""
        keep = []
        p = _native.Props()

        def put(param, value, what):
            v = self._leaf(value)
            if isinstance(v, torch.Tensor):
                if v.shape[0] != B or v.ndim != 1:
                    if v.numel() == 1:
                        param.value, param.per_env = float(v.reshape(()).item()), None
                        return
                    raise ValueError(f"env property {what} has shape {tuple(v.shape)}; expected a scalar or ({B},)")
                v = v.contiguous()
                keep.append(v)
                param.value, param.per_env = float("nan"), v.data_ptr()
            else:
                param.value, param.per_env = v, None

        for j, n in enumerate(self.PARAM_FIELDS):
            put(p.static_params[j], getattr(env_properties.static_params, n), f"static_params.{n}")
        for j, n in enumerate(self.STATE_FIELDS):
            nm = getattr(env_properties.physical_normalizations, n)
            put(p.state_min[j], nm.min, f"physical_normalizations.{n}.min")
            put(p.state_max[j], nm.max, f"physical_normalizations.{n}.max")
        for j, n in enumerate(self.ACTION_FIELDS):
            nm = getattr(env_properties.action_normalizations, n)
            put(p.action_min[j], nm.min, f"action_normalizations.{n}.min")
            put(p.action_max[j], nm.max, f"action_normalizations.{n}.max")
        return p, keep

    def _props_for(self, env_properties, B: int):
        if env_properties is self.env_properties and B == self.batch_size:
            if self._packed_props is None or self._packed_for is not env_properties:
                self._packed_props = self._pack_props(env_properties, B)
                self._packed_for = env_properties
            return self._packed_props
        return self._pack_props(env_properties, B)

    def _t(self, x, shape=None):
        if (isinstance(x, torch.Tensor) and x.dtype == self.dtype and x.device == self.device and x.is_contiguous()
                and (shape is None or tuple(x.shape) == tuple(shape))):
            return x  # fast path: already a device tensor of the working dtype
        t = torch.as_tensor(x).to(device=self.device, dtype=self.dtype)
        if shape is not None and tuple(t.shape) != tuple(shape):
            t = t.expand(shape)
        return t.contiguous()

    def _norm_leaf(self, x):
        return self._leaf(x)

    # ------------------------------------------------------------------ normalisation (torch, elementwise)
    def normalize_state(self, state, env_properties):
        """core_env.py:292-314."""
        pn = env_properties.physical_normalizations
        phys, ref = {}, {}
        for n in self.STATE_FIELDS:
            nm = getattr(pn, n)
            lo, hi = self._norm_leaf(nm.min), self._norm_leaf(nm.max)
            phys[n] = 2 * (getattr(state.physical_state, n) - lo) / (hi - lo) - 1
            ref[n] = 2 * (getattr(state.reference, n) - lo) / (hi - lo) - 1
        return replace(state, physical_state=self.PhysicalState(**phys), reference=self.PhysicalState(**ref))

    def denormalize_state(self, norm_state, env_properties):
        """core_env.py:316-340."""
        pn = env_properties.physical_normalizations
        phys, ref = {}, {}
        for n in self.STATE_FIELDS:
            nm = getattr(pn, n)
            lo, hi = self._norm_leaf(nm.min), self._norm_leaf(nm.max)
            phys[n] = (getattr(norm_state.physical_state, n) + 1) / 2 * (hi - lo) + lo
            ref[n] = (getattr(norm_state.reference, n) + 1) / 2 * (hi - lo) + lo
        return replace(norm_state, physical_state=self.PhysicalState(**phys), reference=self.PhysicalState(**ref))

    def denormalize_action(self, action_norm, env_properties):
        """core_env.py:342-359 (last axis = action components)."""
        an = env_properties.action_normalizations
        cols = []
        for i, n in enumerate(self.ACTION_FIELDS):
            nm = getattr(an, n)
            lo, hi = self._norm_leaf(nm.min), self._norm_leaf(nm.max)
            cols.append((action_norm[..., i] + 1) / 2 * (hi - lo) + lo)
        return torch.stack(cols, dim=-1)

    # ------------------------------------------------------------------ state construction
    def _nan(self, shape):
        return torch.full(shape, float("nan"), dtype=self.dtype, device=self.device)

    def _additions(self, shape, active: bool):
        # constant flag leaves are shared between states (states are immutable by contract, like the reference's
        # pytrees), so stepping does not launch a fill kernel per call
        key = (tuple(shape), bool(active))
        flag = self._flag_cache.get(key)
        if flag is None:
            flag = torch.full(shape, active, dtype=torch.bool, device=self.device)
            if len(self._flag_cache) < 64:
                self._flag_cache[key] = flag
        return self.Additions(solver_state=self._solver_state_leaf(shape), active_solver_state=flag)

    # number of ODE variables the reference hands to diffrax (y0 tuples, e.g. pendulum_env.py:175; PMSM integrates
    # (i_d, i_q, eps) only, pmsm_env.py:555): the arity of the FSAL derivative in the solver state
    N_ODE: Optional[int] = None

    def _solver_state_leaf(self, shape):
        """Additions.solver_state with the reference's pytree structure (e.g. pendulum_env.py:177-192, 249-251, 289-290):
        None for Euler (diffrax.Euler has no solver state) and the RK4 extension; for Tsit5 the pair diffrax's FSAL Runge-Kutta
        keeps, (first_step, f0) with f0 a tuple shaped like the ODE state — filled with NaN in EVERY state. The reference
        holds NaN there after a reset too (`tree_map(lambda x: x * jnp.nan, solver_state)`) and diffrax-internal values after
        a step, which nothing but diffrax's own step reads; the fixed-step kernels need none of it (parity of the structure,
        not of those values; diffrax's source is not in the container)."""
        if not getattr(self._solver, "fsal", False):
            return None
        key = ("solver_state", tuple(shape))
        leaf = self._flag_cache.get(key)
        if leaf is None:
            leaf = self._nan(tuple(shape))
            if len(self._flag_cache) < 64:
                self._flag_cache[key] = leaf
        n = self.N_ODE if self.N_ODE is not None else len(self.STATE_FIELDS)
        return (leaf, tuple(leaf for _ in range(n)))

    def _random_norm_state(self, rng, shape):
        """Random normalised initial state (e.g. pendulum_env.py:270-276). `rng` is either a key tensor ([2] / [B, 2]
        uint32 words, `exciting_environments_amd.random.PRNGKey / split`): then the draw restates
        jax.random.uniform(key, shape=(S,), minval=-1, maxval=1) — or a torch.Generator / int seed (own stream)."""
        if _random.is_key(rng):
            key = rng.to(self.device)
            assert tuple(key.shape[:-1]) == tuple(shape), f"rng keys must have shape {tuple(shape) + (2,)}"
            lo = 0.0 if self.ENV_ID == 4 else -1.0
            u = _random.uniform(key, len(self.STATE_FIELDS), self.dtype, lo, 1.0)
            return {n: u[..., j].contiguous() for j, n in enumerate(self.STATE_FIELDS)}
        gen = rng
        if not isinstance(rng, torch.Generator):
            gen = torch.Generator(device=self.device)
            gen.manual_seed(int(rng))
        lo = 0.0 if self.ENV_ID == 4 else -1.0  # FluidTank draws the height from [0, 1) (fluid_tank_env.py:226)
        return {
            n: (torch.rand(shape, generator=gen, dtype=self.dtype, device=self.device) * (1.0 - lo) + lo)
            for n in self.STATE_FIELDS
        }

    def _init_state_device_keys(self, env_properties, rng, shape):
        """One launch (excenv_random_state) for a [B, 2] batch of keys on the HIP device; None when this path does not apply
        (CPU, single key, property arrays of another batch size) and the torch twin has to do it."""
        if not (_random.is_key(rng) and len(shape) == 1 and rng.ndim == 2 and self.device.type == "cuda"):
            return None
        B = shape[0]
        assert tuple(rng.shape) == (B, 2), f"rng keys must have shape {(B, 2)}"
        try:
            props, _keep = self._props_for(env_properties, B)
        except ValueError:
            return None
        S = self.physical_state_dim
        al = 16 // (4 if self.dtype == torch.float32 else 8)
        Bp = (B + al - 1) // al * al
        buf = torch.empty((S, Bp), dtype=self.dtype, device=self.device)
        leaves = [buf[j, :B] for j in range(S)]
        keys = rng.to(device=self.device, dtype=torch.int64).contiguous()
        leaf = torch.empty_like(keys)
        _native.random_state(self.ENV_ID, self.dtype, B, props, keys, leaves, leaf)
        ref = {n: self._nan(shape) for n in self.STATE_FIELDS}
        return self.State(physical_state=self.PhysicalState(*leaves), PRNGKey=leaf, additions=self._additions(shape, False),
                          reference=self.PhysicalState(**ref))

    def _init_state(self, env_properties, rng, shape):
        dev = self._init_state_device_keys(env_properties, rng, shape)
        if dev is not None:
            return dev
        if rng is None:
            norm = {n: torch.full(shape, v, dtype=self.dtype, device=self.device)
                    for n, v in zip(self.STATE_FIELDS, self.DEFAULT_NORM_STATE)}
        else:
            norm = self._random_norm_state(rng, shape)
        ref = {n: self._nan(shape) for n in self.STATE_FIELDS}
        # the state's PRNGKey leaf: jax.random.split(rng)[1] for key input (pendulum_env.py:276), NaN otherwise
        key_leaf = _random.split(rng.to(self.device))[..., 1, :] if _random.is_key(rng) else self._nan(shape)
        norm_state = self.State(physical_state=self.PhysicalState(**norm), PRNGKey=key_leaf,
                                additions=self._additions(shape, False), reference=self.PhysicalState(**ref))
        return self.denormalize_state(norm_state, env_properties)

    def init_state(self, env_properties, rng=None, vmap_helper=None):
        """Default (rng=None) or random initial state for one environment (e.g. pendulum_env.py:261-295)."""
        return self._init_state(env_properties, rng, ())

    def vmap_init_state(self, rng=None):
        """core_env.py:649-662."""
        return self._init_state(self.env_properties, rng, (self.batch_size,))

    def _observe_device(self, state, env_properties):
        if self.device.type != "cuda":
            return None
        leaves = [getattr(state.physical_state, n) for n in self.STATE_FIELDS]
        l0 = leaves[0]
        if not (isinstance(l0, torch.Tensor) and l0.is_cuda and l0.ndim == 1 and l0.shape[0] > 0):
            return None
        B = l0.shape[0]
        if not all(isinstance(l, torch.Tensor) and l.ndim == 1 and l.shape[0] == B for l in leaves):
            return None
        try:
            props, _keep = self._props_for(env_properties, B)
        except (ValueError, AssertionError, RuntimeError):
            return None  # properties shaped for something else than a [B] batch (e.g. lifted for trajectories)
        st = [self._t(l, (B,)) for l in leaves]
        control = None
        if self.control_state:
            refs = [self._t(getattr(state.reference, n), (B,)) for n in self.control_state]
            control = _native.make_control([self.STATE_FIELDS.index(n) for n in self.control_state], refs)
        obs = torch.empty((B, self._obs_dim()), dtype=self.dtype, device=self.device)
        _native.observe(self.ENV_ID, self.dtype, B, props, control, st, obs)
        return obs

    def generate_observation(self, state, env_properties):
        """Normalised physical state (+ normalised reference for each name in control_state), stacked on the last
        axis (e.g. pendulum_env.py:311-329). A batch of states on the HIP device ([B] leaves) is one launch (excenv_observe, the
        device function the step kernels fuse); anything else (a single environment, trajectories, CPU tensors) goes through the
        elementwise torch twin below."""
        dev = self._observe_device(state, env_properties)
        if dev is not None:
            return dev
        ns = self.normalize_state(state, env_properties)
        cols = [getattr(ns.physical_state, n) for n in self.STATE_FIELDS]
        cols += [getattr(ns.reference, n) for n in self.control_state]
        return torch.stack(torch.broadcast_tensors(*cols), dim=-1)

    def generate_state_from_observation(self, obs, env_properties, key=None):
        """e.g. pendulum_env.py:331-364. A [B', O] batch on the HIP device goes through one kernel
        (excenv_state_from_observation); other shapes (a single observation, extra leading axes) and CPU tensors use the
        elementwise torch mirror."""
        obs = self._t(obs)
        if obs.ndim == 2 and obs.is_cuda and obs.shape[1] == self._obs_dim():
            try:
                props, keep = self._props_for(env_properties, obs.shape[0])
            except ValueError:
                props = None  # per-env property arrays of another batch size: broadcast semantics of the torch mirror
            if props is not None:
                return self._state_from_obs_device(obs, props, key)
        return self._state_from_obs_torch(obs, env_properties, key)

    def _state_from_obs_device(self, obs, props, key):
        Bq, S = obs.shape[0], self.physical_state_dim
        idx = [self.STATE_FIELDS.index(n) for n in self.control_state]
        isz = obs.element_size()
        al = 16 // isz
        Bp = (Bq + al - 1) // al * al
        buf = torch.empty((S + len(idx), Bp), dtype=self.dtype, device=self.device)
        leaves = [buf[j, :Bq] for j in range(S)]
        refs = [buf[S + j, :Bq] for j in range(len(idx))]
        _native.state_from_observation(self.ENV_ID, self.dtype, Bq, props, idx, obs, leaves, refs)
        ref = {n: self._nan((Bq,)) for n in self.STATE_FIELDS}
        for n, r in zip(self.control_state, refs):
            ref[n] = r
        return self.State(physical_state=self.PhysicalState(*leaves), PRNGKey=self._nan((Bq,)) if key is None else key,
                          additions=self._additions((Bq,), False), reference=self.PhysicalState(**ref))

    def _state_from_obs_torch(self, obs, env_properties, key=None):
        shape = tuple(obs.shape[:-1])
        S = len(self.STATE_FIELDS)
        phys = {n: obs[..., j] for j, n in enumerate(self.STATE_FIELDS)}
        ref = {n: self._nan(shape) for n in self.STATE_FIELDS}
        for pos, n in enumerate(self.control_state):
            ref[n] = obs[..., S + pos]
        norm_state = self.State(physical_state=self.PhysicalState(**phys),
                                PRNGKey=self._nan(shape) if key is None else key,
                                additions=self._additions(shape, False), reference=self.PhysicalState(**ref))
        return self.denormalize_state(norm_state, env_properties)

    def vmap_generate_state_from_observation(self, obs, key=None):
        """core_env.py:689-705."""
        return self.generate_state_from_observation(obs, self.env_properties, key)

    # ------------------------------------------------------------------ reset
    def reset(self, env_properties, rng=None, initial_state=None, vmap_helper=None):
        """core_env.py:361-391."""
        if initial_state is not None:
            assert tree_structure(self.init_state(env_properties)) == tree_structure(
                initial_state
            ), "initial_state should have the same dataclass structure as init_state()"
            state = initial_state
        else:
            state = self.init_state(env_properties, rng)
        obs = self.generate_observation(state, env_properties)
        return obs, state

    def vmap_reset(self, rng=None, initial_state=None):
        """core_env.py:664-687."""
        if initial_state is not None:
            assert tree_structure(self.vmap_init_state()) == tree_structure(
                initial_state
            ), "initial_state should have the same dataclass structure as self.vmap_init_state()"
            state = initial_state
        else:
            state = self.vmap_init_state(rng)
        obs = self.generate_observation(state, self.env_properties)
        return obs, state

    # ------------------------------------------------------------------ the hot path
    def _control(self, state, shape):
        if not self.control_state:
            return None, []
        idx = [self.STATE_FIELDS.index(n) for n in self.control_state]
        refs = [self._t(getattr(state.reference, n), shape) for n in self.control_state]
        return _native.make_control(idx, refs), refs

    def _run_step(self, state, action, env_properties, B):
        S, O = self.physical_state_dim, self._obs_dim()
        props, keep = self._props_for(env_properties, B)
        st_in = [self._t(getattr(state.physical_state, n), (B,)) for n in self.STATE_FIELDS]
        act = self._t(action, (B, self.action_dim))
        if act.data_ptr() % 16:  # a row sliced out of a larger array (actions[b, k]): the kernels read action rows as 16-byte pieces
            act = act.clone()
        control, refs = self._control(state, (B,))
        pad = (-B) % 4  # keep every leaf 16-byte aligned inside the single allocation
        buf = torch.empty(S * (B + pad) + B * O, dtype=self.dtype, device=self.device)
        st_out = [buf[j * (B + pad): j * (B + pad) + B] for j in range(S)]
        obs = buf[S * (B + pad):].view(B, O)
        _native.step(self.ENV_ID, self._solver.id, self.dtype, B, props, control, float(self.tau), st_in, act,
                     st_out, obs, self.launch_opts)
        return obs, st_out

    def _obs_dim(self):
        if self._obs_dim_cache is None or self._obs_dim_cache[0] != len(self.control_state):
            self._obs_dim_cache = (len(self.control_state), _native.env_dims(self.ENV_ID)[2] + len(self.control_state))
        return self._obs_dim_cache[1]

    def step(self, state, action_norm, env_properties):
        """One simulation step of a single environment (core_env.py:393-425)."""
        action_norm = torch.as_tensor(action_norm)
        assert tuple(action_norm.shape) == (self.action_dim,), (
            "The action needs to be of shape (action_dim,) which is "
            + f"{(self.action_dim,)}, but {tuple(action_norm.shape)} is given"
        )
        physical_state_shape = self._phys_shape(state.physical_state)
        assert physical_state_shape == (self.physical_state_dim,), (
            "The physical state needs to be of shape (physical_state_dim,) which is "
            + f"{(self.physical_state_dim,)}, but {physical_state_shape} is given"
        )
        obs, st_out = self._run_step(state, action_norm.reshape(1, -1), env_properties, 1)
        new_phys = self.PhysicalState(**{n: t.reshape(()) for n, t in zip(self.STATE_FIELDS, st_out)})
        new_state = replace(state, physical_state=new_phys, additions=self._additions((), True))
        return obs[0], new_state

    # -- vmap_step fast path -------------------------------------------------------------------------------------------
    # Per call: identity check of the incoming state leaves (did we return them last time? then their pointer array is already
    # built), one data_ptr() for the action, one ctypes call with pre-built arguments, two small dataclass constructions. The
    # output slot comes from _placement.StepSlotPool (fresh memory per call; a dead slot's tensor objects are used again).
    def _vmap_step_launch(self, state, action, gym: bool, obs_refs=None):
        B = self.batch_size
        dev = self.device
        getter = self._leaf_getter
        if getter is None:
            names = self.STATE_FIELDS
            getter = self._leaf_getter = attrgetter(*names) if len(names) > 1 else (lambda ps, _g=attrgetter(names[0]): (_g(ps),))
        leaves = getter(state.physical_state)
        last = self._last_out
        if last is not None and last[0] == tuple(map(id, leaves)):
            in_ptrs = last[2]  # the state we returned last: its output pointer array is this call's input
        else:
            shape = tuple(torch.as_tensor(leaves[0]).shape) + (len(leaves),)
            assert shape == (B, self.physical_state_dim), (
                "The physical state needs to be of shape (batch_size, physical_state_dim) which is "
                + f"{(B, self.physical_state_dim)}, but {shape} is given"
            )
            st_in = [self._t(l, (B,)) for l in leaves]
            in_ptrs = _native._ptrs(st_in)
            self._last_out = None
        if action.dtype is not self.dtype or action.device != dev or not action.is_contiguous():
            action = self._t(action, (B, self.action_dim))
        if action.data_ptr() % 16:  # a contiguous slice that starts inside a 16-byte piece
            action = action.clone()
        _native._require_device(action, "vmap_step")
        control_ref = None
        if self.control_state:
            refs = tuple(getattr(state.reference, n) for n in self.control_state)
            cc = self._ctl_cache
            key = (tuple(self.control_state), tuple(map(id, refs)), None if obs_refs is None else tuple(map(id, obs_refs)))
            if cc is None or cc[0] != key:
                tens = [self._t(r, (B,)) for r in refs]
                otens = None if obs_refs is None else [self._t(r, (B,)) for r in obs_refs]  # gym_step: obs shows these
                ctl = _native.make_control([self.STATE_FIELDS.index(n) for n in self.control_state], tens, otens)
                cc = (key, (refs, obs_refs), (tens, otens), ctl, ctypes.byref(ctl))
                # The struct holds the pointers of `tens`. Keep it across calls only when those ARE the caller's leaves:
                # a converted copy (CPU / other dtype / non-contiguous leaf) would go stale when the caller updates the
                # original in place (same id), so such leaves are converted again on every call.
                same = all(t is r for t, r in zip(tens, refs)) and (
                    otens is None or all(t is r for t, r in zip(otens, obs_refs)))
                self._ctl_cache = cc if same else None
            control_ref = cc[4]
        props, _keep = self._props_for(self.env_properties, B)
        idx = dev.index
        cur = _cuda_get_device()
        if idx is None:
            idx = cur
        if idx == cur:
            stream = _native.raw_stream(idx)
            capturing = _cuda_is_capturing()
        else:
            with torch.cuda.device(dev):
                stream = _native.raw_stream(idx)
                capturing = _cuda_is_capturing()
        oc = self._obs_dim_cache
        sl, i = self._step_pool.take(gym, stream, capturing, oc[1] if oc and oc[0] == len(self.control_state) else self._obs_dim())
        if self._placement.wait_events:  # pool_wait_stream(): a foreign stream still reads what an earlier call returned
            self._placement.drain_waits()
        opts = self.launch_opts
        args = (self.ENV_ID, self._solver.id, 0 if self.dtype is torch.float32 else 1, B, ctypes.byref(props), control_ref,
                self.tau, in_ptrs, action.data_ptr(), sl.out_ptrs[i], sl.obs_ptrs[i],
                None if opts is None else ctypes.byref(opts), stream, sl.gym_ptrs[i] if gym else None)
        if idx == cur:
            _native.step_raw(*args)
        else:
            with torch.cuda.device(dev):
                _native.step_raw(*args)
        new_leaves = sl.leaves[i]
        if not capturing:
            self._last_out = (tuple(map(id, new_leaves)), new_leaves, sl.out_ptrs[i])
        add = self._active_additions
        if add is None:
            add = self._active_additions = self._additions((B,), True)
        new_state = self.State(sl.phys[i], state.PRNGKey, add, state.reference)
        if gym:
            rew, term, trunc = sl.gym[i]
            return sl.obs[i], rew, term, trunc, new_state
        return sl.obs[i], new_state

    def vmap_step(self, state, action):
        """One simulation step of all batch_size environments (core_env.py:533-569): one fused HIP launch."""
        if type(action) is not torch.Tensor:
            action = torch.as_tensor(action)
        assert action.shape == (self.batch_size, self.action_dim), (
            "The action needs to be of shape (batch_size, action_dim) which is "
            + f"{(self.batch_size, self.action_dim)}, but {tuple(action.shape)} is given"
        )
        return self._vmap_step_launch(state, action, False)

    def vmap_gym_step(self, state, action):
        """vmap_step fused with generate_reward / generate_terminated / generate_truncated in ONE launch — what
        GymWrapper.gym_step computes per step (gym_wrapper.py:88-130). Returns
        (obs [B,O], reward [B,1], terminated [B,1] bool, truncated [B,TW] bool, new_state)."""
        if type(action) is not torch.Tensor:
            action = torch.as_tensor(action)
        assert action.shape == (self.batch_size, self.action_dim), (
            "The action needs to be of shape (batch_size, action_dim) which is "
            + f"{(self.batch_size, self.action_dim)}, but {tuple(action.shape)} is given"
        )
        return self._vmap_step_launch(state, action, True)

    # ------------------------------------------------------------------ reward / truncated / terminated (torch mirrors)
    ANGLE_FIELDS: tuple = ()

    def generate_reward(self, state, action, env_properties):
        """e.g. pendulum_env.py:297-309 (angles through sin/cos), mass_spring_damper_env.py:296-302. Trailing axis 1."""
        ns = self.normalize_state(state, env_properties)
        first = getattr(state.physical_state, self.STATE_FIELDS[0])
        reward = torch.zeros_like(torch.as_tensor(first, dtype=self.dtype, device=self.device))
        for name in self.control_state:
            if name in self.ANGLE_FIELDS:
                th, ref = getattr(state.physical_state, name), getattr(state.reference, name)
                reward = reward + -((torch.sin(th) - torch.sin(ref)) ** 2 + (torch.cos(th) - torch.cos(ref)) ** 2)
            else:
                reward = reward + -((getattr(ns.physical_state, name) - getattr(ns.reference, name)) ** 2)
        return reward[..., None]

    def generate_truncated(self, state, env_properties):
        """|obs| > 1 per observation column (e.g. pendulum_env.py:381-385)."""
        return self.generate_observation(state, env_properties).abs() > 1

    def generate_terminated(self, state, reward, env_properties):
        """reward == 0 (e.g. pendulum_env.py:387-390)."""
        return reward == 0

    def repeat_values(self, x, n_repeat):
        """core_env.py:279-290: repeats the values of x n_repeat times (None / tuple / tensor / float / bool)."""
        if x is None:
            return None
        if isinstance(x, tuple):
            return tuple(self.repeat_values(i, n_repeat) for i in x)
        if isinstance(x, torch.Tensor):
            return x.expand((n_repeat,) + tuple(x.shape)).clone() if x.ndim else torch.full((n_repeat,), x.item(), dtype=x.dtype, device=x.device)
        if isinstance(x, (float, bool)):
            return torch.full((n_repeat,), x, dtype=torch.bool if isinstance(x, bool) else self.dtype, device=self.device)
        raise ValueError(f"State needs to consist of jnp.array, tuple, float or bool, but {type(x)} is given.")

    def generate_rew_trunc_term_ahead(self, states, actions, env_properties):
        """core_env.py:490-531 for the trajectory of ONE environment returned by `sim_ahead`: reward [n,1] on rows 1..,
        truncated [n+1,TW] on all rows, terminated [n,1] on rows 1.. (elementwise, the same per-row functions)."""
        actions = torch.as_tensor(actions)
        assert actions.ndim == 2, "The actions need to have two dimensions: (n_action_steps, action_dim)"
        assert (
            actions.shape[-1] == self.action_dim
        ), f"The last dimension does not correspond to the action dim which is {self.action_dim}, but {actions.shape[-1]} is given"
        cut = lambda tree: replace(tree, physical_state=self.PhysicalState(**{n: getattr(tree.physical_state, n)[1:] for n in self.STATE_FIELDS}),
                                   reference=self.PhysicalState(**{n: getattr(tree.reference, n)[1:] for n in self.STATE_FIELDS}))
        tail = cut(states)
        reward = self.generate_reward(tail, None, env_properties)
        truncated = self.generate_truncated(states, env_properties)
        terminated = self.generate_terminated(tail, reward, env_properties)
        return reward, truncated, terminated

    def vmap_generate_rew_trunc_term_ahead(self, states, actions):
        """core_env.py:618-647 / :490-531 for trajectories returned by vmap_sim_ahead: reward [B,K,1] on rows 1..,
        truncated [B,K+1,TW] on all rows, terminated [B,K,1] on rows 1... One HIP launch over the stored trajectory
        (excenv_rew_trunc_term; `vmap_sim_ahead(..., return_rew_trunc_term=True)` produces the same values inside the
        trajectory launch itself). CPU tensors fall back to the elementwise torch mirror."""
        actions = torch.as_tensor(actions)
        assert actions.ndim == 3, "The actions need to have three dimensions: (batch_size, n_action_steps, action_dim)"
        assert (
            actions.shape[0] == self.batch_size
        ), f"The first dimension does not correspond to the batch size which is {self.batch_size}, but {actions.shape[0]} is given"
        assert (
            actions.shape[-1] == self.action_dim
        ), f"The last dimension does not correspond to the action dim which is {self.action_dim}, but {actions.shape[-1]} is given"
        leaves = [torch.as_tensor(getattr(states.physical_state, n)) for n in self.STATE_FIELDS]
        if leaves[0].is_cuda and leaves[0].ndim == 2 and leaves[0].shape[0] == self.batch_size:
            return self._rew_trunc_term_device(states, leaves)
        return self._rew_trunc_term_torch(states)

    def _rew_trunc_term_device(self, states, leaves):
        B, rows = leaves[0].shape
        N = rows - 1
        leaves = [l if (l.dtype == self.dtype and l.device == self.device) else l.to(device=self.device, dtype=self.dtype)
                  for l in leaves]
        strides = {tuple(l.stride()) for l in leaves}
        if len(strides) != 1 or any(l.shape != (B, rows) for l in leaves):
            leaves = [l.contiguous() for l in leaves]
        s_sb, s_sk = leaves[0].stride()
        lane_major = (s_sb == 1 and rows > 1)
        props, keep = self._props_for(self.env_properties, B)
        control, ref_strides, refs = None, None, []
        if self.control_state:
            idx = [self.STATE_FIELDS.index(n) for n in self.control_state]
            ref_strides = []
            for n in self.control_state:
                r = torch.as_tensor(getattr(states.reference, n))
                if not (r.is_cuda and r.dtype == self.dtype):
                    r = r.to(device=self.device, dtype=self.dtype)
                if r.ndim != 2 or tuple(r.shape) != (B, rows):
                    r = r.reshape(B, -1).expand(B, rows) if r.ndim >= 1 else r.expand(B, rows)
                refs.append(r)
                ref_strides += list(r.stride())
            control = _native.make_control(idx, refs)
        TW = _native.truncated_width(self.ENV_ID, len(self.control_state))
        if lane_major:  # outputs as [B, ., .] views over lane-major memory, like the trajectories themselves
            rew_buf = torch.empty((max(N, 0), B), dtype=self.dtype, device=self.device)
            term_buf = torch.empty((max(N, 0), B), dtype=torch.bool, device=self.device)
            trunc_buf = torch.empty((rows, B, TW), dtype=torch.bool, device=self.device)
            out = (rew_buf.t()[..., None], trunc_buf.permute(1, 0, 2), term_buf.t()[..., None])
            layout = _native.LAYOUT_LANE_MAJOR
        else:
            rew_buf = torch.empty((B, max(N, 0), 1), dtype=self.dtype, device=self.device)
            term_buf = torch.empty((B, max(N, 0), 1), dtype=torch.bool, device=self.device)
            trunc_buf = torch.empty((B, rows, TW), dtype=torch.bool, device=self.device)
            out = (rew_buf, trunc_buf, term_buf)
            layout = _native.LAYOUT_ENV_MAJOR
        _native.rew_trunc_term(self.ENV_ID, self.dtype, B, rows, props, control, ref_strides, leaves, s_sb, s_sk, rew_buf,
                               term_buf, trunc_buf, layout)
        return out

    def _rew_trunc_term_torch(self, states):
        props = self._traj_properties()
        cut = lambda tree: replace(tree, physical_state=self.PhysicalState(**{n: getattr(tree.physical_state, n)[:, 1:] for n in self.STATE_FIELDS}),
                                   reference=self.PhysicalState(**{n: getattr(tree.reference, n)[:, 1:] for n in self.STATE_FIELDS}))
        tail = cut(states)
        reward = self.generate_reward(tail, None, props)
        truncated = self.generate_truncated(states, props)
        terminated = self.generate_terminated(tail, reward, props)
        return reward, truncated, terminated

    def _traj_properties(self):
        """env_properties whose [B] leaves are reshaped to [B, 1] so they broadcast along the trajectory axis."""
        from .tree import tree_map

        def lift(x):
            if _is_array(x) and x.ndim == 1 and x.shape[0] == self.batch_size:
                return torch.as_tensor(x).to(device=self.device, dtype=self.dtype)[:, None]
            return x

        return tree_map(lift, self.env_properties)

    def _phys_shape(self, physical_state):
        leaves = [torch.as_tensor(getattr(physical_state, n)) for n in self.STATE_FIELDS]
        return tuple(leaves[0].shape) + (len(leaves),)

    @staticmethod
    def _n_substeps(K, obs_stepsize, action_stepsize):
        """Number of solver steps per action. The reference saves 1 + int(t1 / obs_stepsize) rows with
        t1 = action_stepsize * K evaluated in Python doubles (pendulum_env.py:222-225)."""
        ratio = action_stepsize / obs_stepsize
        sub = int(round(ratio))
        if sub < 1 or abs(ratio - sub) > 1e-9 * max(1.0, ratio):
            raise ValueError("action_stepsize must be an integer multiple of obs_stepsize "
                             f"(got {action_stepsize} / {obs_stepsize})")
        n_ref = 1 + int((action_stepsize * K) / obs_stepsize)
        if n_ref != K * sub + 1:
            warnings.warn(
                f"the reference's 1 + int(t1/obs_stepsize) evaluates to {n_ref} rows for these step sizes "
                f"(floating-point floor); returning the intended {K * sub + 1} rows", RuntimeWarning)
        return sub

    # -- large trajectory outputs: pooled and placed by _placement.TrajectoryPlacement -----------------------------------------------
    @classmethod
    def placement_memory_budget(cls, B: int, rows: int, OW: int, S: int, itemsize: int, free_bytes: int) -> dict:
        """Upper bounds (bytes) of what the pooled, placed output sets of one shape can hold (TrajectoryPlacement.memory_budget)."""
        return TrajectoryPlacement.memory_budget(B, rows, OW, S, itemsize, free_bytes)

    def release_trajectory_buffers(self):
        """Drop the pooled (dead) trajectory output sets so that their memory returns to torch's allocator."""
        self._placement.release()

    @property
    def trajectory_placement_settled(self) -> bool:
        """True once no later call of the shapes seen so far can run a placement search: a caller that wants steady timings
        (bench.py) steps until then. Always True when nothing is pooled or placed."""
        return self._placement.settled

    @property
    def last_placement(self):
        """Diagnostics of the most recent placement decision (dict) or None."""
        return self._placement.last

    def pool_wait_stream(self, stream=None):
        """Tell the output pools that `stream` (default: the current stream) is still reading tensors an earlier call returned:
        the next call that hands a pooled buffer out again first makes its launch stream wait for everything queued on `stream`
        up to now. The pools see Python references, C++ holders and views — not `Tensor.record_stream`; a consumer on a side
        stream that drops its reference early calls this (or keeps the reference until it has synchronised, or switches the pools
        off: `trajectory_pool = False`)."""
        self._placement.wait_stream(stream)

    def _traj_state(self, init_state, st_views, lead_shape, N):
        """Rebuild the State pytree of a trajectory: reference / PRNGKey broadcast along the saved rows,
        active_solver_state all True (e.g. pendulum_env.py:243-259)."""
        shape = lead_shape + (N + 1,)
        phys = self.PhysicalState(*st_views)
        # the broadcast reference / key leaves depend only on the incoming leaves and the row count: keep the views of the last
        # call (the cache holds the source leaves, so their ids cannot be reused while it is valid)
        src = tuple(getattr(init_state.reference, n) for n in self.STATE_FIELDS) + (init_state.PRNGKey,)
        ck = self._traj_bcast_cache
        if ck is not None and ck[0] == shape and len(ck[1]) == len(src) and all(a is b for a, b in zip(ck[1], src)):
            ref, key = ck[2], ck[3]
        else:
            conv = [self._t(r) for r in src[:-1]]
            ref = self.PhysicalState(*[t.reshape(lead_shape + (1,)).expand(shape) for t in conv])
            if _random.is_key(init_state.PRNGKey):
                key = init_state.PRNGKey.reshape(lead_shape + (1, 2)).expand(shape + (2,))
                key_same = True
            else:
                kt = self._t(init_state.PRNGKey)
                key = kt.reshape(lead_shape + (1,)).expand(shape)
                key_same = kt is init_state.PRNGKey
            # views of the caller's own leaves follow in-place updates of those leaves; converted COPIES (CPU / other dtype /
            # non-contiguous leaf) would go stale under the same id, so those are rebuilt on every call
            same = key_same and all(t is r for t, r in zip(conv, src[:-1]))
            self._traj_bcast_cache = (shape, src, ref, key) if same else None
        return self.State(physical_state=phys, PRNGKey=key, additions=self._additions(shape, True), reference=ref)

    def sim_ahead(self, init_state, actions, env_properties, obs_stepsize, action_stepsize):
        """Trajectory of a single environment (core_env.py:427-488): actions (n_action_steps, action_dim) ->
        observations (n+1, obs_dim), states, last_state."""
        actions = torch.as_tensor(actions)
        assert actions.ndim == 2, "The actions need to have two dimensions: (n_action_steps, action_dim)"
        assert (
            actions.shape[-1] == self.action_dim
        ), f"The last dimension does not correspond to the action dim which is {self.action_dim}, but {actions.shape[-1]} is given"
        init_physical_state_shape = self._phys_shape(init_state.physical_state)
        assert init_physical_state_shape == (self.physical_state_dim,), (
            "The initial physical state needs to be of shape (env.physical_state_dim,) which is "
            + f"{(self.physical_state_dim,)}, but {init_physical_state_shape} is given"
        )
        obs, st_views, last, N = self._run_sim_ahead(init_state, actions[None], env_properties, obs_stepsize,
                                                     action_stepsize, 1)
        states = self._traj_state(init_state, [v[0] for v in st_views], (), N) if st_views is not None else None
        last_state = replace(init_state, physical_state=self.PhysicalState(
            **{n: t.reshape(()) for n, t in zip(self.STATE_FIELDS, last)}), additions=self._additions((), True))
        return obs[0], states, last_state

    def vmap_sim_ahead(self, init_state, actions, obs_stepsize, action_stepsize, return_rew_trunc_term=False, out=None):
        """Trajectories of all batch_size environments in one persistent kernel launch (core_env.py:571-616):
        actions (batch_size, n_action_steps, action_dim) -> observations (batch_size, n+1, obs_dim), states with
        leaves (batch_size, n+1), last_state with leaves (batch_size,).

        return_rew_trunc_term=True (extension): the same launch also evaluates what
        vmap_generate_rew_trunc_term_ahead(states, actions) would (core_env.py:618-647) and the call returns
        (observations, states, last_state, reward [B,n,1], truncated [B,n+1,TW], terminated [B,n,1]).

        out=(observations, states, last_state) (extension, SURVEY.md §8b "caller-provided via an explicit out="): the triple an
        earlier call of the same shape returned is written again instead of allocating — for chained chunks of a long run
        (`out=prev` with `init_state=prev[2]` is allowed: last_state may alias the initial state).

        Output memory. Every call returns tensors nothing else refers to. Large output sets (>= 1 GiB) are POOLED: a set whose
        tensors have no Python reference, no C++ holder (autograd, DLPack, a view) and were produced on the current stream is
        written again two calls later instead of being re-allocated. The pool cannot see `Tensor.record_stream`: if another stream
        still reads a returned tensor after you dropped your last reference to it, keep that reference until the stream has
        synchronised, or call `env.pool_wait_stream(that_stream)` (the next reuse then waits for it), or set
        `env.trajectory_pool = False`. The first large call of a shape also PLACES its set: it times its own launch into up to four
        candidate state blocks (three launches each), may call `torch.cuda.empty_cache()` once and transiently holds the
        rejected blocks plus a spacer of at most a third of the free device memory; a set that then runs more than 3 % slower than
        its sibling in real launches is replaced, at most twice per shape (`env.trajectory_placement = "off"` switches all of it
        off; `env.release_trajectory_buffers()` frees the dead sets; `env.trajectory_placement_settled` says when it is over)."""
        assert (
            obs_stepsize <= action_stepsize
        ), "The action stepsize should be greater or equal to the observation stepsize."
        actions = torch.as_tensor(actions)
        tiled_in = actions.ndim == 4 and actions.shape[0] * actions.shape[1] == self.batch_size
        assert actions.ndim == 3 or tiled_in, "The actions need to have three dimensions: (batch_size, n_action_steps, action_dim)"
        assert (
            tiled_in or actions.shape[0] == self.batch_size
        ), f"The first dimension does not correspond to the batch size which is {self.batch_size}, but {actions.shape[0]} is given"
        assert (
            actions.shape[-1] == self.action_dim
        ), f"The last dimension does not correspond to the action dim which is {self.action_dim}, but {actions.shape[-1]} is given"
        init_physical_state_shape = self._phys_shape(init_state.physical_state)
        assert init_physical_state_shape == (self.batch_size, self.physical_state_dim), (
            "The initial physical state needs to be of shape (batch_size, physical_state_dim,) which is "
            + f"{(self.batch_size, self.physical_state_dim)}, but {init_physical_state_shape} is given"
        )
        B = self.batch_size
        gym_out = None
        if return_rew_trunc_term:
            if out is not None:
                raise ValueError("vmap_sim_ahead: out= cannot be combined with return_rew_trunc_term")
            obs, st_views, last, N, gym_out = self._run_sim_ahead(init_state, actions, self.env_properties, obs_stepsize,
                                                                  action_stepsize, B, want_gym=True)
        else:
            obs, st_views, last, N = self._run_sim_ahead(init_state, actions, self.env_properties, obs_stepsize,
                                                         action_stepsize, B, out=out)
        if st_views is None:
            states = None
        elif self.traj_layout == "tiled":
            states = self.State(physical_state=self.PhysicalState(**dict(zip(self.STATE_FIELDS, st_views))),
                                PRNGKey=None, additions=None, reference=None)
        else:
            states = self._traj_state(init_state, st_views, (B,), N)
        last_state = self.State(self.PhysicalState(*last), init_state.PRNGKey, self._additions((B,), True), init_state.reference)
        if gym_out is not None:
            return (obs, states, last_state) + tuple(gym_out)
        return obs, states, last_state

    def new_trajectory_buffers(self, init_state, actions, obs_stepsize, action_stepsize, candidates: int = 1):
        """Output buffers for `vmap_sim_ahead(..., out=...)`: the (observations, states, last_state) triple of a first call with
        these inputs. With candidates > 1 that many sets are allocated side by side, one launch into each is timed (HIP events)
        and the fastest set is kept, the others are freed: where the driver places tens of GB of trajectory buffers in physical
        memory moves the trajectory kernel by up to 25 % (HBM write-credit stalls), and about one placement in five
        is a slow one — a long chunked run that reuses its buffers should start on a good one. Returns (triple, probe_ms list)."""
        sets, times = [], []
        for _ in range(max(1, int(candidates))):
            try:
                trip = self.vmap_sim_ahead(init_state, actions, obs_stepsize, action_stepsize)
            except torch.OutOfMemoryError:
                if not sets:
                    raise
                break  # not enough memory for another candidate next to the ones held: choose among those
            if candidates > 1:
                t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                self.vmap_sim_ahead(init_state, actions, obs_stepsize, action_stepsize, out=trip)  # warm (clocks, caches)
                t0.record()
                self.vmap_sim_ahead(init_state, actions, obs_stepsize, action_stepsize, out=trip)
                t1.record()
                t1.synchronize()
                times.append(float(t0.elapsed_time(t1)))
            sets.append(trip)
        best = min(range(len(sets)), key=lambda i: times[i]) if times else 0
        keep = sets[best]
        del sets
        return keep, times

    def make_stepper(self, n_steps: int = 1, graph: bool = False, gym: bool = False):
        """In-place multi-step stepping with static buffers (opt-in; see stepper.Stepper): `n_steps` chained vmap_step
        (gym=True: vmap_gym_step) launches per `run()`, eagerly with pre-built arguments or as one HIP-graph replay."""
        from .stepper import Stepper

        return Stepper(self, n_steps=n_steps, graph=graph, gym=gym)

    def new_actions_buffer(self, n_action_steps: int, layout: str = "lane_major"):
        """A (batch_size, n_action_steps, action_dim) tensor whose memory is lane-major ([K, A, B]); filling this
        and passing it to vmap_sim_ahead lets the kernel read actions fully coalesced with no transposition.
        layout="tiled": a (B/T, T, K, A) view over [B/T, K, A, T] memory (opt-in tiled layout, T = 1024)."""
        if layout == "tiled":
            T = _native.TILE
            assert self.batch_size % T == 0
            buf = torch.empty((self.batch_size // T, n_action_steps, self.action_dim, T), dtype=self.dtype, device=self.device)
            return buf.permute(0, 3, 1, 2)
        buf = torch.empty((n_action_steps, self.action_dim, self.batch_size), dtype=self.dtype, device=self.device)
        return buf.permute(2, 0, 1)

    # ------------------------------------------------------------------ descriptions
    @property
    def obs_description(self):
        return np.hstack([np.array(list(self.STATE_FIELDS)), np.array([n + "_ref" for n in self.control_state])])

    @property
    def action_description(self):
        return np.array(list(self.ACTION_FIELDS))
