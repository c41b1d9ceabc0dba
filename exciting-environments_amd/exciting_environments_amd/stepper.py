"""Stepper — in-place multi-step stepping with static buffers, optionally as ONE HIP-graph replay.

No reference counterpart: the reference's per-step loop (README.md:28-32) dispatches one XLA executable per step and
allocates fresh arrays every time. At RL batch sizes (B <= 2^16) one fused step kernel takes a few microseconds, so the host
decides the rate: ``env.vmap_step`` costs a handful of microseconds of Python per call, a Stepper's eager ``run`` one
pre-built ctypes call per step, and with ``graph=True`` the whole n-step chain is a single ``hipGraphLaunch``.

The price is the functional contract: a Stepper OWNS its buffers and updates the state in place. ``run()`` returns views
of those buffers; copy what must survive the next ``run()``. Results are bit-identical to the same sequence of
``env.vmap_step`` / ``env.vmap_gym_step`` calls (tests/test_gpu_stepper.py).
"""
from __future__ import annotations

import ctypes

import torch

from . import _native


class Stepper:
    def __init__(self, env, n_steps: int = 1, graph: bool = False, gym: bool = False):
        if n_steps < 1:
            raise ValueError("n_steps must be >= 1")
        if env.device.type != "cuda":
            raise RuntimeError("Stepper: the environment must live on a HIP device (there is no CPU fallback)")
        self.env, self.n_steps, self.gym = env, int(n_steps), bool(gym)
        B, S, A, O = env.batch_size, env.physical_state_dim, env.action_dim, env._obs_dim()
        dt, dev = env.dtype, env.device
        self.actions = torch.zeros((n_steps, B, A), dtype=dt, device=dev)  # fill in place (e.g. a policy's output)
        self.obs = torch.empty((n_steps, B, O), dtype=dt, device=dev)
        isz = self.actions.element_size()
        al = 16 // isz
        Bp = (B + al - 1) // al * al
        self._leaf_buf = torch.zeros((S, Bp), dtype=dt, device=dev)
        self._leaves = tuple(self._leaf_buf[j, :B] for j in range(S))
        self._ref_buf = None
        self.reward = self.terminated = self.truncated = None
        if gym:
            TW = _native.truncated_width(env.ENV_ID, len(env.control_state))
            self.reward = torch.empty((n_steps, B, 1), dtype=dt, device=dev)
            self.terminated = torch.empty((n_steps, B, 1), dtype=torch.bool, device=dev)
            self.truncated = torch.empty((n_steps, B, TW), dtype=torch.bool, device=dev)
        _, init = env.vmap_reset()
        self._template = init
        self.state = None
        self._graph = None
        self._want_graph = bool(graph)
        self._args = None
        self.reset(init)

    # ------------------------------------------------------------------------------------------------------------------
    def reset(self, state):
        """Copy `state` (leaves [B]) into the static buffers; reference leaves of the controlled fields are copied too."""
        env = self.env
        B = env.batch_size
        for dst, n in zip(self._leaves, env.STATE_FIELDS):
            dst.copy_(env._t(getattr(state.physical_state, n), (B,)))
        reference = state.reference
        if env.control_state:
            if self._ref_buf is None:
                self._ref_buf = {n: torch.empty(B, dtype=env.dtype, device=env.device) for n in env.control_state}
            for n in env.control_state:
                self._ref_buf[n].copy_(env._t(getattr(state.reference, n), (B,)))
            ref = {n: getattr(state.reference, n) for n in env.STATE_FIELDS}
            ref.update(self._ref_buf)
            reference = env.PhysicalState(**ref)
        self.state = env.State(env.PhysicalState(*self._leaves), state.PRNGKey, env._additions((B,), True), reference)
        if self._args is None:
            self._build_args()
        return self.state

    def set_reference(self, name: str, value):
        """Overwrite the static reference leaf of a controlled field in place (visible to the next run / replay)."""
        self._ref_buf[name].copy_(self.env._t(value, (self.env.batch_size,)))

    def _build_args(self):
        env = self.env
        B, A, O = env.batch_size, env.action_dim, env._obs_dim()
        isz = self.actions.element_size()
        props, keep = env._props_for(env.env_properties, B)
        self._keep = [props, keep]
        control_ref = None
        if env.control_state:
            tens = [self._ref_buf[n] for n in env.control_state]
            ctl = _native.make_control([env.STATE_FIELDS.index(n) for n in env.control_state], tens)
            self._keep.append(ctl)
            control_ref = ctypes.byref(ctl)
        ptrs = _native._ptrs(self._leaves)  # state_out aliases state_in element for element (allowed by the C ABI)
        a0, o0 = self.actions.data_ptr(), self.obs.data_ptr()
        per = []
        for k in range(self.n_steps):
            g = None
            if self.gym:
                TW = self.truncated.shape[-1]
                g = (self.reward.data_ptr() + k * B * isz, self.terminated.data_ptr() + k * B,
                     self.truncated.data_ptr() + k * B * TW)
            per.append((a0 + k * B * A * isz, o0 + k * B * O * isz, g))
        opts = env.launch_opts
        self._args = (env.ENV_ID, env._solver.id, _native.dtype_id(env.dtype), B, ctypes.byref(props), control_ref,
                      float(env.tau), ptrs, None if opts is None else ctypes.byref(opts),
                      env.device.index if env.device.index is not None else torch.cuda.current_device(), per)

    def _launch_all(self):
        eid, sid, dtc, B, pref, cref, tau, ptrs, oref, idx, per = self._args
        step_raw = _native.step_raw
        stream = _native.raw_stream(idx)  # once per run: all n_steps launches go to the caller's current stream
        for a_ptr, o_ptr, g in per:
            step_raw(eid, sid, dtc, B, pref, cref, tau, ptrs, a_ptr, ptrs, o_ptr, oref, stream, g)

    def run(self):
        """n_steps steps from the current in-place state with `self.actions`; returns (obs [n,B,O], state) — plus
        (reward, terminated, truncated) when built with gym=True — all views of the static buffers."""
        env = self.env
        with _native._on_device(env.device):
            if self._want_graph:
                if self._graph is None:
                    self._capture()
                self._graph.replay()
            else:
                self._launch_all()
        if self.gym:
            return self.obs, self.reward, self.terminated, self.truncated, self.state
        return self.obs, self.state

    def _capture(self):
        # capture only enqueues: the in-place state is not advanced by building the graph
        side = torch.cuda.Stream(device=self.env.device)
        side.wait_stream(torch.cuda.current_stream(self.env.device))
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            self._launch_all()
        torch.cuda.current_stream(self.env.device).wait_stream(side)
        self._graph = g
