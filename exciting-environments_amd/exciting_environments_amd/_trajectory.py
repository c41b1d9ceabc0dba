"""How a `vmap_sim_ahead` / `sim_ahead` call becomes ONE launch of the trajectory kernel: which layout the incoming actions have, where
the output arrays come from (one shared allocation for small calls, the pooled and placed sets of `_placement.py` for large ones, the
caller's own triple for `out=`), the optional transposition workspace and the gym-output arrays — then a single ctypes call with
plain arguments (`excenv_sim_ahead_ws`, include/excenv.h). Mixed into `CoreEnvironment` (core_env.py), which keeps the reference's
API surface: argument checks, shapes, pytrees."""
from __future__ import annotations

import ctypes

import torch

from . import _native


class TrajectoryLaunchMixin:
    def _run_sim_ahead(self, init_state, actions, env_properties, obs_stepsize, action_stepsize, B, want_gym=False, out=None):
        S, A, OW = self.physical_state_dim, self.action_dim, self._obs_dim()
        actions = torch.as_tensor(actions)
        K = actions.shape[-2]
        sub = self._n_substeps(K, obs_stepsize, action_stepsize)
        N = K * sub
        props, keep = self._props_for(env_properties, B)
        st_in = [self._t(getattr(init_state.physical_state, n), (B,)) for n in self.STATE_FIELDS]
        control, refs = self._control(init_state, (B,))

        if actions.device != self.device or actions.dtype != self.dtype:
            actions = actions.to(device=self.device, dtype=self.dtype)
        T = _native.TILE
        if actions.ndim == 4:  # [B/T, T, K, A] view over tiled [B/T, K, A, T] memory (new_actions_buffer(layout="tiled"))
            assert tuple(actions.shape[:2]) == (B // T, T) and tuple(actions.stride()) == (K * A * T, 1, A * T, T), \
                "4-D actions must come from new_actions_buffer(K, layout='tiled')"
            a_layout = _native.LAYOUT_TILED
        elif K > 0 and B > 0 and tuple(actions.stride()) == (1, A * B, B):
            a_layout = _native.LAYOUT_LANE_MAJOR  # a [K, A, B] buffer viewed as [B, K, A]
        else:
            actions = actions.contiguous()
            a_layout = _native.LAYOUT_ENV_MAJOR

        want_states = self.store_state_trajectory
        if self.traj_layout == "lane_major" and B > 0 and self.device.type == "cuda" and not (want_gym and out is not None):
            # (with the gym trajectories too since round 4: their launches used to write unpooled, unplaced buffers — observations
            # and seven leaves allocated back to back, the slow placement level — and cost 5.5 ... 6.4 ms where this path gives 5.5)
            return self._run_sim_ahead_lane_major(init_state, actions, a_layout, props, control, st_in, obs_stepsize, B, K, sub,
                                                  want_states, out, want_gym)
        if out is not None:
            raise ValueError("vmap_sim_ahead(out=...) is available for the default lane-major trajectories without gym outputs")
        isz = 4 if self.dtype is torch.float32 else 8
        if (self.traj_layout == "env_major" and not want_gym and B > 0 and self.device.type == "cuda"
                and (OW + (S if want_states else 0)) * (N + 1) * B * isz >= self._placement.PLACED_BYTES):
            return self._run_sim_ahead_env_major_large(actions, a_layout, props, control, st_in, obs_stepsize, B, K, sub, want_states)
        if self.traj_layout == "lane_major":
            obs_buf = torch.empty((N + 1, OW, B), dtype=self.dtype, device=self.device)
            st_buf = [torch.empty((N + 1, B), dtype=self.dtype, device=self.device) for _ in range(S)] if want_states else None
            t_layout = _native.LAYOUT_LANE_MAJOR
            observations = obs_buf.permute(2, 0, 1)
            st_views = [b.t() for b in st_buf] if want_states else None
        elif self.traj_layout == "tiled":
            # opt-in, NOT reference-shaped: tiles of T envs, each tile lane-major -> views [B/T, T, N+1, OW] / [B/T, T, N+1]
            assert B % T == 0, f"traj_layout='tiled' needs batch_size % {T} == 0"
            obs_buf = torch.empty((B // T, N + 1, OW, T), dtype=self.dtype, device=self.device)
            st_buf = [torch.empty((B // T, N + 1, T), dtype=self.dtype, device=self.device) for _ in range(S)] if want_states else None
            t_layout = _native.LAYOUT_TILED
            observations = obs_buf.permute(0, 3, 1, 2)
            st_views = [b.permute(0, 2, 1) for b in st_buf] if want_states else None
        elif self.traj_layout == "env_major":
            obs_buf = torch.empty((B, N + 1, OW), dtype=self.dtype, device=self.device)
            st_buf = [torch.empty((B, N + 1), dtype=self.dtype, device=self.device) for _ in range(S)] if want_states else None
            t_layout = _native.LAYOUT_ENV_MAJOR
            observations, st_views = obs_buf, st_buf
        else:
            raise ValueError(f"traj_layout must be 'lane_major', 'env_major' or 'tiled', got {self.traj_layout!r}")
        last = [torch.empty(B, dtype=self.dtype, device=self.device) for _ in range(S)]
        sem = {"ahead": _native.SEM_AHEAD, "step": _native.SEM_STEP}[self.sim_ahead_semantics]
        workspace = None
        if self.env_major_workspace and B > 0 and _native.LAYOUT_ENV_MAJOR in (a_layout, t_layout):
            # env-major (row-major) buffers: let the library transpose through a scratch buffer instead of issuing
            # scattered 4-byte accesses (excenv_sim_ahead_ws)
            nbytes = _native.sim_ahead_workspace_bytes(self.ENV_ID, self.dtype, B, K, sub, len(self.control_state),
                                                       a_layout, t_layout, want_states)
            if nbytes > 0:
                workspace = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        opts = self.launch_opts
        if _native.LAYOUT_ENV_MAJOR in (a_layout, t_layout) and not self.env_major_fused:
            opts = _native.launch_opts(opts.envs_per_lane if opts else 0, 1, opts.lds_pad_bytes if opts else 0)
        gym_bufs = gym_out = None
        if want_gym:  # reward / terminated / truncated trajectories from the same launch (excenv_traj_gym_t)
            if t_layout == _native.LAYOUT_TILED:
                raise ValueError("return_rew_trunc_term is not available with traj_layout='tiled'")
            TW = _native.truncated_width(self.ENV_ID, len(self.control_state))
            if t_layout == _native.LAYOUT_LANE_MAJOR:
                rew = torch.empty((N, B), dtype=self.dtype, device=self.device)
                term = torch.empty((N, B), dtype=torch.bool, device=self.device)
                trunc = torch.empty((N + 1, B, TW), dtype=torch.bool, device=self.device)
                gym_out = (rew.t()[..., None], trunc.permute(1, 0, 2), term.t()[..., None])
            else:
                rew = torch.empty((B, N, 1), dtype=self.dtype, device=self.device)
                term = torch.empty((B, N, 1), dtype=torch.bool, device=self.device)
                trunc = torch.empty((B, N + 1, TW), dtype=torch.bool, device=self.device)
                gym_out = (rew, trunc, term)
            gym_bufs = (rew, term, trunc)
        _native.sim_ahead(self.ENV_ID, self._solver.id, self.dtype, B, K, sub, props, control, float(obs_stepsize),
                          float(self.tau), st_in, actions, a_layout, obs_buf, st_buf, t_layout, last, sem, workspace, opts,
                          gym_bufs)
        if want_gym:
            return observations, st_views, last, N, gym_out
        return observations, st_views, last, N

    def _run_sim_ahead_env_major_large(self, actions, a_layout, props, control, st_in, obs_stepsize, B, K, sub, want_states):
        """Row-major (reference-shaped) trajectories of at least _PLACED_TRAJ_BYTES: the fused env-major kernels write scattered
        runs and depend on where observations and state leaves lie even more than the lane-major kernel does
        (tools/em_placement.py: 7.0 ... 10.9 ms for the same launch), so these sets are pooled and placed like the lane-major
        ones."""
        S, OW = self.physical_state_dim, self._obs_dim()
        N = K * sub
        rows = N + 1
        dt, dev = self.dtype, self.device
        isz = 4 if dt is torch.float32 else 8
        last_e = (B * isz + 15) // 16 * 16 // isz
        opts = self.launch_opts
        if not self.env_major_fused:
            opts = _native.launch_opts(opts.envs_per_lane if opts else 0, 1, opts.lds_pad_bytes if opts else 0)
        ws = ws_ptr = None
        ws_bytes = 0
        if self.env_major_workspace:
            ws_bytes = _native.sim_ahead_workspace_bytes(self.ENV_ID, dt, B, K, sub, len(self.control_state), a_layout,
                                                         _native.LAYOUT_ENV_MAJOR, want_states)
            if ws_bytes > 0:
                ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)  # stream-ordered: free to die when this function returns
                ws_ptr = ws.data_ptr()
        sem = _native.SEM_AHEAD if self.sim_ahead_semantics == "ahead" else _native.SEM_STEP
        st_in_ptrs = _native._ptrs(st_in)

        def launch(o_ptr, t_ptrs, l_ptrs):
            with _native._on_device(dev):
                _native.sim_ahead_raw(self.ENV_ID, self._solver.id, 0 if dt is torch.float32 else 1, B, K, sub, ctypes.byref(props),
                                      ctypes.byref(control) if control is not None else None, float(obs_stepsize), float(self.tau),
                                      st_in_ptrs, actions.data_ptr() if K > 0 else None, a_layout, o_ptr,
                                      t_ptrs if want_states else None, _native.LAYOUT_ENV_MAJOR, l_ptrs, sem, ws_ptr,
                                      ws_bytes if ws_ptr is not None else 0, None if opts is None else ctypes.byref(opts),
                                      _native._raw_stream(dev))

        ts = self._placement.acquire(B, rows, OW, S, want_states, last_e, isz, launch, env_major=True)
        self._placement.drain_waits()
        self._placement.timed_launch(ts, lambda: launch(ts.obs_ptr, ts.traj_ptrs, ts.last_ptrs), (OW + (S if want_states else 0)) * rows * B * isz)
        return ts.observations, ts.st_views, ts.last, N

    # Trajectories up to this size come out of ONE allocation (observations, state leaves and last_state are views of it):
    # at RL / MPC batch sizes the launch takes ~100 us and 2 S + 1 allocator calls plus as many view objects cost as much.
    # Larger outputs keep one allocation per returned array so that dropping the states frees their memory.
    _SHARED_TRAJ_BYTES = 32 << 20

    def _run_sim_ahead_lane_major(self, init_state, actions, a_layout, props, control, st_in, obs_stepsize, B, K, sub,
                                  want_states, out=None, want_gym=False):
        """The default layout: buffers carved from one or two allocations, pointers computed from the base address, one ctypes call
        with plain arguments (same launch as the general path above). want_gym: the reward / terminated / truncated trajectories
        come from the same launch into arrays of their own (returned as a fifth element)."""
        S, OW = self.physical_state_dim, self._obs_dim()
        N = K * sub
        rows = N + 1
        dt, dev = self.dtype, self.device
        isz = 4 if dt is torch.float32 else 8
        al = 16 // isz
        up = lambda n: (n + al - 1) // al * al
        obs_e, leaf_e, last_e = up(rows * OW * B), up(rows * B), up(B)
        traj_e = obs_e + (S * leaf_e if want_states else 0)
        opts = self.launch_opts
        ws_e = ws_bytes = 0
        if a_layout == _native.LAYOUT_ENV_MAJOR:
            # row-major actions (a plain [B, K, A] tensor, what the reference's call hands over): large batches of broadcast-
            # property environments read them inside the trajectory kernel (per-wave LDS piece ring); the rest
            # is transposed through scratch by the library
            if not self.env_major_fused:
                opts = _native.launch_opts(opts.envs_per_lane if opts else 0, 1, opts.lds_pad_bytes if opts else 0,
                                           (opts.flags if opts else 0) | _native.OPT_NO_FUSED_ACTIONS)
            # the launch fuses only when every state array allows 16-byte accesses as well (launch.hpp: vec_ok); the output arrays
            # made below always do. A "fused" answer for a call that then does not fuse would skip the workspace and drop the call
            # to the generic-stride read of the actions — correct, but far slower than the transposition it replaced.
            aligned = all(t.data_ptr() % 16 == 0 for t in st_in)
            fk = (B, K, sub, len(self.control_state), actions.data_ptr() % 16, id(props), bool(want_gym), aligned,
                  None if opts is None else (opts.envs_per_lane, opts.flags))
            if self._fused_actions_cache is None or self._fused_actions_cache[0] != fk:
                self._fused_actions_cache = (fk, aligned and _native.sim_ahead_fuses_actions(
                    self.ENV_ID, self._solver.id, dt, B, K, props, len(self.control_state), bool(want_gym), a_layout,
                    _native.LAYOUT_LANE_MAJOR, actions.data_ptr(), opts))
            if self.env_major_workspace and not self._fused_actions_cache[1]:
                wk = (B, K, sub, len(self.control_state), want_states)
                if self._ws_bytes_cache is None or self._ws_bytes_cache[0] != wk:
                    self._ws_bytes_cache = (wk, _native.sim_ahead_workspace_bytes(
                        self.ENV_ID, dt, B, K, sub, len(self.control_state), a_layout, _native.LAYOUT_LANE_MAJOR, want_states))
                ws_bytes = self._ws_bytes_cache[1]
                ws_e = up((ws_bytes + isz - 1) // isz)
        shared = (traj_e + S * last_e + ws_e) * isz <= self._SHARED_TRAJ_BYTES
        st_views = None
        traj_ptrs = None
        ws_ptr = None
        sem = _native.SEM_AHEAD if self.sim_ahead_semantics == "ahead" else _native.SEM_STEP
        st_in_ptrs = _native._ptrs(st_in)

        gym_out = gym_ref = None
        if want_gym:  # excenv_traj_gym_t, lane-major: reward / terminated [N][B], truncated [N + 1][B][TW]
            TW = _native.truncated_width(self.ENV_ID, len(self.control_state))
            rew = torch.empty((N, B), dtype=dt, device=dev)
            term = torch.empty((N, B), dtype=torch.bool, device=dev)
            trunc = torch.empty((N + 1, B, TW), dtype=torch.bool, device=dev)
            gym_out = (rew.t()[..., None], trunc.permute(1, 0, 2), term.t()[..., None])
            gym_struct = _native.TrajGym(rew.data_ptr(), term.data_ptr(), trunc.data_ptr())
            gym_ref = ctypes.byref(gym_struct)
        done = (lambda *r: r + (gym_out,)) if want_gym else (lambda *r: r)

        def launch(o_ptr, t_ptrs, l_ptrs):  # the trajectory launch of this call into the given output buffers
            with _native._on_device(dev):
                _native.sim_ahead_raw(self.ENV_ID, self._solver.id, 0 if dt is torch.float32 else 1, B, K, sub, ctypes.byref(props),
                                      ctypes.byref(control) if control is not None else None, float(obs_stepsize), float(self.tau),
                                      st_in_ptrs, actions.data_ptr() if K > 0 else None, a_layout, o_ptr,
                                      t_ptrs if want_states else None, _native.LAYOUT_LANE_MAJOR, l_ptrs, sem, ws_ptr,
                                      ws_bytes if ws_ptr is not None else 0, None if opts is None else ctypes.byref(opts),
                                      _native._raw_stream(dev), gym_ref)

        if out is not None:
            # the caller hands back what an earlier call of the same shape returned: same buffers, no allocation
            observations, o_states, o_last = out
            on_dev = lambda t: t.device.type == dev.type and (dev.index is None or t.device.index == dev.index)
            ok = (isinstance(observations, torch.Tensor) and observations.dtype is dt and on_dev(observations)
                  and tuple(observations.shape) == (B, rows, OW) and tuple(observations.stride()) == (1, OW * B, B))
            last = tuple(getattr(o_last.physical_state, n) for n in self.STATE_FIELDS)
            ok = ok and all(isinstance(t, torch.Tensor) and t.dtype is dt and on_dev(t) and tuple(t.shape) == (B,)
                            and t.is_contiguous() for t in last)
            if want_states:
                ok = ok and o_states is not None
                st_views = tuple(getattr(o_states.physical_state, n) for n in self.STATE_FIELDS) if ok else None
                ok = ok and all(isinstance(t, torch.Tensor) and t.dtype is dt and on_dev(t)
                                and tuple(t.shape) == (B, rows) and tuple(t.stride()) == (1, B) for t in st_views)
            if not ok:
                raise ValueError("vmap_sim_ahead(out=...): pass the (observations, states, last_state) an earlier call with the "
                                 "same batch, horizon, layout and dtype returned")
            obs_ptr = observations.data_ptr()
            if want_states:
                traj_ptrs = _native._ptrs(st_views)
            last_ptrs = _native._ptrs(last)
            if ws_e:
                ws = torch.empty(ws_e, dtype=dt, device=dev)
                ws_ptr = ws.data_ptr()
        elif shared:
            buf = torch.empty(traj_e + S * last_e + ws_e, dtype=dt, device=dev)
            base = buf.data_ptr()
            if ws_e:
                ws_ptr = base + (traj_e + S * last_e) * isz
            observations = buf.as_strided((B, rows, OW), (1, OW * B, B))
            if want_states:
                st_views = buf.as_strided((S, B, rows), (leaf_e, 1, B), obs_e).unbind(0)
                traj_ptrs = _native.ptr_array([base + (obs_e + j * leaf_e) * isz for j in range(S)])
            last = buf.as_strided((S, B), (last_e, 1), traj_e).unbind(0)
            last_ptrs = _native.ptr_array([base + (traj_e + j * last_e) * isz for j in range(S)])
            obs_ptr = base
        else:
            if ws_e:
                ws = torch.empty(ws_e, dtype=dt, device=dev)  # stream-ordered: free to die when this function returns
                ws_ptr = ws.data_ptr()
            # lane-major actions: the launch's access pattern can be replayed without arithmetic to judge a placement
            # (row-major actions too, since round 5: the replay then reads the same K * A * B values in lane-major order — the
            # actions are 12 % of the traffic at most and what is being judged is where the OUTPUTS lie; without it such calls fell
            # back to timing the launch itself into candidate blocks, and a traced process sat at 6.5 ms where the bench had 5.0)
            pctx = (actions.data_ptr(), self.action_dim) if (a_layout in (_native.LAYOUT_LANE_MAJOR, _native.LAYOUT_ENV_MAJOR)
                                                              and sub == 1 and K >= 9) else None
            ts = self._placement.acquire(B, rows, OW, S, want_states, last_e, isz, lambda o, t, l: launch(o, t, l), pattern_ctx=pctx)
            observations, st_views, last = ts.observations, ts.st_views, ts.last
            obs_ptr, traj_ptrs, last_ptrs = ts.obs_ptr, ts.traj_ptrs, ts.last_ptrs
            self._placement.drain_waits()
            self._placement.timed_launch(ts, lambda: launch(obs_ptr, traj_ptrs, last_ptrs), (OW + (S if want_states else 0)) * rows * B * isz)
            return done(observations, st_views, last, N)
        launch(obs_ptr, traj_ptrs, last_ptrs)
        return done(observations, st_views, last, N)

