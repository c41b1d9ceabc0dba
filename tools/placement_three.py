#!/usr/bin/env python3
"""Two or three regions for the few-stream workloads (C2 pendulum, C4 mass-spring-damper: 2 observation components + 2 state
leaves)? The real launch through vmap_sim_ahead(out=...) into hand-placed buffers: observations | leaves together, against
observations | leaf 0 | leaf 1, each '|' a 16 GiB hipMalloc spacer held while the next buffer is allocated."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "exciting-environments_amd")]
import numpy as np
import torch
import bench
from exciting_environments_amd import _native

dev = torch.device("cuda", 0)


def timed(fn, n=7):
    fn(); fn()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.min(ts))


class A: pass
for wl in ("pendulum_euler_f32", "msd_tsit5_f64", "pmsm_euler_f32"):
    a = A(); a.workload = wl; a.batch = 0; a.chunk = 0; a.semantics = "ahead"; a.traj_layout = "lane_major"; a.action_layout = "lane_major"; a.path = "sim_ahead"; a.obs_only = False; a.no_workspace = False; a.no_fused = False; a.no_pool = True
    env, state, actions, B, Kc, *_ = bench.build_env(a, dev, 0)
    S, O, rows = len(env.STATE_FIELDS), len(env.obs_description), Kc + 1
    dt = env.dtype
    _, _, last = env.vmap_sim_ahead(state, actions, env.tau, env.tau)
    bytes_per = _native.sim_ahead_bytes(env.ENV_ID, dt, True) * B * Kc
    for groups in ([list(range(S))], [list(range((S + 1) // 2)), list(range((S + 1) // 2, S))], [[j] for j in range(S)][:4] if S <= 4 else [[0, 1, 2], [3, 4], [5, 6]]):
        torch.cuda.empty_cache()
        spacers = []
        obs_buf = torch.empty((rows, O, B), dtype=dt, device=dev)
        leaves = [None] * S
        for g in groups:
            spacers.append(_native.raw_malloc(16 << 30))
            blk = torch.empty((len(g), rows, B), dtype=dt, device=dev)
            for i, j in enumerate(g):
                leaves[j] = blk[i]
        for sp in spacers:
            _native.raw_free(sp)
        states = env.State(env.PhysicalState(*[l.t() for l in leaves]), None, None, None)
        trip = (obs_buf.permute(2, 0, 1), states, last)
        t = timed(lambda: env.vmap_sim_ahead(state, actions, env.tau, env.tau, out=trip))
        print(f"{wl:20s} obs | " + " | ".join("leaves " + ",".join(map(str, g)) for g in groups) + f": {t:.3f} ms  frac {bytes_per / t / 1e6 / 8000:.3f}", flush=True)
        del obs_buf, leaves, states, trip
    del env, state, actions, last
