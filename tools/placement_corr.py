#!/usr/bin/env python3
"""Does excenv_stream_pattern (the library's no-arithmetic access pattern) see the same slow / fast placements as the real
trajectory kernel? Several placements of the headline's buffers in ONE process: per placement the real launch (out= the same
triple), the pattern over the same buffers, and the pattern with every stream inside the observation buffer."""
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
    return float(np.median(ts))


class A: pass
a = A(); a.workload = "pmsm_euler_f32"; a.batch = 0; a.chunk = 0; a.semantics = "ahead"; a.traj_layout = "lane_major"; a.action_layout = "lane_major"; a.path = "sim_ahead"; a.obs_only = False; a.no_workspace = False; a.no_fused = False; a.no_pool = True
arena_gb = int(os.environ.get("ARENA_GB", "0"))
if arena_gb:
    big = torch.empty(arena_gb << 30, dtype=torch.uint8, device=dev)
    del big
print(f"arena {arena_gb} GB; columns: dummy MB | real kernel ms | pattern ms (GB/s) | one-region pattern ms (GB/s) | pattern NW=15 NR=0 | fill GB/s")
for shift_mb in (0, 3, 129, 700, 1500, 2049, 4097, 9000, 17000, 33000):
    if not arena_gb:
        torch.cuda.empty_cache()
    dummy = torch.empty(shift_mb << 20, dtype=torch.uint8, device=dev) if shift_mb else None
    env, state, actions, B, Kc, *_ = bench.build_env(a, dev, 0)
    trip = env.vmap_sim_ahead(state, actions, env.tau, env.tau)
    t_real = timed(lambda: env.vmap_sim_ahead(state, actions, env.tau, env.tau, out=trip))
    obs, states, _ = trip
    leaves = [getattr(states.physical_state, n) for n in env.STATE_FIELDS]
    rb, A_, O, S = B * 4, env.action_dim, obs.shape[2], len(leaves)
    stream = _native.raw_stream(0)
    rd, rd_rs = [actions.data_ptr() + c * rb for c in range(A_)], [A_ * rb] * A_
    ob = obs.data_ptr()
    rows = Kc - 1
    wr, wr_rs = [ob + c * rb for c in range(O)] + [l.data_ptr() for l in leaves], [O * rb] * O + [rb] * S
    t_pat = timed(lambda: _native.stream_pattern(rd, rd_rs, wr, wr_rs, rb, rows, stream))
    W = O + S
    rows1 = min(rows, (Kc + 1) * O // W)
    t_one = timed(lambda: _native.stream_pattern(rd, rd_rs, [ob + q * rb for q in range(W)], [W * rb] * W, rb, rows1, stream)) * rows / rows1
    t_nr0 = timed(lambda: _native.stream_pattern([], [], wr, wr_rs, rb, rows, stream))
    flat = [obs.permute(1, 2, 0)] + [l.t() for l in leaves]
    def fill():
        for t in flat: t.fill_(0.0)
    t_fill = timed(fill)
    pb = (A_ + O + S) * rb * rows
    print(f"{shift_mb:6d} | {t_real:6.3f} | {t_pat:6.3f} ({pb / t_pat / 1e6:5.0f}) | {t_one:6.3f} ({pb / t_one / 1e6:5.0f}) | "
          f"{t_nr0:6.3f} | {sum(t.numel() for t in flat) * 4 / t_fill / 1e6:5.0f}   ptrs obs {ob:#x} st0 {leaves[0].data_ptr():#x} st6 {leaves[6].data_ptr():#x}", flush=True)
    del env, state, actions, dummy, trip, obs, states, leaves, flat
