#!/usr/bin/env python3
"""Can a placement of the headline launch's output buffers be classified ABSOLUTELY (fast / slow level) without launching the
trajectory kernel and without a second placement to compare with? Per placement (observations + state block allocated after a
spacer of a varying size, so that they land in different parts of physical memory): the real launch's time, the write-only
stream pattern over the same buffers (excenv_stream_pattern, 32 rows), the full pattern, and a plain fill of the buffers.
Runs ON THE GPU BOX: python tools/placement_classify.py [n_placements]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "exciting-environments_amd")]
import numpy as np
import torch
import bench
from exciting_environments_amd import _native

dev = torch.device("cuda", 0)
n_place = int(sys.argv[1]) if len(sys.argv) > 1 else 10
workload = sys.argv[2] if len(sys.argv) > 2 else "pmsm_euler_f32"


def timed(fn, n=5):
    fn(); fn()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))


class A: pass
a = A(); a.workload = workload; a.batch = 0; a.chunk = 0; a.semantics = "ahead"; a.traj_layout = "lane_major"; a.action_layout = "lane_major"; a.path = "sim_ahead"; a.obs_only = False; a.no_workspace = False; a.no_fused = False; a.no_pool = True
env, state, actions, B, Kc, *_ = bench.build_env(a, dev, 0)
S, O, rows = len(env.STATE_FIELDS), len(env.obs_description), Kc + 1
_, _, last = env.vmap_sim_ahead(state, actions, env.tau, env.tau)
dt = env.dtype
isz = 4 if dt is torch.float32 else 8
A_ = env.action_dim
rb = B * isz
stream = _native.raw_stream(0)
bytes_per = _native.sim_ahead_bytes(env.ENV_ID, dt, True) * B * Kc
rng = np.random.default_rng(0)
res = []
for i in range(n_place):
    torch.cuda.empty_cache()
    sp = _native.raw_malloc(int(rng.integers(0, 48)) << 30) if i else None
    layout = i % 3  # 0: obs then block back to back; 1: obs | 20 GiB | block; 2: block first, then obs
    hold = []
    if layout == 2:
        blk = torch.empty((S, rows, B), dtype=dt, device=dev)
        obs_buf = torch.empty((rows, O, B), dtype=dt, device=dev)
    else:
        obs_buf = torch.empty((rows, O, B), dtype=dt, device=dev)
        if layout == 1:
            hold.append(_native.raw_malloc(20 << 30))
        blk = torch.empty((S, rows, B), dtype=dt, device=dev)
    for h in hold:
        if h is not None: _native.raw_free(h)
    if sp is not None: _native.raw_free(sp)
    states = env.State(env.PhysicalState(*[blk[j].t() for j in range(S)]), None, None, None)
    trip = (obs_buf.permute(2, 0, 1), states, last)
    t_k = timed(lambda: env.vmap_sim_ahead(state, actions, env.tau, env.tau, out=trip))
    ob = obs_buf.data_ptr()
    wr = [ob + c * rb for c in range(O)] + [blk[j].data_ptr() for j in range(S)]
    wrs = [O * rb] * O + [rb] * S
    R = 32
    t_w = timed(lambda: _native.stream_pattern([], [], wr, wrs, rb, R, stream))
    rd, rds = [actions.data_ptr() + c * rb for c in range(A_)], [A_ * rb] * A_
    t_p = timed(lambda: _native.stream_pattern(rd, rds, wr, wrs, rb, Kc - 1, stream))
    t_f = timed(lambda: (obs_buf.fill_(0.0), blk.fill_(0.0)))
    w_gbs, p_gbs, f_gbs = (O + S) * rb * R / t_w / 1e6, (A_ + O + S) * rb * (Kc - 1) / t_p / 1e6, (obs_buf.numel() + blk.numel()) * isz / t_f / 1e6
    res.append((t_k, w_gbs, p_gbs, f_gbs))
    print(f"placement {i:2d} layout {layout}: kernel {t_k:.3f} ms ({bytes_per / t_k / 1e6 / 8000:.3f})  write-only pattern(32 rows) {w_gbs:6.0f} GB/s  full pattern {p_gbs:6.0f} GB/s  fill {f_gbs:6.0f} GB/s", flush=True)
    del obs_buf, blk, states, trip
r = np.array(res)
print("correlation kernel ms vs write-only pattern GB/s: %.3f; vs full pattern: %.3f; vs fill: %.3f" % (np.corrcoef(r[:, 0], r[:, 1])[0, 1], np.corrcoef(r[:, 0], r[:, 2])[0, 1], np.corrcoef(r[:, 0], r[:, 3])[0, 1]))
