#!/usr/bin/env python3
"""Deterministic placement instead of a search? ONE arena per shape: [obs A | obs B | gap | states A | states B] — the observations
of both pooled sets at the bottom, their state blocks at the top, `gap` GiB of the arena unused in between, so that a launch's two
kinds of write streams are always far apart in one large allocation (which the driver lays out contiguously when the memory is
free). The headline launch into set A and set B alternately, for several gaps, against the library's searched placement.
Runs ON THE GPU BOX: python tools/placement_arena.py [workload] [--env-major]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "exciting-environments_amd")]
import numpy as np
import torch
import bench
from exciting_environments_amd import _native

dev = torch.device("cuda", 0)
wl = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "pmsm_euler_f32"


def timed_pair(fa, fb, n=8):
    fa(); fb(); fa(); fb()
    ta, tb = [], []
    for _ in range(n):
        for f, acc in ((fa, ta), (fb, tb)):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); f(); b.record(); b.synchronize()
            acc.append(a.elapsed_time(b))
    return float(np.median(ta)), float(np.median(tb))


class A: pass
a = A(); a.workload = wl; a.batch = 0; a.chunk = 0; a.semantics = "ahead"; a.traj_layout = "lane_major"; a.action_layout = "lane_major"; a.path = "sim_ahead"; a.obs_only = False; a.no_workspace = False; a.no_fused = False; a.no_pool = True
env, state, actions, B, Kc, *_ = bench.build_env(a, dev, 0)
S, O, rows = len(env.STATE_FIELDS), len(env.obs_description), Kc + 1
dt = env.dtype
isz = 4 if dt is torch.float32 else 8
_, _, last = env.vmap_sim_ahead(state, actions, env.tau, env.tau)
bytes_per = _native.sim_ahead_bytes(env.ENV_ID, dt, True) * B * Kc
obs_e, blk_e = rows * O * B, S * rows * B
print(f"{wl}: observations {obs_e * isz / 2**30:.1f} GiB, state block {blk_e * isz / 2**30:.1f} GiB per set", flush=True)
for gap_gib in (0, 16, 32, 64, 96):
    torch.cuda.empty_cache()
    gap_e = (gap_gib << 30) // isz
    try:
        arena = torch.empty(2 * obs_e + gap_e + 2 * blk_e, dtype=dt, device=dev)
    except torch.OutOfMemoryError:
        print(f"gap {gap_gib} GiB: out of memory"); continue
    trips = []
    for k in range(2):
        obs_buf = arena[k * obs_e:(k + 1) * obs_e].view(rows, O, B)
        base = 2 * obs_e + gap_e + k * blk_e
        leaves = [arena[base + j * rows * B: base + (j + 1) * rows * B].view(rows, B) for j in range(S)]
        states = env.State(env.PhysicalState(*[l.t() for l in leaves]), None, None, None)
        trips.append((obs_buf.permute(2, 0, 1), states, last))
    ta, tb = timed_pair(lambda: env.vmap_sim_ahead(state, actions, env.tau, env.tau, out=trips[0]),
                        lambda: env.vmap_sim_ahead(state, actions, env.tau, env.tau, out=trips[1]))
    print(f"arena gap {gap_gib:3d} GiB ({arena.numel() * isz / 2**30:6.1f} GiB): set A {ta:.3f} ms ({bytes_per / ta / 1e6 / 8000:.3f})  set B {tb:.3f} ms ({bytes_per / tb / 1e6 / 8000:.3f})", flush=True)
    del arena, trips, obs_buf, leaves, states
# the library's own searched + pooled placement, same process
torch.cuda.empty_cache()
env.trajectory_pool = True
out = None
for it in range(40):
    out = env.vmap_sim_ahead(state, actions, env.tau, env.tau)
    torch.cuda.synchronize()
    if it >= 7 and env.trajectory_placement_settled:
        break
print("library (searched, pooled):", [None if t.steady_ms is None else round(t.steady_ms, 3) for t in env._placement.sets], flush=True)
