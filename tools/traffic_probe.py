#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc passes: a calibration copy of known size (to check how FETCH_SIZE /
WRITE_SIZE count 16-byte-per-lane streams on gfx950, MI355X_MICROARCH.md §HBM) followed by a few launches of
the headline sim_ahead chunk. Usage: python3 tools/traffic_probe.py [workload] [launches]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "exciting-environments_amd")]
import argparse

import torch

import bench

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="pmsm_euler_f32")
ap.add_argument("--launches", type=int, default=3)
ap.add_argument("--traj-layout", default="lane_major")
ap.add_argument("--action-layout", default="lane_major")
ap.add_argument("--batch", type=int, default=0)
ap.add_argument("--chunk", type=int, default=0)
ap.add_argument("--vec", type=int, default=0)
ap.add_argument("--path", default="sim_ahead")
a = ap.parse_args()
a.semantics = "ahead"
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
# calibration: elementwise copy of 1 GiB (reads 2^30 B, writes 2^30 B), far larger than the 256 MiB Infinity Cache
src = torch.empty(1 << 28, dtype=torch.float32, device=dev).normal_()
dst = torch.empty_like(src)
for _ in range(3):
    torch.trunc(src, out=dst)  # a kernel name nothing else in this process uses ("trunc")
torch.cuda.synchronize()
del src, dst
env, state, actions, B, Kc, reg, solver, dtype = bench.build_env(a, dev, 0)
if a.vec:
    from exciting_environments_amd import _native
    env.launch_opts = _native.launch_opts(envs_per_lane=a.vec)
for _ in range(a.launches):
    if a.path == "step":
        obs, state = env.vmap_step(state, actions[:, 0, :].contiguous())
    else:
        obs, states, state = env.vmap_sim_ahead(state, actions, env.tau, env.tau)
torch.cuda.synchronize()
