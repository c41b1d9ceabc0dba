#!/usr/bin/env python3
"""One environment per lane (per-environment property arrays, or small batches): the trajectory launch with and without the per-row
workgroup barrier (kernels.hpp row_sync) over the batch size. Run twice: EXCENV_ROW_SYNC=0 (off) and with
EXCENV_HIP_LIB=ab_libs/libexcenv_rsall.so (on at every batch; tools/build_variant.sh rsall -DEXCENV_ROW_SYNC_MIN_BATCH=1)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "exciting-environments_amd")]
import torch
from exciting_environments_amd import EnvironmentRegistry, _native

K = 100
for name in ("PMSM", "PENDULUM", "CART_POLE"):
    for lb in (14, 16, 17, 18, 19, 20, 22):
        B = 1 << lb
        env = getattr(EnvironmentRegistry, name).make(batch_size=B, device="cuda:0")
        env.launch_opts = _native.launch_opts(envs_per_lane=1)
        env.trajectory_placement = "off"
        _, st = env.vmap_reset()
        acts = env.new_actions_buffer(K)
        acts.uniform_(-1, 1)
        for _ in range(6):
            out = env.vmap_sim_ahead(st, acts, env.tau, env.tau)
        torch.cuda.synchronize()
        n = 10 if lb >= 20 else 50
        t0 = time.perf_counter()
        for _ in range(n):
            out = env.vmap_sim_ahead(st, acts, env.tau, env.tau)
        torch.cuda.synchronize()
        print(f"{name:10s} B=2^{lb:<2d} {(time.perf_counter() - t0) / n * 1e3:8.3f} ms  [{_native.last_launch()}]", flush=True)
        del out, env, acts, st
