#!/usr/bin/env python3
"""env-steps/s of vmap_sim_ahead (chunk 100, full outputs) and vmap_step vs batch size, PMSM Euler fp32 (run on the GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "exciting-environments_amd")]
import argparse
import torch
from exciting_environments_amd import EnvironmentRegistry, _native

ap = argparse.ArgumentParser()
ap.add_argument("--vec", type=int, default=0, help="envs per lane (0 = library heuristic)")
ap.add_argument("--sim-only", action="store_true")
ap.add_argument("--log2", type=int, nargs="*", default=[10, 12, 14, 16, 17, 18, 19, 20, 22, 24])
args = ap.parse_args()
print(f"library: {_native.library_path()}  envs_per_lane: {args.vec or 'auto'}")

print("| batch | sim_ahead env-steps/s | GB/s (68 B) | ms/launch | vmap_step env-steps/s | us/step eager |")
print("|---|---|---|---|---|---|")
for lb in args.log2:
    B, K = 1 << lb, 100
    env = EnvironmentRegistry.PMSM.make(batch_size=B, device="cuda:0")
    if args.vec:
        env.launch_opts = _native.launch_opts(envs_per_lane=args.vec)
    _, st = env.vmap_reset()
    st.physical_state.omega_el = torch.rand(B, device="cuda:0") * 600
    acts = env.new_actions_buffer(K)
    acts.uniform_(-1, 1)
    s = st
    for it in range(40):  # large batches: until the pooled output sets are made, placed and past their replacement window
        o, _, s = env.vmap_sim_ahead(s, acts, env.tau, env.tau)
        torch.cuda.synchronize()
        if it >= 2 and env.trajectory_placement_settled:
            break
    n = max(5, min(200, (1 << 24) // B))
    t0 = time.perf_counter()
    for _ in range(n):
        o, _, s = env.vmap_sim_ahead(s, acts, env.tau, env.tau)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    a0 = acts[:, 0, :].contiguous()
    del o
    if args.sim_only:
        print(f"| 2^{lb} | {B * K / dt:.3e} | {68 * B * K / dt / 1e9:.0f} | {dt * 1e3:.3f} | - | - |", flush=True)
        del env, st, s, acts
        torch.cuda.empty_cache()
        continue
    s2 = st
    for _ in range(20):
        ob, s2 = env.vmap_step(s2, a0)
    torch.cuda.synchronize()
    m = max(50, min(2000, (1 << 26) // B))
    t0 = time.perf_counter()
    for _ in range(m):
        ob, s2 = env.vmap_step(s2, a0)
    torch.cuda.synchronize()
    ds = (time.perf_counter() - t0) / m
    print(f"| 2^{lb} | {B * K / dt:.3e} | {68 * B * K / dt / 1e9:.0f} | {dt * 1e3:.3f} | {B / ds:.3e} | {ds * 1e6:.1f} |", flush=True)
    del env, st, s, s2, acts
    torch.cuda.empty_cache()
