#!/usr/bin/env python3
"""Does the headline time depend on where the allocator places the buffers? A dummy allocation of varying size shifts everything."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "exciting-environments_amd")]
import torch
import bench
dev = torch.device("cuda", 0)
def timeit(fn, n=15):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
class A: pass
a = A(); a.workload = "pmsm_euler_f32"; a.batch = 0; a.chunk = 0; a.semantics = "ahead"; a.traj_layout = "lane_major"; a.action_layout = "lane_major"; a.path = "sim_ahead"; a.obs_only = False; a.no_workspace = False; a.no_fused = False
import statistics
arena_gb = int(os.environ.get("ARENA_GB", "0"))
if arena_gb:  # one big segment first: later allocations are carved out of it by the caching allocator
    big = torch.empty(arena_gb << 30, dtype=torch.uint8, device=dev)
    del big
v = []
a.batch = (1 << 22) + int(os.environ.get("EXTRA_B", "0"))
for shift_mb in (0, 3, 129, 257, 700, 1000, 1500, 2049, 3000, 4097, 6000, 9000):
    if not arena_gb:
        torch.cuda.empty_cache()
    dummy = torch.empty(shift_mb << 20, dtype=torch.uint8, device=dev) if shift_mb else None
    env, state, actions, B, Kc, *_ = bench.build_env(a, dev, 0)
    v.append(timeit(lambda: env.vmap_sim_ahead(state, actions, env.tau, env.tau), 10) * (1 << 22) / B)
    del env, state, actions, dummy
print(f"dummy-sweep arena {arena_gb} GB B=2^22{int(os.environ.get('EXTRA_B', '0')):+d} " + os.environ.get("PYTORCH_HIP_ALLOC_CONF", "-") + ": " + " ".join(f"{x:.2f}" for x in v) + f"   best {min(v):.2f} median {statistics.median(v):.2f} mean {sum(v) / len(v):.2f} worst {max(v):.2f}")
