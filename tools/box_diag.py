#!/usr/bin/env python3
"""Why do gpurun boxes differ by up to 30 % on the headline? Copy rate (HBM), a VALU-only kernel (shader clock) and the
headline launch on the same box, plus the clocks rocm-smi reports under load."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "exciting-environments_amd")]
import torch
import bench

dev = torch.device("cuda", 0)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
src = torch.empty(1 << 28, dtype=torch.float32, device=dev).normal_(); dst = torch.empty_like(src)
ms = timeit(lambda: dst.copy_(src)); print(f"copy 1 GiB -> 1 GiB: {2 * (1 << 30) / ms / 1e6:.0f} GB/s")
x = torch.randn(1 << 24, device=dev)
def valu():
    y = x
    for _ in range(8): y = torch.sin(y) * 1.0001 + 0.1
    return y
ms = timeit(valu, 10); print(f"VALU chain (8 x sin*a+b over 2^24): {ms:.3f} ms")
del src, dst
class A: pass
a = A(); a.workload = "pmsm_euler_f32"; a.batch = 0; a.chunk = 0; a.semantics = "ahead"; a.traj_layout = "lane_major"; a.action_layout = "lane_major"; a.path = "sim_ahead"; a.obs_only = False; a.no_workspace = False; a.no_fused = False
env, state, actions, B, Kc, reg, solver, dtype = bench.build_env(a, dev, 0)
ms = timeit(lambda: env.vmap_sim_ahead(state, actions, env.tau, env.tau), 20); print(f"headline launch: {ms:.3f} ms  ({68 * B * Kc / ms / 1e6 / 8000:.3f} of 8 TB/s)")
for extra in (4096, 65536 + 4096, -4096):  # a trajectory pitch that is not a power of two: same work per environment
    a.batch = (1 << 22) + extra
    e2, s2, a2, B2, K2, *_ = bench.build_env(a, dev, 0)
    ms2 = timeit(lambda: e2.vmap_sim_ahead(s2, a2, e2.tau, e2.tau), 20)
    print(f"  B = 2^22 {extra:+d}: {ms2:.3f} ms  ({68 * B2 * K2 / ms2 / 1e6 / 8000:.3f} of 8 TB/s)")
    del e2, s2, a2
p = subprocess.Popen(["rocm-smi", "--showclocks", "--showpower"], stdout=subprocess.PIPE, text=True)
for _ in range(60): env.vmap_sim_ahead(state, actions, env.tau, env.tau)
out = p.communicate()[0]; torch.cuda.synchronize()
print("\n".join(l for l in out.splitlines() if "clk" in l or "Power" in l))
