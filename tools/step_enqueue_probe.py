#!/usr/bin/env python3
"""Host time of eager vmap_step calls over the life of a process (enqueue time per call, no synchronisation inside the loop)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "exciting-environments_amd")]
import torch
from exciting_environments_amd import EnvironmentRegistry
B = 1 << int(sys.argv[1]) if len(sys.argv) > 1 else 1024
env = EnvironmentRegistry.PMSM.make(batch_size=B, device="cuda:0")
_, st = env.vmap_reset()
act = torch.rand(B, 2, device="cuda:0") * 2 - 1
ts = []
for i in range(1200):
    t0 = time.perf_counter()
    obs, st = env.vmap_step(st, act)
    ts.append(time.perf_counter() - t0)
torch.cuda.synchronize()
for a in range(0, 1200, 100):
    seg = sorted(ts[a:a + 100])
    print(f"calls {a:4d}-{a + 99:4d}: median {1e6 * seg[50]:7.1f} us  max {1e6 * seg[-1]:9.1f} us  sum {1e3 * sum(seg):7.2f} ms", flush=True)
