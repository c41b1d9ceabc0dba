#!/usr/bin/env python3
"""vmap_step at a large batch: microseconds per chained eager step (no events inside the loop) by lane width and library build
(EXCENV_HIP_LIB). usage: python tools/step_large_batch.py [log2 B] [REGISTRY_NAME]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "exciting-environments_amd")]
import torch
from exciting_environments_amd import EnvironmentRegistry, _native
lb = int(sys.argv[1]) if len(sys.argv) > 1 else 22
name = sys.argv[2] if len(sys.argv) > 2 else "PMSM"
B = 1 << lb
env = getattr(EnvironmentRegistry, name).make(batch_size=B, device="cuda:0")
_, st = env.vmap_reset()
act = torch.rand(B, env.action_dim, device="cuda:0") * 2 - 1
nbytes = _native.step_bytes(env.ENV_ID, env.dtype) * B
for vec in (0, 4, 2, 1):
    env.launch_opts = _native.launch_opts(envs_per_lane=vec) if vec else None
    for _ in range(60):
        obs, st = env.vmap_step(st, act)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        n = 400
        t0 = time.perf_counter()
        for _ in range(n):
            obs, st = env.vmap_step(st, act)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n)
    print(f"{name} B=2^{lb} vec={vec}: {1e6 * best:7.2f} us/step  {nbytes / best / 1e9:7.0f} GB/s ({nbytes / best / 8e12:.3f} of 8 TB/s)  lib={os.path.basename(_native.library_path())}", flush=True)
