#!/usr/bin/env python3
"""Per-call wall times of a chained large vmap_sim_ahead run (is there a call that runs a placement search after
trajectory_placement_settled said True?). usage: python tools/glitch_probe.py [REGISTRY_NAME] [control,...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "exciting-environments_amd")]
import torch
from exciting_environments_amd import EnvironmentRegistry, _native

name = sys.argv[1] if len(sys.argv) > 1 else "CART_POLE"
control = sys.argv[2].split(",") if len(sys.argv) > 2 and sys.argv[2] else []
B, K = 1 << 22, 100
env = getattr(EnvironmentRegistry, name).make(batch_size=B, device="cuda:0", control_state=list(control))
_, st = env.vmap_reset()
for n in control:
    setattr(st.reference, n, torch.zeros(B, device="cuda:0"))
acts = env.new_actions_buffer(K)
acts.uniform_(-1, 1)
out = None
for it in range(40):
    t0 = time.perf_counter()
    out = env.vmap_sim_ahead(st, acts, env.tau, env.tau)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    sets = [(t.uses, None if t.steady_ms is None else round(t.steady_ms, 3)) for t in env._placement.sets]
    print(f"call {it:2d} {ms:8.2f} ms settled={env.trajectory_placement_settled} sets={sets} replaced={dict(env._placement.replaced)} "
          f"placement={ {k: v for k, v in (env.last_placement or {}).items() if k in ('pattern_over_fill', 'chosen_ms', 'arena_rejected')} }", flush=True)
