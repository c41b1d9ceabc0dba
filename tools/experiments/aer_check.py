#!/usr/bin/env python3
"""Goes with register_windows.patch (a measured-and-removed experiment; apply the patch and rebuild first): the
register-window form of the row-major action read (kernels.hpp, AEM == 2), narrow and in 1024-thread workgroups — same bits as the
lane-major call at a batch that takes the wide form, and how fast? (GPU box; the patched library reads EXCENV_AEM_REG per call)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "exciting-environments_amd")]
import torch

from exciting_environments_amd import EnvironmentRegistry, _native

B, K = 1 << 20, int(sys.argv[1]) if len(sys.argv) > 1 else 100
VAR = sys.argv[2] if len(sys.argv) > 2 else "EXCENV_AEM_REG"   # the per-call switch of the experiment's build
MODES = sys.argv[3].split(",") if len(sys.argv) > 3 else ["0", "1", "2"]
for name in ("PENDULUM", "MASS_SPRING_DAMPER", "FLUID_TANK"):
    env = getattr(EnvironmentRegistry, name).make(batch_size=B, device="cuda:0")
    _, st = env.vmap_reset()
    g = torch.Generator(device="cuda:0").manual_seed(3)
    plain = torch.rand((B, K, env.action_dim), generator=g, device="cuda:0") * 2 - 1
    lane = env.new_actions_buffer(K)
    lane.copy_(plain)
    os.environ[VAR] = "0"
    want = env.vmap_sim_ahead(st, lane, env.tau, env.tau)
    w_obs, w_last = want[0].clone(), [getattr(want[2].physical_state, n).clone() for n in env.STATE_FIELDS]
    w_states = [getattr(want[1].physical_state, n).clone() for n in env.STATE_FIELDS]
    print(name, "lane-major:", _native.last_launch())
    for mode in MODES:
        os.environ[VAR] = mode
        got = env.vmap_sim_ahead(st, plain, env.tau, env.tau)
        torch.cuda.synchronize()
        form = _native.last_launch()
        ok = torch.equal(got[0], w_obs) and all(torch.equal(getattr(got[2].physical_state, n), w) for n, w in zip(env.STATE_FIELDS, w_last))
        ok = ok and all(torch.equal(getattr(got[1].physical_state, n), w) for n, w in zip(env.STATE_FIELDS, w_states))
        ts = []
        for _ in range(12):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            got = env.vmap_sim_ahead(st, plain, env.tau, env.tau)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        print(f"  {VAR}={mode}: {form:80s} same bits: {ok}   wall ms min {min(ts):.3f} median {sorted(ts)[len(ts) // 2]:.3f}", flush=True)
    del env, want, got
