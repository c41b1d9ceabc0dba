#!/usr/bin/env python3
"""Randomised cross-check of the launch forms (GPU box): for random (model, dtype, solver, batch size around the thresholds of the
launch rules, horizon, sub-steps, observations only) the default launch must have the bits of the same call forced to one and to two
environments per lane. usage: python tools/fuzz_forms.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "exciting-environments_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import torch
from exciting_environments_amd import _native
from helpers import NP_DTYPE, make_env, random_state, to_state

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
MODELS = ["pendulum", "mass_spring_damper", "fluid_tank", "cartpole", "acrobot", "pmsm"]
seen = {}
bad = 0
for case in range(n_cases):
    name = MODELS[rng.integers(len(MODELS))]
    dtype = [torch.float32, torch.float64][rng.integers(2)]
    solver = ["euler", "euler", "rk4", "tsit5"][rng.integers(4)]
    base = 1 << int(rng.choice([16, 17, 17, 18, 18, 19, 20, 20]))
    B = int(base + rng.choice([0, 0, 4, 64, 256, 1000, 1024, 4096, -4, -256, 12, 2]) )
    K = int(rng.integers(2, 14))
    sub = int(rng.choice([1, 1, 1, 2, 3])) if name != "pmsm" else 1
    obs_only = bool(rng.integers(3) == 0)
    env, props, keep, spec = make_env(name, B, dtype, solver=solver)
    env.trajectory_pool = False
    env.store_state_trajectory = not obs_only
    env.sim_ahead_semantics = ["ahead", "step"][rng.integers(2)]
    st = random_state(name, B, NP_DTYPE[dtype], spec, seed=int(rng.integers(1 << 30)))
    acts = env.new_actions_buffer(K)
    acts.copy_(torch.as_tensor(rng.uniform(-1, 1, (B, K, env.action_dim)).astype(NP_DTYPE[dtype]), device=env.device))
    outs, names = {}, {}
    for vec in (0, 1, 2):
        env.launch_opts = _native.launch_opts(envs_per_lane=vec) if vec else None
        outs[vec] = env.vmap_sim_ahead(to_state(env, st), acts, env.tau / sub, env.tau)
        torch.cuda.synchronize()
        names[vec] = _native.last_launch()
    seen[names[0]] = seen.get(names[0], 0) + 1
    ok = True
    for vec in (1, 2):
        ok &= torch.equal(outs[0][0], outs[vec][0])
        for n in env.STATE_FIELDS:
            ok &= torch.equal(getattr(outs[0][2].physical_state, n), getattr(outs[vec][2].physical_state, n))
            if not obs_only:
                ok &= torch.equal(getattr(outs[0][1].physical_state, n), getattr(outs[vec][1].physical_state, n))
    finite = bool(torch.isfinite(outs[0][0]).all())
    print(f"case {case:3d} {name:18s} {str(dtype)[6:]:8s} {solver:6s} B={B:8d} K={K:2d} sub={sub} obs_only={int(obs_only)} {env.sim_ahead_semantics:5s} "
          f"[{names[0]}] {'OK' if ok else 'MISMATCH'}{'' if finite else ' (non-finite values present)'}", flush=True)
    bad += 0 if ok else 1
    del env, outs, acts, st
    torch.cuda.empty_cache()
print("forms seen:", seen)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
