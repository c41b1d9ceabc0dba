#!/usr/bin/env python3
"""Randomised cross-check of the launch forms that tools/fuzz_forms.py does not reach (GPU box): for random (model, dtype, solver,
control_state subset, batch size around the thresholds of the launch rules, horizon, semantics)
  * the reward / terminated / truncated trajectories of the default launch (the wide lean kernel where its preconditions hold) must
    have the bits of the same call with one environment per lane (the general instantiation), and
  * the same call with a plain row-major actions[B, K, A] tensor must have the bits of the call with the lane-major buffer.
usage: python tools/fuzz_gym.py [cases] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "exciting-environments_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import torch

from exciting_environments_amd import _native
from helpers import NP_DTYPE, make_env, random_state, to_state

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
MODELS = ["pendulum", "mass_spring_damper", "fluid_tank", "cartpole", "acrobot", "pmsm"]
CONTROLLABLE = {"pmsm": ["i_d", "i_q", "torque"]}
seen = {}
bad = 0
for case in range(n_cases):
    name = MODELS[rng.integers(len(MODELS))]
    dtype = [torch.float32, torch.float64][rng.integers(2)]
    solver = ["euler", "euler", "rk4", "tsit5"][rng.integers(4)]
    base = 1 << int(rng.choice([15, 16, 17, 18, 19, 20]))
    B = int(base + rng.choice([0, 0, 0, 256, 1024, 4096, 4, -4, 64, 1000]))
    K = int(rng.choice([4, 8, 12, 16, 20]))
    env0, _, _, spec = make_env(name, 4, dtype, solver=solver)
    fields = CONTROLLABLE.get(name, list(env0.STATE_FIELDS))
    del env0
    nctl = int(rng.integers(0, len(fields) + 1))
    control = [fields[i] for i in sorted(rng.choice(len(fields), nctl, replace=False))]
    env, props, keep, spec = make_env(name, B, dtype, solver=solver, control_state=control)
    env.trajectory_pool = False
    env.sim_ahead_semantics = ["ahead", "step"][rng.integers(2)]
    st = random_state(name, B, NP_DTYPE[dtype], spec, seed=int(rng.integers(1 << 30)))
    st[0] = (st[0] * 1.2).astype(NP_DTYPE[dtype])  # some states outside the normalisation box: truncated flags of both kinds
    refs = {}
    for n in control:
        lo, hi = spec["phys_norm"][n]
        lo, hi = float(np.min(lo)), float(np.max(hi))
        refs[n] = rng.uniform(1.2 * lo if lo < 0 else lo, 1.2 * hi, B).astype(NP_DTYPE[dtype])
    plain = torch.as_tensor(rng.uniform(-1, 1, (B, K, env.action_dim)).astype(NP_DTYPE[dtype]), device=env.device)
    lane = env.new_actions_buffer(K)
    lane.copy_(plain)
    state = lambda: to_state(env, st, reference=refs if control else None)
    outs, names = {}, {}
    for tag, acts, vec, gym in (("gym", lane, 0, True), ("gym1", lane, 1, True), ("lane", lane, 0, False), ("rows", plain, 0, False)):
        env.launch_opts = _native.launch_opts(envs_per_lane=vec) if vec else None
        outs[tag] = env.vmap_sim_ahead(state(), acts, env.tau, env.tau, return_rew_trunc_term=gym)
        torch.cuda.synchronize()
        names[tag] = _native.last_launch()
    seen[names["gym"]] = seen.get(names["gym"], 0) + 1
    seen[names["rows"]] = seen.get(names["rows"], 0) + 1

    def same(a, b, with_gym):
        ok = torch.equal(a[0], b[0])
        for n in env.STATE_FIELDS:
            ok &= torch.equal(getattr(a[1].physical_state, n), getattr(b[1].physical_state, n))
            ok &= torch.equal(getattr(a[2].physical_state, n), getattr(b[2].physical_state, n))
        if with_gym:
            for k in (3, 4, 5):
                ok &= a[k].shape == b[k].shape and torch.equal(a[k], b[k])
        return bool(ok)

    ok_gym = same(outs["gym"], outs["gym1"], True)
    ok_rows = same(outs["rows"], outs["lane"], False) and same(outs["gym"], outs["lane"], False)
    flags = int(outs["gym"][4].sum())
    print(f"case {case:3d} {name:18s} {str(dtype)[6:]:8s} {solver:6s} B={B:8d} K={K:2d} control={control} {env.sim_ahead_semantics:5s} "
          f"[{names['gym']} | {names['rows']}] gym {'OK' if ok_gym else 'MISMATCH'} rows {'OK' if ok_rows else 'MISMATCH'} "
          f"(truncated flags set: {flags})", flush=True)
    bad += (0 if ok_gym else 1) + (0 if ok_rows else 1)
    del env, outs, plain, lane, st
    torch.cuda.empty_cache()
print("forms seen:", seen)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
