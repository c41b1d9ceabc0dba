#!/usr/bin/env python3
"""Acrobot fp32: what does evaluating cos(theta_1 + pi/2), cos(theta_1 + theta_2 + pi/2) through the angle-sum identities
(models.hpp, EXCENV_ACROBOT_ANGLE_SUM) change against the literal forms? Run ON THE GPU BOX once per library
(EXCENV_HIP_LIB=ab_libs/libexcenv_acro_literal.so for the literal build) — each run stores its trajectory over the reference's
acrobot fixture (fp64 reference data, 10 000 Euler steps) and a 64-step random batch, and prints its error against the fixture;
the second run also prints the difference between the two builds.
usage: python tools/acrobot_angle_sum_deviation.py <tag>"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "exciting-environments_amd")]
from conftest import load_golden  # noqa: E402
from helpers import make_env, random_state, to_state  # noqa: E402

tag = sys.argv[1]
g = load_golden("acrobot")
out = {}
for solver in ("euler", "tsit5"):
    env, props, keep, spec = make_env("acrobot", 8, torch.float32, solver=solver)
    env.sim_ahead_semantics = "step"
    obs0 = torch.as_tensor(np.repeat(g["observations"][:1], 8, axis=0), dtype=torch.float32, device=env.device)
    st = env.vmap_generate_state_from_observation(obs0)
    acts = torch.as_tensor(np.repeat(g["actions"][None], 8, axis=0), dtype=torch.float32, device=env.device)
    obs, _, _ = env.vmap_sim_ahead(st, acts, env.tau, env.tau)
    out[f"fixture_{solver}"] = obs[0].cpu().numpy()
    B = 4096
    env2, _, _, spec2 = make_env("acrobot", B, torch.float32, solver=solver)
    s2 = to_state(env2, random_state("acrobot", B, np.float32, spec2, seed=7))
    a2 = torch.as_tensor(np.random.default_rng(8).uniform(-1, 1, (B, 64, 1)).astype(np.float32), device=env2.device)
    o2, _, _ = env2.vmap_sim_ahead(s2, a2, env2.tau, env2.tau)
    out[f"random_{solver}"] = o2.cpu().numpy()


def circ(d):
    d = np.abs(d)
    for c in (0, 1):
        d[..., c] = np.minimum(d[..., c], np.abs(2 - d[..., c]))
    return d


e = circ(out["fixture_euler"].astype(np.float64) - g["observations"]).max(axis=1)
print(tag, "fp32 Euler vs the fp64 fixture, running max at rows 100 / 1000 / 10000:", [float(np.maximum.accumulate(e)[r]) for r in (100, 1000, 10000)])
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
path = os.path.join(ROOT, "gpurun_out", "acrobot_dev_{}.npz")
np.savez(path.format(tag), **out)
other = [t for t in ("sum", "literal") if t != tag and os.path.exists(path.format(t))]
if other:
    o = np.load(path.format(other[0]))
    for k in out:
        d = circ(out[k].astype(np.float64) - o[k])
        rows = d.reshape(-1, d.shape[-2], d.shape[-1]).max(axis=(0, 2))
        run = np.maximum.accumulate(rows)
        print(f"{tag} vs {other[0]} {k}: max |difference| at rows 1 / 10 / 64 / last:", [float(run[min(r, len(run) - 1)]) for r in (1, 10, 64, len(run) - 1)])
