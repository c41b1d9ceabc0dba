#!/usr/bin/env python3
"""What do control_state columns / per-env property arrays / fused gym trajectories cost on vmap_sim_ahead at a large batch?
(they select the GENERAL instantiation: one environment per lane). PMSM Euler fp32, B = 2^22, K = 100, lane-major buffers."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "exciting-environments_amd")]
import torch

from exciting_environments_amd import EnvironmentRegistry, _native

B, K = 1 << 22, 100
dev = "cuda:0"


def run(tag, vec=0, **kw):
    env = EnvironmentRegistry.PMSM.make(batch_size=B, device=dev, **kw)
    if vec:
        env.launch_opts = _native.launch_opts(envs_per_lane=vec)
    _, st = env.vmap_reset()
    acts = env.new_actions_buffer(K)
    acts.uniform_(-1, 1)
    O = env._obs_dim()
    out = None
    for it in range(40):  # the first calls create, place and compare the pooled output sets (core_env.py), outside the timing
        out = env.vmap_sim_ahead(st, acts, env.tau, env.tau)
        torch.cuda.synchronize()
        if it >= 7 and env.trajectory_placement_settled:
            break
    t0 = time.perf_counter()
    for _ in range(5):
        out = env.vmap_sim_ahead(st, acts, env.tau, env.tau)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    nbytes = 4 * (2 + O + 7) * B * K
    tag = f"{tag} [{_native.last_launch()}]"
    print(f"{tag:86s} {ms:7.3f} ms  {B * K / ms / 1e-3:.3e} env-steps/s  {nbytes / ms / 1e6:6.0f} GB/s ({nbytes / ms / 1e6 / 8000:.3f} of 8 TB/s; {4 * (2 + O + 7)} B/env-step)")
    del out, env, acts


run("lean kernel (auto: 4 environments per lane)")
run("lean kernel forced to 1 environment per lane", vec=1)
run("control_state = [i_d, i_q] (GENERAL)", control_state=["i_d", "i_q"])
pn = EnvironmentRegistry.PMSM.make(batch_size=8, device="cpu").env_properties.static_params
RS = dict(static_params={**{k: getattr(pn, k) for k in pn.__dataclass_fields__},
                                                   "r_s": torch.full((B,), float(pn.r_s)).numpy()})
run("per-env r_s array (GENERAL, auto)", **RS)
run("per-env r_s array (GENERAL, one environment per lane)", vec=1, **RS)
run("per-env r_s array + control_state = [i_d, i_q]", control_state=["i_d", "i_q"], **RS)
