#!/usr/bin/env python3
"""What do the fused reward / terminated / truncated trajectories (vmap_sim_ahead(..., return_rew_trunc_term=True),
core_env.py:490-531) cost on top of the plain trajectory launch? B = 2^22, K = 100, lane-major buffers, fp32 Euler.
Per model three launches: plain, with the gym outputs from the wide lean kernel (LGYM, all six models since round 4), and the same with
one environment per lane (the general instantiation). tools/profile_gym.sh is the profiled version of the first two."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "exciting-environments_amd")]
import torch

from exciting_environments_amd import EnvironmentRegistry, _native

B, K = 1 << 22, 100
CASES = (("PMSM", []), ("PMSM", ["i_d", "i_q"]), ("PENDULUM", ["theta"]), ("PENDULUM", ["theta", "omega"]), ("MASS_SPRING_DAMPER", ["deflection"]),
         ("CART_POLE", ["theta", "deflection"]), ("ACROBOT", ["theta_1", "theta_2"]), ("FLUID_TANK", ["height"]))
if len(sys.argv) > 1:
    CASES = tuple(c for c in CASES if c[0] in sys.argv[1:])
for name, control in CASES:
    env = getattr(EnvironmentRegistry, name).make(batch_size=B, device="cuda:0", control_state=list(control))
    _, st = env.vmap_reset()
    for n in control:
        setattr(st.reference, n, torch.zeros(B, device="cuda:0"))
    acts = env.new_actions_buffer(K)
    acts.uniform_(-1, 1)
    for gym, vec in ((False, 0), (True, 0), (True, 1)):
        env.launch_opts = _native.launch_opts(envs_per_lane=vec) if vec else None
        out = None
        for it in range(40):
            out = env.vmap_sim_ahead(st, acts, env.tau, env.tau, return_rew_trunc_term=gym)
            torch.cuda.synchronize()
            if it >= 7 and env.trajectory_placement_settled:
                break
        t0 = time.perf_counter()
        for _ in range(5):
            out = env.vmap_sim_ahead(st, acts, env.tau, env.tau, return_rew_trunc_term=gym)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        print(f"{name:9s} control {str(control):16s} {'gym' if gym else 'plain':5s} {'(one env per lane)' if vec else '':18s} [{_native.last_launch()}] {ms:.3f} ms", flush=True)
    del env, out, acts
