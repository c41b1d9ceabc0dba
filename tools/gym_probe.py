#!/usr/bin/env python3
"""Workload for the rocprofv3 passes of the fused gym trajectories (vmap_sim_ahead(..., return_rew_trunc_term=True),
core_env.py:490-531): per model the plain launch and the launch that also writes reward / terminated / truncated, B = 2^22, K = 100,
fp32 Euler, lane-major buffers. Writes the ORDER of its trajectory launches (one JSON line per variant: how many launches, which
kernel form) so that tools/summarize_gym.py can cut the trace's dispatch sequence into variants.
usage: python3 tools/gym_probe.py <log.jsonl> [launches per variant] [settle 0|1]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "exciting-environments_amd")]
import torch

from exciting_environments_amd import EnvironmentRegistry, _native

LOG = sys.argv[1]
LAUNCHES = int(sys.argv[2]) if len(sys.argv) > 2 else 5
SETTLE = int(sys.argv[3]) if len(sys.argv) > 3 else 1
B, K = 1 << 22, 100
CASES = (("PMSM", []), ("PMSM", ["i_d", "i_q"]), ("PENDULUM", ["theta"]), ("PENDULUM", ["theta", "omega"]), ("MASS_SPRING_DAMPER", ["deflection"]),
         ("CART_POLE", ["theta", "deflection"]), ("ACROBOT", ["theta_1", "theta_2"]), ("FLUID_TANK", ["height"]))
log = open(LOG, "w")


for name, control in CASES:
    env = getattr(EnvironmentRegistry, name).make(batch_size=B, device="cuda:0", control_state=list(control))
    _, st = env.vmap_reset()
    for n in control:
        setattr(st.reference, n, torch.zeros(B, device="cuda:0"))
    acts = env.new_actions_buffer(K)
    acts.uniform_(-1, 1)
    for gym in (False, True):
        n_launch = 0
        out = None
        for it in range(40 if SETTLE else 2):
            out = env.vmap_sim_ahead(st, acts, env.tau, env.tau, return_rew_trunc_term=gym)
            n_launch += 1
            torch.cuda.synchronize()
            if it >= 7 and env.trajectory_placement_settled:
                break
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(LAUNCHES):
            out = env.vmap_sim_ahead(st, acts, env.tau, env.tau, return_rew_trunc_term=gym)
            n_launch += 1
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / LAUNCHES * 1e3
        obs = out[0]
        # what the trajectory kernel itself moves per env-step (the control columns of the observations are the fill kernel's)
        O, S, A, nc = env._obs_dim() - len(control), env.physical_state_dim, env.action_dim, len(control)
        TW = _native.truncated_width(env.ENV_ID, nc)
        kernel_b = 4 * (O + S + A) + ((4 + 1 + TW) if gym else 0)
        log.write(json.dumps({"model": name, "control": control, "gym": gym, "launches": n_launch, "timed": LAUNCHES,
                              "form": _native.last_launch(), "ms_per_call": ms,
                              "O": O, "S": S, "A": A, "TW": TW, "kernel_bytes_per_env_step": kernel_b,
                              "call_bytes_per_env_step": kernel_b + 4 * nc, "kernel_written_per_env_step": kernel_b - 4 * A,
                              "env_steps": B * K}) + "\n")
        log.flush()
    del env, out, acts, obs
