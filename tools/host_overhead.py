#!/usr/bin/env python3
"""Where does the host time of one vmap_step go at small batch sizes? (cProfile over 3000 steps, B = 1024)"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "exciting-environments_amd")]
import torch

from exciting_environments_amd import EnvironmentRegistry

for name in ("PENDULUM", "PMSM"):
    env = getattr(EnvironmentRegistry, name).make(batch_size=1024, device="cuda:0")
    _, state = env.vmap_reset()
    act = torch.zeros((1024, env.action_dim), device="cuda:0")
    for _ in range(50):
        obs, state = env.vmap_step(state, act)
    torch.cuda.synchronize()
    n = 3000
    best = float("inf")
    for _ in range(3):  # the first timed loop of a process runs ~2x slower (clocks / allocator warm-up): report the best of 3
        t0 = time.perf_counter()
        for _ in range(n):
            obs, state = env.vmap_step(state, act)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n)
    print(f"{name}: {best * 1e6:.1f} us per vmap_step (B=1024, eager, best of 3 x {n})")
    # HIP graph replay of 16 chained steps
    g = torch.cuda.CUDAGraph()
    s_in = state
    with torch.cuda.graph(g):
        s = s_in
        for _ in range(16):
            o, s = env.vmap_step(s, act)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        g.replay()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / (200 * 16) * 1e6:.1f} us per vmap_step inside a 16-step HIP graph")
    for graph in (False, True):
        stp = env.make_stepper(n_steps=16, graph=graph)
        stp.reset(state)
        for _ in range(5):
            stp.run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(300):
            stp.run()
        torch.cuda.synchronize()
        print(f"{name}: {(time.perf_counter() - t0) / (300 * 16) * 1e6:.2f} us per step, Stepper(n_steps=16, graph={graph})")
    for K in (10, 100):  # vmap_sim_ahead: host + launch per call (the kernel itself takes ~1 us per solver step at this size)
        acts = torch.zeros((1024, K, env.action_dim), device="cuda:0")
        lm = env.new_actions_buffer(K)
        for a, tag in ((acts, "row-major actions"), (lm, "lane-major actions")):
            for _ in range(20):
                env.vmap_sim_ahead(state, a, env.tau, env.tau)
            torch.cuda.synchronize()
            best = float("inf")
            for _ in range(3):
                t0 = time.perf_counter()
                for _ in range(500):
                    out = env.vmap_sim_ahead(state, a, env.tau, env.tau)
                t_host = (time.perf_counter() - t0) / 500
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / 500)
            print(f"{name}: vmap_sim_ahead K={K} ({tag}): {best * 1e6:.1f} us per call end to end, {t_host * 1e6:.1f} us host time to enqueue")
    from exciting_environments_amd import GymWrapper
    gw = GymWrapper(env)
    for _ in range(50):
        gw.step(act)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        gw.step(act)
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / n * 1e6:.1f} us per GymWrapper.step (fused gym kernel, B=1024)")
    from exciting_environments_amd import random as jr
    ctl = ["theta"] if name == "PENDULUM" else ["i_d", "i_q"]
    env_c = getattr(EnvironmentRegistry, name).make(batch_size=1024, device="cuda:0", control_state=ctl)
    gw = GymWrapper(env_c, control_state=ctl)
    gw.reset(rng_env=jr.split(jr.PRNGKey(1), 1024).cuda(), rng_ref=jr.PRNGKey(2).cuda())
    for _ in range(50):
        gw.step(act)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        gw.step(act)
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / n * 1e6:.1f} us per GymWrapper.step with the key-stream reference generator armed")

env = EnvironmentRegistry.PENDULUM.make(batch_size=1024, device="cuda:0")
_, state = env.vmap_reset()
act = torch.zeros((1024, 1), device="cuda:0")
pr = cProfile.Profile()
pr.enable()
for _ in range(3000):
    obs, state = env.vmap_step(state, act)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
