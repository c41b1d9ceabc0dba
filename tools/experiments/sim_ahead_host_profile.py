#!/usr/bin/env python3
"""cProfile of the host side of vmap_sim_ahead at an RL-sized call (PMSM, B = 1024, K = 10, lane-major actions): where the ~28 us per
call go (GPU box). Round 5: the C call incl. the launch 4 us, two unbind() 4 us, the trajectory glue 4 us, seven state leaves through
_t() 3 us, views 2 us — nothing dominant."""
import cProfile, pstats, sys, os, io
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "exciting-environments_amd")]
import torch
from exciting_environments_amd import EnvironmentRegistry
env = EnvironmentRegistry.PMSM.make(batch_size=1024, device="cuda:0")
_, st = env.vmap_reset()
acts = env.new_actions_buffer(10); acts.uniform_(-1, 1)
for _ in range(200): out = env.vmap_sim_ahead(st, acts, env.tau, env.tau)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(3000): out = env.vmap_sim_ahead(st, acts, env.tau, env.tau)
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:4500])
