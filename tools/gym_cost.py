import sys, time, torch
sys.path[:0] = ["/root/repo", "/root/repo/exciting-environments_amd"]
from exciting_environments_amd import EnvironmentRegistry
B, K = 1 << 22, 100
for name in ("PMSM", "PENDULUM"):
    env = getattr(EnvironmentRegistry, name).make(batch_size=B, device="cuda:0")
    _, st = env.vmap_reset()
    acts = env.new_actions_buffer(K); acts.uniform_(-1, 1)
    for gym in (False, True):
        out = None
        for _ in range(8):
            out = env.vmap_sim_ahead(st, acts, env.tau, env.tau, return_rew_trunc_term=gym)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            out = env.vmap_sim_ahead(st, acts, env.tau, env.tau, return_rew_trunc_term=gym)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
        print(name, "gym" if gym else "plain", f"{ms:.3f} ms")
    del env, out
