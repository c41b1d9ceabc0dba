"""Rank body of tests/test_gpu_distributed.py (started once per rank by torch.distributed.run, every rank on cuda:0, gloo
backend — RCCL refuses two ranks on one device; the 1-GPU box has no second card). Each rank builds ITS shard of the
product environment with distributed.make_sharded_env (per-env property arrays sliced to the shard), steps it with the HIP
kernels and all-gathers the final observations; rank 0 also runs the un-sharded batch and writes the comparison."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "exciting-environments_amd")]

import numpy as np
import torch
import torch.distributed as dist


def main():
    out_path, B = sys.argv[1], int(sys.argv[2])
    from exciting_environments_amd import EnvironmentRegistry
    from exciting_environments_amd.distributed import ObservationGatherer, make_sharded_env, shard_range

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    K = 37
    rng = np.random.default_rng(5)
    length = torch.as_tensor(rng.uniform(1.0, 3.0, B), dtype=torch.float32, device=dev)  # a per-env [B] property
    theta = torch.as_tensor(rng.uniform(-3, 3, B), dtype=torch.float32, device=dev)
    omega = torch.as_tensor(rng.uniform(-1, 1, B), dtype=torch.float32, device=dev)
    actions = torch.as_tensor(rng.uniform(-1, 1, (B, K, 1)), dtype=torch.float32, device=dev)
    kw = dict(tau=1e-2, dtype=torch.float32, device=dev, static_params={"g": 9.81, "l": length, "m": 1.0})

    env, (lo, hi) = make_sharded_env(EnvironmentRegistry.PENDULUM, B, **kw)
    assert (lo, hi) == shard_range(B, world, rank) and env.batch_size == hi - lo
    _, st = env.vmap_reset()
    st.physical_state.theta, st.physical_state.omega = theta[lo:hi].clone(), omega[lo:hi].clone()
    obs, states, last = env.vmap_sim_ahead(st, actions[lo:hi].contiguous(), env.tau, env.tau)
    gatherer = ObservationGatherer(B)
    full = gatherer.start(obs[:, -1, :])
    gatherer.wait()
    torch.cuda.synchronize()

    result = {"rank": rank, "world": world, "collective": gatherer.collective, "shard": [lo, hi]}
    if rank == 0:
        ref_env = EnvironmentRegistry.PENDULUM.make(batch_size=B, **kw)
        _, s0 = ref_env.vmap_reset()
        s0.physical_state.theta, s0.physical_state.omega = theta.clone(), omega.clone()
        ref_obs, _, _ = ref_env.vmap_sim_ahead(s0, actions, ref_env.tau, ref_env.tau)
        result["gathered_equals_unsharded"] = bool(torch.equal(full, ref_obs[:, -1, :]))
        result["finite"] = bool(torch.isfinite(full).all())
        from exciting_environments_amd import _native
        result["library"] = _native.library_path()
        with open(out_path, "w") as f:
            json.dump(result, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
