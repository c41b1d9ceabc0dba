"""Does the env-major fused kernel care where its output buffers lie (profiles/r03_placement_regions.md)?
One process, the same launch into differently placed buffer sets. usage: python tools/em_placement.py"""
import sys

import torch

sys.path.insert(0, "exciting-environments_amd")
from exciting_environments_amd import EnvironmentRegistry, _native  # noqa: E402

B, K = 1 << 22, 100
dev = torch.device("cuda")
env = EnvironmentRegistry.PMSM.make(batch_size=B, dtype=torch.float32)
obs0, state = env.vmap_reset()
S, OW, N = env.physical_state_dim, env._obs_dim(), K
props, keep = env._props_for(env.env_properties, B)
st_in = [env._t(getattr(state.physical_state, n), (B,)) for n in env.STATE_FIELDS]
control, refs = env._control(state, (B,))
g = torch.Generator(device=dev)
g.manual_seed(1)
actions = (torch.rand((B, K, env.action_dim), generator=g, device=dev) * 2 - 1).contiguous()
last = [torch.empty(B, device=dev) for _ in range(S)]
sem = _native.SEM_AHEAD


def run(obs_buf, st_buf, reps=6):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    for i in range(reps):
        ev[i].record()
        _native.sim_ahead(env.ENV_ID, env._solver.id, env.dtype, B, K, 1, props, control, float(env.tau), float(env.tau), st_in, actions,
                          _native.LAYOUT_ENV_MAJOR, obs_buf, st_buf, _native.LAYOUT_ENV_MAJOR, last, sem, None, env.launch_opts, None)
    ev[reps].record()
    ev[reps].synchronize()
    ts = [ev[i].elapsed_time(ev[i + 1]) for i in range(reps)]
    return min(ts[1:]), sorted(ts[1:])[len(ts[1:]) // 2]


def mk_obs():
    return torch.empty((B, N + 1, OW), device=dev)


def mk_leaves(n=S):
    return [torch.empty((B, N + 1), device=dev) for _ in range(n)]


GiB = 1 << 30
MiB = 1 << 20
leaf_bytes = B * (N + 1) * 4
obs_bytes = leaf_bytes * OW
big = torch.empty(232 * GiB, dtype=torch.uint8, device=dev)


def carve(off, shape):
    n = 4
    for d in shape:
        n *= d
    assert off % 16 == 0 and off + n <= big.numel()
    return big[off: off + n].view(torch.float32).view(*shape)


def case(name, obs_off, leaf_offs):
    ob = carve(obs_off, (B, N + 1, OW))
    lv = [carve(o, (B, N + 1)) for o in leaf_offs]
    print(f"{name}: min %.3f median %.3f" % run(ob, lv), flush=True)


up = lambda x, a: (x + a - 1) // a * a
pitch = up(leaf_bytes, 2 * MiB)
base = up(obs_bytes, 2 * MiB)
case("packed at 0", 0, [base + j * pitch for j in range(S)])
case("obs | 16 GiB | leaves packed", 0, [32 * GiB + j * pitch for j in range(S)])
case("obs | leaves packed at 64", 0, [64 * GiB + j * pitch for j in range(S)])
case("leaves 4 GiB apart from 16", 0, [16 * GiB + j * 4 * GiB for j in range(S)])
case("leaves 8 GiB apart from 16", 0, [16 * GiB + j * 8 * GiB for j in range(S)])
case("leaves 16 GiB apart from 16", 0, [16 * GiB + j * 16 * GiB for j in range(S)])
case("leaves 30 GiB apart from 2, obs at 216", 216 * GiB, [2 * GiB + j * 30 * GiB for j in range(S)])
case("leaves alternate two areas 64 GiB apart", 0, [16 * GiB + (j % 2) * 64 * GiB + (j // 2) * pitch for j in range(S)])
case("leaves in three areas 48 GiB apart", 0, [16 * GiB + (j % 3) * 48 * GiB + (j // 3) * pitch for j in range(S)])
case("obs in the middle (100), leaves packed at 0", 100 * GiB, [j * pitch for j in range(S)])
