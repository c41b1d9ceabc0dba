#!/usr/bin/env python3
"""Same-process A/B of builds of libexcenv_hip.so (tools/build_variant.sh -> ab_libs/): the headline call runs into the SAME pooled
output sets with every library in turn, so the placement of the buffers — which moves this kernel by more than most code changes
do — is common to all of them. usage (GPU box): python tools/ab_same_buffers.py [--workload pmsm_euler_f32] [--rounds 3] name ..."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "exciting-environments_amd"))
import bench  # noqa: E402
from exciting_environments_amd import _native  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="pmsm_euler_f32")
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--calls", type=int, default=20)
ap.add_argument("--action-layout", default="lane_major")
ap.add_argument("--obs-only", action="store_true")
ap.add_argument("--traj-layout", default="lane_major", help="env_major: the reference's row-major output arrays (register-ring kernel)")
ap.add_argument("--batch", type=int, default=0, help="log2 of the batch size (default: the workload's)")
ap.add_argument("--gym", action="store_true", help="with the fused reward / terminated / truncated trajectories")
ap.add_argument("names", nargs="*")
a = ap.parse_args()

base_path = _native.library_path()
# a name of the form vec=N is not another library but the in-tree one with N environments per lane (launch option); env:NAME=VALUE is
# the in-tree library with that environment variable set for its launches (switches the library reads per call)
inline = lambda n: n.startswith("vec=") or n.startswith("env:")
libs = [("in-tree", base_path)] + [(n, base_path if inline(n) else os.path.join(ROOT, "ab_libs", f"libexcenv_{n}.so")) for n in a.names]
env_names = {n[4:].split("=", 1)[0] for n in a.names if n.startswith("env:")}


def use(path):
    _native._lib = None
    _native._LIB_PATH = path
    _native.lib()


use(base_path)
class Args:
    pass


ba = Args()
ba.workload, ba.batch, ba.chunk, ba.semantics, ba.traj_layout, ba.action_layout = a.workload, (1 << a.batch) if a.batch else 0, 0, "ahead", a.traj_layout, a.action_layout
ba.path, ba.obs_only, ba.no_workspace, ba.no_fused, ba.no_pool = "sim_ahead", a.obs_only, False, False, False
env, state, actions, B, Kc, *_ = bench.build_env(ba, torch.device("cuda", 0), 0)


def call():
    return env.vmap_sim_ahead(state, actions, env.tau, env.tau, return_rew_trunc_term=a.gym)


for _ in range(40):
    out = call()
    del out
    if env.trajectory_placement_settled:
        break
torch.cuda.synchronize()
env.trajectory_placement = "off"  # no replacement searches from here on: the sets' real-launch feedback would compare times of different libraries
print("placement:", {k: v for k, v in (env.last_placement or {}).items() if k in ("what", "pattern_over_fill", "candidate_pattern_over_fill")}, flush=True)
res = {n: [] for n, _ in libs}
for r in range(a.rounds):
    for n, p in libs:
        use(p)
        for e in env_names:
            os.environ.pop(e, None)
        if n.startswith("env:"):
            os.environ[n[4:].split("=", 1)[0]] = n[4:].split("=", 1)[1]
        env.launch_opts = _native.launch_opts(envs_per_lane=int(n[4:])) if n.startswith("vec=") else None
        for _ in range(4):
            out = call()
            del out
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.calls + 1)]
        ev[0].record()
        for i in range(a.calls):
            out = call()
            del out
            ev[i + 1].record()
        torch.cuda.synchronize()
        ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(a.calls)]
        res[n].append(ms)
        even, odd = ms[0::2], ms[1::2]
        print(f"round {r} {n:18s} launch {_native.last_launch():28s} mean {sum(ms) / len(ms):.3f}  sets {sum(even) / len(even):.3f} / {sum(odd) / len(odd):.3f}  min {min(ms):.3f}",
              flush=True)
print()
for n, _ in libs:
    allms = [x for ms in res[n] for x in ms]
    print(f"{n:18s} mean {sum(allms) / len(allms):.3f} ms  min {min(allms):.3f}")
