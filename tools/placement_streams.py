#!/usr/bin/env python3
"""How many physical regions should the write streams of the small-stream-count workloads (pendulum / mass-spring-damper: two
observation components + two state leaves) be spread over? excenv_stream_pattern over one large arena with the buffers at chosen
offsets (40 GiB apart = certainly different regions where the arena is physically contiguous)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "exciting-environments_amd")]
import numpy as np
import torch
from exciting_environments_amd import _native

dev = torch.device("cuda", 0)
arena = torch.empty(200 << 30, dtype=torch.uint8, device=dev)
base = arena.data_ptr()
GiB = 1 << 30
stream = _native.raw_stream(0)


def timed(fn, n=5):
    fn(); fn()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))


def run(name, isz, B, rows, A, O, S, obs_off, leaf_offs, act_off):
    rb = B * isz
    rd = [base + act_off + c * rb for c in range(A)]
    rd_rs = [A * rb] * A
    wr = [base + obs_off + c * rb for c in range(O)] + [base + o for o in leaf_offs]
    wr_rs = [O * rb] * O + [rb] * S
    t = timed(lambda: _native.stream_pattern(rd, rd_rs, wr, wr_rs, rb, rows, stream))
    print(f"{name:70s} {(A + O + S) * rb * rows / t / 1e6:6.0f} GB/s", flush=True)


for label, isz, B, rows, A, O, S in (("pendulum fp32 B=2^20 K=1000", 4, 1 << 20, 1000, 1, 2, 2), ("msd fp64 B=2^20 K=500", 8, 1 << 20, 500, 1, 2, 2),
                                    ("pmsm fp32 B=2^22 K=100", 4, 1 << 22, 100, 2, 8, 7)):
    leaf = rows * B * isz
    obs = rows * O * B * isz
    act = 150 * GiB
    print("==", label, f"(obs {obs / GiB:.1f} GiB, leaf {leaf / GiB:.2f} GiB)")
    run("all back to back", isz, B, rows, A, O, S, 0, [obs + j * leaf for j in range(S)], act)
    run("obs | leaves (40 GiB apart)", isz, B, rows, A, O, S, 0, [40 * GiB + j * leaf for j in range(S)], act)
    if S == 2:
        run("obs | leaf 0 | leaf 1 (40 GiB apart each)", isz, B, rows, A, O, S, 0, [40 * GiB, 80 * GiB], act)
    else:
        h = (S + 1) // 2
        run("obs | leaves 0..3 | leaves 4..6 (40 GiB apart each)", isz, B, rows, A, O, S, 0,
            [40 * GiB + j * leaf for j in range(h)] + [80 * GiB + j * leaf for j in range(S - h)], act)
        run("obs | leaves in 3 groups (40 GiB apart each)", isz, B, rows, A, O, S, 0,
            [40 * GiB + j * leaf for j in range(3)] + [80 * GiB + j * leaf for j in range(2)] + [120 * GiB + j * leaf for j in range(2)], act)
    run("all back to back again", isz, B, rows, A, O, S, 0, [obs + j * leaf for j in range(S)], act)
