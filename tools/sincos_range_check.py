#!/usr/bin/env python3
"""ulp error of the in-kernel fp32 sin/cos over the fast-path range and beyond (run on the GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "exciting-environments_amd")]
import numpy as np, torch
from exciting_environments_amd import _native
for lo, hi in ((0, 4), (4, 50), (50, 128), (128, 512), (512, 1024), (1024, 65536), (65536, 1e6)):
    x = torch.empty(4_000_000, dtype=torch.float32).uniform_(lo, hi)
    x = torch.cat([x, -x]).cuda()
    for which, fn in ((0, np.sin), (1, np.cos)):
        got = _native.probe_math(which, x).cpu().numpy().astype(np.float64)
        want = fn(x.cpu().numpy().astype(np.float64))
        ulp = np.spacing(np.abs(want).astype(np.float32)).astype(np.float64)
        err = np.abs(got - want) / ulp
        print(f"|x| in [{lo},{hi}) {'sin' if which == 0 else 'cos'}: max ulp {err.max():.2f}, max abs {np.abs(got - want).max():.3e}, p99.99 ulp {np.quantile(err, 0.9999):.2f}")
