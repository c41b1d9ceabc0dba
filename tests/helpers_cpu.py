"""Seeded PMSM problem shared by the parent and the spawned ranks of the gloo test."""
import numpy as np

from conftest import load_golden


def pmsm_problem(B, K):
    g = load_golden("pmsm")
    spec = dict(params=g["params"], phys_norm=g["phys_norm"], act_norm=g["act_norm"], tau=g["tau"])
    rng = np.random.default_rng(99)
    z = np.zeros(B, dtype=np.float32)
    st = [z.copy(), z.copy(), rng.uniform(-3, 3, B).astype(np.float32), np.full(B, -125, dtype=np.float32), z.copy(),
          z.copy(), rng.uniform(0, 600, B).astype(np.float32)]
    acts = rng.uniform(-1, 1, (B, K, 2)).astype(np.float32)
    return st, acts, spec
