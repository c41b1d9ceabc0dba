#!/usr/bin/env python3
"""Where do the saturated-PMSM kernels and the oracle first separate on the reference's motor tables?
Runs ON THE GPU BOX: python tools/f3_divergence.py [BRUSA|SEW] [euler|tsit5] — same inputs as
tests/test_gpu_saturated.py::test_reference_motor_tables_step_and_sim_ahead_match_oracle, fp64.
 1. trajectory (K exact steps): per saved row the largest |kernel - oracle| of every state leaf, the environment it occurs in;
 2. ONE step from the oracle's own state at the row before the first separation (so nothing has accumulated): kernel vs oracle
    for Euler and Tsit5 — a one-step difference shows the operation, none shows amplification of last-place differences;
 3. the amplification itself: the oracle's trajectory from inputs moved by one ulp."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "exciting-environments_amd"), os.path.join(ROOT, "tests")]
import exciting_environments_amd as ex  # noqa: E402
import oracle  # noqa: E402
from exciting_environments_amd import EnvironmentRegistry, MotorVariant  # noqa: E402
from helpers import random_state, to_state  # noqa: E402

motor = sys.argv[1] if len(sys.argv) > 1 else "BRUSA"
solver = sys.argv[2] if len(sys.argv) > 2 else "tsit5"
B, K, dtype = 2048, 40, torch.float64
solv = {"euler": ex.Euler(), "tsit5": ex.Tsit5()}
env = EnvironmentRegistry.PMSM.make(batch_size=B, saturated=True, motor_variant=MotorVariant[motor], solver=solv[solver], dtype=dtype, device="cuda")
ep = env.env_properties
params = {n: getattr(ep.static_params, n) for n in env.PARAM_FIELDS}
pn = {n: (getattr(ep.physical_normalizations, n).min, getattr(ep.physical_normalizations, n).max) for n in env.STATE_FIELDS}
an = {n: (getattr(ep.action_normalizations, n).min, getattr(ep.action_normalizations, n).max) for n in env.ACTION_FIELDS}
props, keep = oracle.make_props("pmsm", params, pn, an, np.float64, B, pmsm_lut=env._lut_host)
spec = dict(params=params, phys_norm=pn, act_norm=an, tau=env.tau)
st = random_state("pmsm", B, np.float64, spec, seed=431)
st[6] = st[6] * (0.25 if motor == "BRUSA" else 0.1)
st[3][::7] = pn["i_d"][0] * 1.6
st[4][3::11] = pn["i_q"][1] * 1.3
acts = np.random.default_rng(432).uniform(-1, 1, (B, K, 2))
names = env.STATE_FIELDS
print(f"== {motor} {solver} fp64, B = {B}, K = {K}")
for sem, osem in (("step", oracle.SEM_STEP), ("ahead", oracle.SEM_AHEAD)):
    env.sim_ahead_semantics = sem
    o, s, l = env.vmap_sim_ahead(to_state(env, st), torch.as_tensor(acts, device="cuda"), env.tau, env.tau)
    o_ref, s_ref, l_ref = oracle.sim_ahead("pmsm", solver, st, acts, props, env.tau, semantics=osem)
    got = [getattr(s.physical_state, n).cpu().numpy() for n in names]
    scale = [max(abs(pn[n][0]), abs(pn[n][1])) for n in names]
    err = np.stack([np.abs(g - r) / sc for g, r, sc in zip(got, s_ref, scale)])  # [S, B, K+1], in units of each leaf's full scale
    per_row = err.max(axis=1)  # [S, K+1]
    print(f"-- semantics {sem}: per row, largest |kernel - oracle| / full scale over the batch (leaves {names})")
    for r in range(min(K + 1, 12)):
        print(f"   row {r:2d}: " + "  ".join(f"{per_row[j, r]:.2e}" for j in range(len(names))))
    first = next((r for r in range(K + 1) if per_row[:, r].max() > 0), None)
    print(f"   first row with any difference: {first}; rows with max > 1e-12: {[r for r in range(K + 1) if per_row[:, r].max() > 1e-12][:5]}")
    if sem == "step" and first is not None:
        r0 = max(first - 1, 0)
        j, b = np.unravel_index(np.argmax(err[:, :, first]), err[:, :, first].shape)
        print(f"   worst at row {first}: leaf {names[j]}, environment {b}: kernel {got[j][b, first]!r} oracle {s_ref[j][b, first]!r}")
        print(f"   its state at row {r0}: " + ", ".join(f"{n}={s_ref[q][b, r0]!r}" for q, n in enumerate(names)) + f"; action {acts[b, r0].tolist()}")
        # one step from the ORACLE's state at row r0 (all environments), both solvers
        st_r0 = [x[:, r0].copy() for x in s_ref]
        for sv in ("euler", "tsit5"):
            e1 = EnvironmentRegistry.PMSM.make(batch_size=B, saturated=True, motor_variant=MotorVariant[motor], solver=solv[sv], dtype=dtype, device="cuda")
            ob, new = e1.vmap_step(to_state(e1, st_r0), torch.as_tensor(acts[:, r0], device="cuda"))
            o1, s1 = oracle.step("pmsm", sv, st_r0, acts[:, r0], props, env.tau)
            d = [np.abs(getattr(new.physical_state, n).cpu().numpy() - s1[q]) / scale[q] for q, n in enumerate(names)]
            ulp = [np.abs(getattr(new.physical_state, n).cpu().numpy() - s1[q]) / np.maximum(np.spacing(np.abs(s1[q])), 1e-300) for q, n in enumerate(names)]
            print(f"   ONE {sv} step from the oracle's row-{r0} state: max |kernel - oracle| / full scale per leaf: "
                  + "  ".join(f"{n}={x.max():.2e} ({u.max():.0f} ulp)" for n, x, u in zip(names, d, ulp)))
            nz = [(n, int((x > 0).sum())) for n, x in zip(names, d)]
            print(f"      environments that differ at all, per leaf: {nz}")
    # amplification: the oracle against itself from inputs one ulp away
    o_p, s_p, _ = oracle.sim_ahead("pmsm", solver, [np.nextafter(x, np.inf) for x in st], acts, props, env.tau, semantics=osem)
    amp = np.stack([np.abs(a - r) / sc for a, r, sc in zip(s_p, s_ref, scale)]).max(axis=1)
    print("   oracle vs oracle with every input moved by one ulp, per row (max over leaves): "
          + " ".join(f"{amp[:, r].max():.1e}" for r in range(min(K + 1, 12))))
    print("   kernel vs oracle, per row (max over leaves):                                 "
          + " ".join(f"{per_row[:, r].max():.1e}" for r in range(min(K + 1, 12))))
