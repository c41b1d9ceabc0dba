import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_ROOT = os.path.join(ROOT, "exciting-environments_amd")
for p in (ROOT, PKG_ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")
ENV_NAMES = ["pendulum", "mass_spring_damper", "cartpole", "acrobot", "fluid_tank", "pmsm"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # fresh checkout: the in-tree HIP library / oracle are build artefacts (git-ignored) — build them once with hipcc / gcc
    lib = os.path.join(PKG_ROOT, "exciting_environments_amd", "lib", "libexcenv_hip.so")
    if not os.path.exists(lib) and "EXCENV_HIP_LIB" not in os.environ:
        import subprocess

        subprocess.run(["make", "-C", os.path.join(PKG_ROOT, "csrc"), "-j", str(min(8, os.cpu_count() or 1))], check=True,
                       capture_output=True)


def load_golden(env):
    """The reference's fixture triplet tests/envs/<env>/data/* (copied verbatim to tests/golden/)."""
    d = os.path.join(GOLDEN, env)
    with open(os.path.join(d, "sim_properties.json")) as f:
        sp = json.load(f)
    actions = np.load(os.path.join(d, "actions.npy"))
    observations = np.load(os.path.join(d, "observations.npy"))
    phys = {k: (v["min"], v["max"]) for k, v in sp["physical_normalizations"].items()}
    act = {k: (v["min"], v["max"]) for k, v in sp["action_normalizations"].items()}
    return dict(params=sp["params"], phys_norm=phys, act_norm=act, tau=sp["tau"], actions=actions,
                observations=observations)


@pytest.fixture(scope="session")
def golden():
    return {e: load_golden(e) for e in ENV_NAMES}


def golden_rtol(env):
    # tolerances of the reference's own test_step_results (rtol 1e-16, PMSM 1e-8; jnp.allclose atol 1e-8)
    return 1e-8 if env == "pmsm" else 1e-16
