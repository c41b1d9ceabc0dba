"""Seeded runs of the two launch-form fuzzers (tools/fuzz_forms.py, tools/fuzz_gym.py) inside the GPU suite: random model / dtype /
solver / batch size around the thresholds of the launch rules / horizon / control_state subsets — the default launch must have the
bits of the forced narrow forms, the gym trajectories of the wide lean kernel those of the general instantiation, and a plain
row-major actions tensor those of the lane-major buffer. ``-m gpu``."""
import os
import runpy
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tool,cases,seed", [("fuzz_forms.py", 24, 5), ("fuzz_gym.py", 32, 6)])
def test_seeded_fuzz_of_the_launch_forms(tool, cases, seed, monkeypatch):
    monkeypatch.setattr(sys, "argv", [tool, str(cases), str(seed)])
    with pytest.raises(SystemExit) as e:
        runpy.run_path(os.path.join(ROOT, "tools", tool), run_name="__main__")
    assert e.value.code == 0
