"""Static guards for the hand-written synchronisation of the trajectory kernels (tools/isa_guards.py), run on the disassembly of
the BUILT library in the CPU suite — and proven to bite: the same guards must reject three deliberately broken builds
(`-DEXCENV_FAULT=1/2/4`, one translation unit each, compiled here with hipcc and never linked).

Why these exist: the row-major action windows are filled by LDS-direct loads hidden from the compiler's wait insertion, the one wait
in front of a window's first read is counted by hand, the rows of the one-environment kernels pass a raw `s_barrier`, and M0 is
written inside an asm string. A miscount, a missing `lgkmcnt(0)`, a missing wait state are all timing-dependent on the GPU —
bit-equality tests can pass by luck (round 4 shipped `ds_write ...; s_barrier` until a look at the disassembly found it)."""
import importlib.util
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "exciting-environments_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


def _tool():
    spec = importlib.util.spec_from_file_location("isa_guards", os.path.join(ROOT, "tools", "isa_guards.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if not os.path.exists(mod.OBJDUMP):
        pytest.skip("llvm-objdump not available")
    return mod


@pytest.fixture(scope="module")
def shipped():
    """Every kernel of the built library that contains an LDS-direct load or a barrier, checked once."""
    mod = _tool()
    return mod, mod.run(mod.LIB)


def test_action_window_fills_are_complete_before_their_first_read(shipped):
    """(a) Every `global_load_lds_dwordx4` of every AEM / register-ring instantiation is proven complete — by a vmcnt wait with at
    least as many vector-memory instructions behind the fill as the wait leaves outstanding — on every path to the first LDS read."""
    mod, r = shipped
    assert r["dma_kernels"] >= 200, r["dma_kernels"]  # AEM instantiations of six models + 72 register-ring kernels
    bad = {k: v for k, v in r["problems"].items() if any("fill" in m for m in v)}
    assert not bad, list(bad.items())[:3]
    counted = {(vm, cnt) for (vm, cnt, marked) in r["dma_sites"] if marked and vm > 0}
    # the counted waits are exact, not merely sufficient: vmcnt(N) with exactly N stores behind the fill (N = O + S or O per model;
    # PMSM 15 / 8, cart-pole and acrobot 8 / 4, pendulum and mass-spring-damper 4 / 2, tank 2 / 1)
    assert counted and all(vm == cnt for vm, cnt in counted), sorted(counted)
    assert {15, 8, 4, 2, 1} <= {vm for vm, _ in counted}, sorted(counted)


def test_no_barrier_behind_an_outstanding_lds_write(shipped):
    """(b) `s_barrier` is never reached with a `ds_write` of the same wave outstanding (gfx950 does not wait implicitly)."""
    mod, r = shipped
    assert r["barrier_kernels"] >= 100, r["barrier_kernels"]
    bad = {k: v for k, v in r["problems"].items() if any("s_barrier" in m for m in v)}
    assert not bad, list(bad.items())[:3]


def test_m0_is_written_one_wait_state_before_every_lds_direct_load(shipped):
    """(c) `s_mov_b32 m0, sN; s_nop 0; global_load_lds_dwordx4` — nothing of the compiler's in between — and the statement declares
    M0 as clobbered in both kernels."""
    mod, r = shipped
    bad = {k: v for k, v in r["problems"].items() if any("s_mov_b32 m0" in m for m in v)}
    assert not bad, list(bad.items())[:3]
    for name in ("kernels.hpp", "kernels_emr.hpp"):
        src = open(os.path.join(CSRC, name)).read()
        stmts = re.findall(r'asm volatile\("s_mov_b32 m0[^;]*;', src)
        assert stmts, name
        for s in stmts:
            assert '"m0"' in s, (name, s)
            assert "s_nop 0" in s or "EXCENV_FAULT" in src[max(0, src.index(s) - 200):src.index(s)], (name, s)


def test_the_guards_reject_deliberately_broken_builds(tmp_path):
    """Self-test: the fluid-tank translation unit (smallest model; it instantiates the AEM, row-through-LDS and register-ring forms
    like every other) compiled three times with one fault each. Each guard must fail on its fault and only on it."""
    mod = _tool()
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    flags = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-Wno-unused-function"]
    procs = {}
    for fault in (1, 2, 4):
        obj = str(tmp_path / f"env_tank_fault{fault}.o")
        procs[fault] = (obj, subprocess.Popen([HIPCC, *flags, f"-DEXCENV_FAULT={fault}", "-c", "env_tank.hip", "-o", obj], cwd=CSRC,
                                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    expect = {1: "does not prove it complete", 2: "s_barrier", 4: "s_mov_b32 m0"}
    for fault, (obj, p) in procs.items():
        out, _ = p.communicate(timeout=900)
        assert p.returncode == 0, out.decode()[-2000:]
        r = mod.run(obj)
        msgs = [m for v in r["problems"].values() for m in v]
        assert msgs, f"fault {fault} was not detected"
        assert all(expect[fault] in m for m in msgs), (fault, msgs[:3])
    # and the product's own static_assert keeps such an object out of the library
    assert "static_assert(EXCENV_FAULT == 0" in open(os.path.join(CSRC, "excenv_api.hip")).read()
