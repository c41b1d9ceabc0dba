#!/usr/bin/env python3
"""Static guards for what the trajectory kernels synchronise BY HAND (runs in the build container, no GPU): the device code of the
built library (or of one object file) is disassembled and three properties are checked per kernel.

  dma   An LDS-direct load (`global_load_lds_dwordx4`, the row-major action windows of kernels.hpp `AEM` / kernels_emr.hpp) is
        invisible to the compiler's wait insertion; the wait in front of the first LDS read of the window is written by hand as
        `s_waitcnt vmcnt(N) expcnt(6)` (expcnt(6) never blocks in these kernels — no exports, no GDS — and marks the hand-written
        waits in the disassembly). vmcnt retires in issue order, so `vmcnt(N)` proves the fill complete iff AT LEAST N vector-memory
        instructions were issued behind it: fewer (a store the compiler merged or skipped on some path) and the wait returns while
        the fill may still be in flight -> stale LDS, timing-dependent. More than N is safe but waits for trajectory stores.
        Check: on every path from a fill to the first `ds_read*`, a vmcnt wait with N <= (vector-memory instructions since the fill)
        is passed first. The skip edge of a wave-uniform `if (first piece read behind a fill) wait;` is not followed (which step
        that is is source-level logic, covered by the bit-equality tests; what this guard covers is the COUNT) — but a fill that
        reaches such a wait with NOTHING behind it is reported: that is how the prologue hole of round 5 was found (a window
        re-requested before the loop, the loop's first read following without a saved row in between). Scalar state the compiler
        parks in vector lanes (`v_readlane` / `v_writelane`, the big instantiations run out of SGPRs) counts as scalar bookkeeping
        when the tool looks for the wait behind a branch.
  bar   gfx950 backs `s_barrier` off instead of waiting for the wave's outstanding LDS writes, and the compiler adds no wait in
        front of the raw builtin: every `s_barrier` must be reached with no `ds_write*` outstanding, i.e. behind an
        `s_waitcnt lgkmcnt(0)` on every path (forward dataflow over the kernel's control-flow graph).
  m0    Every LDS-direct load is the third instruction of `s_mov_b32 m0, sN ; s_nop 0 ; global_load_lds_dwordx4` — M0 written by
        the statement itself, one wait state between the SALU write of M0 and the load (the hazard the compiler's recognizer pads
        behind the builtin but cannot see inside an asm string), and nothing of the compiler's in between.

usage: tools/isa_guards.py [--lib PATH | --obj PATH] [--only REGEX]      exit code 1 when a guard fails
"""
import argparse
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "exciting-environments_amd", "exciting_environments_amd", "lib", "libexcenv_hip.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"

VMEM_PREFIXES = ("global_", "buffer_", "flat_", "scratch_", "tbuffer_", "image_")
_SYM = re.compile(r"^([0-9a-f]+) <(\S+)>:")
_INSN = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")
_TARGET = re.compile(r"<[^>]*\+0x([0-9a-fA-F]+)>\s*$")
_CNT = re.compile(r"(vmcnt|lgkmcnt|expcnt)\((\d+)\)")


class Insn:
    __slots__ = ("addr", "op", "args", "target")

    def __init__(self, addr, op, args, target):
        self.addr, self.op, self.args, self.target = addr, op, args, target

    def __repr__(self):
        return f"{self.addr:#x}: {self.op} {self.args}"


def parse_listing(lines, only=None):
    """objdump -d text -> {kernel symbol: [Insn]} (branch targets as absolute addresses). `only`: compiled regex on the symbol."""
    out, sym, base, cur = {}, None, 0, None
    for line in lines:
        m = _SYM.match(line)
        if m:
            sym, base = m.group(2), int(m.group(1), 16)
            cur = [] if (only is None or only.search(sym)) else None
            if cur is not None:
                out[sym] = cur
            continue
        if cur is None:
            continue
        m = _INSN.match(line)
        if not m:
            continue
        op, args, addr = m.group(1), m.group(2), int(m.group(3), 16)
        target = None
        if op.startswith(("s_branch", "s_cbranch")):
            t = _TARGET.search(line)
            target = base + int(t.group(1), 16) if t else None
        cur.append(Insn(addr, op, args, target))
    return out


def device_listing(path, only=None):
    """Disassemble every gfx950 code object embedded in a shared library / object file."""
    out = {}
    with tempfile.TemporaryDirectory() as td:
        so = os.path.join(td, "lib.so")
        shutil.copy(path, so)
        subprocess.run([OBJDUMP, "--offloading", so], check=True, cwd=td, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        elfs = sorted(glob.glob(os.path.join(td, "lib.so.*amdgcn*")))
        if not elfs:
            raise RuntimeError(f"no device code object found in {path}")
        for elf in elfs:
            p = subprocess.Popen([OBJDUMP, "-d", elf], stdout=subprocess.PIPE, text=True)
            out.update(parse_listing(p.stdout, only))
            p.wait()
    return out


def waitcnt(insn):
    """{'vmcnt': n, 'lgkmcnt': n, 'expcnt': n} of an s_waitcnt (absent counters are not waited for)."""
    return {k: int(v) for k, v in _CNT.findall(insn.args)} if insn.op == "s_waitcnt" else {}


def is_vmem(insn):
    return insn.op.startswith(VMEM_PREFIXES)


def is_dma(insn):
    return insn.op.startswith("global_load_lds")


def _index(insns):
    return {x.addr: i for i, x in enumerate(insns)}


def successors(insns, at, i):
    x = insns[i]
    if x.op == "s_endpgm":
        return []
    if x.op.startswith("s_swappc"):
        # a call of one of the out-of-line slow paths (devmath.hpp sincos_lib / xfmod_slow): it returns behind the call. Whatever
        # vector-memory instructions the callee issues only add to the count behind a fill (the safe direction for check_dma_waits),
        # and run() checks that no callee touches LDS or a barrier.
        return [i + 1] if i + 1 < len(insns) else []
    if x.op.startswith("s_setpc"):
        raise RuntimeError(f"indirect branch at {x!r}: the guards need a static control-flow graph")
    if x.op.startswith("s_branch"):
        return [at[x.target]]
    nxt = [i + 1] if i + 1 < len(insns) else []
    if x.op.startswith("s_cbranch"):
        return nxt + [at[x.target]]
    return nxt


def _leads_to_marked_wait(insns, i):
    """From instruction i, falling through scalar instructions (compares, further conditional branches of an `a && b && c` chain),
    is a hand-written (expcnt-marked) vmcnt wait the first thing reached?"""
    for j in range(i, min(i + 32, len(insns))):
        x = insns[j]
        if x.op == "s_waitcnt":
            w = waitcnt(x)
            return "vmcnt" in w and w.get("expcnt") == 6
        if x.op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
            continue  # scalar state kept in a vector register's lanes (SGPR spills of the big instantiations): still scalar bookkeeping
        if not x.op.startswith("s_") or x.op.startswith(("s_branch", "s_endpgm", "s_barrier", "s_load", "s_buffer")):
            return False
    return False


def check_dma_waits(insns):
    """[(message)] — empty when every LDS-direct fill is proven complete in front of the first LDS read on every path."""
    at = _index(insns)
    problems, sites = [], []
    for d, x in enumerate(insns):
        if not is_dma(x):
            continue
        seen, stack = set(), [(d + 1, 0)]
        while stack:
            i, cnt = stack.pop()
            if (i, cnt) in seen or i >= len(insns):
                continue
            seen.add((i, cnt))
            y = insns[i]
            if is_dma(y):
                continue  # in-order retirement: that fill's own check covers this one
            if is_vmem(y):
                cnt = min(cnt + 1, 63)
            elif y.op == "s_waitcnt":
                w = waitcnt(y)
                if "vmcnt" in w:
                    if w["vmcnt"] == 0 or cnt >= w["vmcnt"]:
                        sites.append((y.addr, w["vmcnt"], cnt, w.get("expcnt") == 6))
                        continue  # fill complete on this path
                    if w.get("expcnt") == 6:
                        problems.append(f"hand-written wait {y!r} reached with only {cnt} vector-memory instructions behind the fill "
                                        f"{x!r}: vmcnt({w['vmcnt']}) does not prove it complete")
                        continue
            elif y.op.startswith("ds_read") or y.op.startswith("ds_load"):
                problems.append(f"{y!r} can execute while the fill {x!r} is in flight (no sufficient vmcnt wait on the path)")
                continue
            if y.op.startswith("s_cbranch"):
                fall, tgt = i + 1, at[y.target]
                a, b = _leads_to_marked_wait(insns, fall), _leads_to_marked_wait(insns, tgt)
                if a != b:  # the wave-uniform `if (window starts) wait;`: only the waiting edge (see the module docstring)
                    stack.append((fall if a else tgt, cnt))
                    continue
            for s in successors(insns, at, i):
                stack.append((s, cnt))
    return problems, sites


def check_barriers(insns):
    """[(message)] — empty when no s_barrier can be reached with an LDS write of this wave outstanding."""
    at = _index(insns)
    n = len(insns)
    dirty_in = [None] * n  # None: unreached; else the ds_write that may be outstanding on entry, or False
    work = [(0, False)]
    problems = {}
    while work:
        i, d = work.pop()
        while i < n:
            prev = dirty_in[i]
            merged = d if prev is None else (prev or d)
            if prev is not None and bool(prev) == bool(merged):
                break
            dirty_in[i] = merged
            d = merged
            x = insns[i]
            if x.op == "s_barrier" and d:
                problems[x.addr] = f"{x!r} can be reached with {d} outstanding (no s_waitcnt lgkmcnt(0) in between)"
            if x.op.startswith(("ds_write", "ds_store")):
                d = repr(x)
            elif x.op == "s_waitcnt" and waitcnt(x).get("lgkmcnt") == 0:
                d = False
            succ = successors(insns, at, i)
            if not succ:
                break
            for s in succ[1:]:
                work.append((s, d))
            i = succ[0]
    return list(problems.values())


def check_m0(insns):
    """[(message)] — every LDS-direct load directly follows `s_mov_b32 m0, ...; s_nop`."""
    problems = []
    for i, x in enumerate(insns):
        if not is_dma(x):
            continue
        ok = i >= 2 and insns[i - 1].op == "s_nop" and insns[i - 2].op == "s_mov_b32" and insns[i - 2].args.split(",")[0].strip() == "m0"
        if not ok:
            problems.append(f"{x!r} is not preceded by `s_mov_b32 m0, ...; s_nop` (found {insns[max(i - 2, 0):i]!r})")
    return problems


def run(path, only=None):
    """{'kernels': n, 'dma_kernels': n, 'barrier_kernels': n, 'problems': {symbol: [messages]}, 'dma_sites': {(vmcnt, behind, marked): count}}"""
    listing = device_listing(path, re.compile(only) if only else None)
    problems, sites, n_dma, n_bar = {}, {}, 0, 0
    callees = [sym for sym, insns in listing.items() if insns and insns[-1].op.startswith("s_setpc")]  # functions, not kernels
    for sym in callees:
        bad = [x for x in listing[sym] if x.op.startswith(("ds_", "s_barrier")) or is_dma(x)]
        if bad:
            problems[sym] = [f"out-of-line function touches LDS / a barrier: {bad[0]!r}"]
    for sym, insns in listing.items():
        if sym in callees:
            continue
        msgs = []
        if any(is_dma(x) for x in insns):
            n_dma += 1
            p, s = check_dma_waits(insns)
            msgs += p + check_m0(insns)
            for _, vm, cnt, marked in set(s):
                sites[(vm, cnt, marked)] = sites.get((vm, cnt, marked), 0) + 1
        if any(x.op == "s_barrier" for x in insns):
            n_bar += 1
            msgs += check_barriers(insns)
        if msgs:
            problems[sym] = msgs
    return {"kernels": len(listing), "dma_kernels": n_dma, "barrier_kernels": n_bar, "problems": problems, "dma_sites": sites}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=LIB)
    ap.add_argument("--obj", default=None, help="an object file (csrc/*.o) instead of the library")
    ap.add_argument("--only", default=None, help="regex on the kernel symbol")
    a = ap.parse_args()
    r = run(a.obj or a.lib, a.only)
    print(f"{r['kernels']} kernels, {r['dma_kernels']} with LDS-direct loads, {r['barrier_kernels']} with barriers")
    for (vm, cnt, marked), k in sorted(r["dma_sites"].items()):
        print(f"  fills proven complete by vmcnt({vm}){' [hand-written]' if marked else ''} with {cnt} vector-memory instructions behind them: {k} sites")
    for sym, msgs in r["problems"].items():
        print(sym)
        for m in msgs[:8]:
            print("   ", m)
    print("FAILED" if r["problems"] else "ok")
    sys.exit(1 if r["problems"] else 0)


if __name__ == "__main__":
    main()
