#!/usr/bin/env python3
"""Compile csrc/env_pmsm.hip to gfx950 assembly and write the K-loop memory / wait skeleton of the headline kernel
(sim_ahead_kernel<Pmsm<float>, float, Euler, AHEAD, !GENERAL, V=4>) to profiles/: every vector-memory instruction and every
s_waitcnt vmcnt between the loop header and its back edge, with the VALU instruction counts in between. Shows where the wait
for the prefetched action row sits relative to the trajectory stores. Runs in the build container (hipcc
cross-compiles; no GPU needed). usage: tools/isa_excerpt.py [out.md]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "exciting-environments_amd", "csrc")
KERNEL = "_ZN6excenv16sim_ahead_kernelINS_4PmsmIfEEfLi0ELb1ELb0ELi4ELi1ELb0ELb0ELb0ELi256EEEvNS_7SimArgsIT0_T_EE"
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r02_sim_ahead_loop_isa.md")
flags = "-O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function".split()
with tempfile.TemporaryDirectory() as td:
    asm = os.path.join(td, "env_pmsm.s")
    subprocess.run(["/opt/rocm/bin/hipcc", *flags, "-S", "--cuda-device-only", "-o", asm, os.path.join(CSRC, "env_pmsm.hip")],
                   check=True, stderr=subprocess.DEVNULL)
    lines = open(asm).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(KERNEL + ":"))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".end_amdhsa_kernel"))
k = lines[start:end]
meta = {m.group(1): m.group(2) for l in k for m in [re.match(r"\s+\.amdhsa_(next_free_vgpr|next_free_sgpr)\s+(\d+)", l)] if m}
hdr = next(i for i, l in enumerate(k) if "Loop Header: Depth=1" in l)
label = re.match(r"(\.LBB\d+_\d+):", k[hdr]).group(1)
# the loop's blocks are those annotated "in Loop: Header=<label>"; block placement may put some before the header label
tag = label.replace(".L", "")
in_loop = [False] * len(k)
cur = False
for i, l in enumerate(k):
    m = re.match(r"(\.LBB\d+_\d+):", l)
    if m:
        cur = (m.group(1) == label) or (f"Header={tag}" in l)
    in_loop[i] = cur
is_inst = lambda l: bool(re.match(r"\s+[a-z]", l)) and not l.strip().startswith(".")
rows, valu, salu = [], 0, 0
n_inst = 0
for i, l in enumerate(k):
    if not in_loop[i] or not is_inst(l):
        continue
    n_inst += 1
    op = l.split()[0]
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")) or (op == "s_waitcnt" and "vmcnt" in l) or op == "s_load_dword" or op.startswith("s_load_dword"):
        rows.append((valu, salu, l.strip()))
        valu = salu = 0
    elif op.startswith("v_"):
        valu += 1
    elif op.startswith("s_"):
        salu += 1
md = [f"# K-loop memory / wait skeleton of the headline kernel (gfx950 ISA, built {flags[0]} -ffp-contract=off)", "",
      f"`{KERNEL}`", "", f"next_free_vgpr {meta.get('next_free_vgpr')}, next_free_sgpr {meta.get('next_free_sgpr')}; "
      f"{n_inst} instructions in the loop (two unrolled solver steps x 4 environments per lane, slow paths included).", "",
      "Order below is program order inside the loop (blocks in layout order). Columns: VALU / SALU instructions since the "
      "previous listed instruction, then the instruction. `global_load_dwordx4` x2 = the action row of the NEXT solver step "
      "(component-major registers, unconditional); `global_store_dwordx4 ... nt` x15 = one trajectory row (8 observation + 7 "
      "state streams). vmcnt counts loads and stores in issue order, so `s_waitcnt vmcnt(N)` lets the N youngest operations "
      "stay in flight.", "", "| VALU | SALU | instruction |", "|---|---|---|"]
for v, s, ins in rows:
    md.append(f"| {v} | {s} | `{ins}` |")
waits = [r for r in rows if r[2].startswith("s_waitcnt")]
md += ["", f"vmcnt waits inside the loop: {', '.join(w[2] for w in waits) or 'none'}.", ""]
open(out, "w").write("\n".join(md) + "\n")
print("\n".join(md[-60:]))
