#!/usr/bin/env python3
"""Guard rail: how many BYTES of code are the K loops of the trajectory kernels? The instruction cache of a CU pair is 64 KB;
the headline loop (PMSM Euler fp32, V = 4, ping-pong: two unrolled solver steps x 4 environments per lane) is close to it and
any growth — gym outputs on the vector path, a third register set — would fall off that cliff unnoticed (misses are < 1e-5
today, tools/icache_probe.sh). This tool takes the device code out of the built library, disassembles every
sim_ahead_kernel / sim_ahead_em_kernel and reports the span of its largest backward branch = the outermost loop. Exit code 1
when the headline loop exceeds --limit bytes (default 61440 = 60 KB). Runs in the build container (no GPU needed).
usage: tools/loop_code_size.py [--limit BYTES] [--all]"""
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
HEADLINE = "_ZN6excenv16sim_ahead_kernelINS_4PmsmIfEEfLi0ELb1ELb0ELi4ELi1ELb0ELb0ELb0ELi256EEEvNS_7SimArgsIT0_T_EE"


def loop_spans(lib=LIB):
    """{kernel symbol: (largest backward-branch span in bytes, kernel size in bytes)} for the trajectory kernels."""
    out = {}
    with tempfile.TemporaryDirectory() as td:
        so = os.path.join(td, "lib.so")
        shutil.copy(lib, so)
        subprocess.run([OBJDUMP, "--offloading", so], check=True, cwd=td, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        for elf in sorted(glob.glob(os.path.join(td, "lib.so.*amdgcn*"))):
            dis = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", elf], check=True, capture_output=True, text=True).stdout
            sym, first, last, span = None, None, None, 0
            for line in dis.splitlines():
                m = re.match(r"^([0-9a-f]+) <(\S+)>:", line)
                if m:
                    if sym and "sim_ahead" in sym:
                        out[sym] = (span, last - first)
                    sym, first, last, span = m.group(2), int(m.group(1), 16), int(m.group(1), 16), 0
                    continue
                m = re.match(r"^\s+(s_c?branch\S*)\s.*?//\s*([0-9A-Fa-f]+):", line) or re.match(r"^\s+(\S+).*//\s*([0-9A-Fa-f]+):", line)
                if not m or sym is None:
                    continue
                addr = int(m.group(2), 16)
                last = max(last, addr)
                if m.group(1).startswith(("s_branch", "s_cbranch")):
                    t = re.search(r"<[^>]*\+0x([0-9a-fA-F]+)>", line)
                    if t:
                        target = first + int(t.group(1), 16)
                        if target <= addr:
                            span = max(span, addr - target)
            if sym and "sim_ahead" in sym:
                out[sym] = (span, last - first)
    return out


READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def kernel_resources(lib=LIB):
    """{kernel symbol: {"scratch": private segment bytes per lane, "vgpr": count, "sgpr": count}} from the code objects' metadata
    notes. A trajectory kernel that starts to spill reloads its registers behind `s_waitcnt vmcnt(0)`, i.e. behind every
    outstanding trajectory store (kernels_emr.hpp: 7.7 -> 7.1 ms when the last spills went): worth failing the build for."""
    out = {}
    with tempfile.TemporaryDirectory() as td:
        so = os.path.join(td, "lib.so")
        shutil.copy(lib, so)
        subprocess.run([OBJDUMP, "--offloading", so], check=True, cwd=td, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        for elf in sorted(glob.glob(os.path.join(td, "lib.so.*amdgcn*"))):
            notes = subprocess.run([READELF, "--notes", elf], check=True, capture_output=True, text=True).stdout
            cur = {}
            for line in notes.splitlines():
                m = re.match(r"\s*-?\s*\.(name|private_segment_fixed_size|vgpr_count|sgpr_count):\s*(\S+)", line)
                if not m:
                    continue
                k, v = m.group(1), m.group(2).strip("'\"")
                if k == "name" and not v.startswith("_Z"):
                    continue  # argument names
                cur[k] = v
                if all(x in cur for x in ("name", "private_segment_fixed_size", "vgpr_count", "sgpr_count")):
                    out[cur["name"]] = {"scratch": int(cur["private_segment_fixed_size"]), "vgpr": int(cur["vgpr_count"]),
                                        "sgpr": int(cur["sgpr_count"])}
                    cur = {}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--limit", type=int, default=60 * 1024)
    ap.add_argument("--all", action="store_true", help="print every trajectory kernel, largest loops first")
    a = ap.parse_args()
    spans = loop_spans()
    if HEADLINE not in spans:
        sys.exit(f"headline kernel {HEADLINE} not found in {LIB}")
    if a.all:
        for s, (sp, sz) in sorted(spans.items(), key=lambda kv: -kv[1][0])[:40]:
            print(f"{sp:8d} B loop  {sz:8d} B kernel  {s}")
    sp, sz = spans[HEADLINE]
    print(f"headline K loop: {sp} bytes of {sz} ({sp / 1024:.1f} KB; limit {a.limit / 1024:.0f} KB, instruction cache 64 KB)")
    worst = max(spans.values())[0]
    print(f"largest loop of any trajectory kernel: {worst} bytes")
    sys.exit(0 if sp <= a.limit else 1)


if __name__ == "__main__":
    main()
