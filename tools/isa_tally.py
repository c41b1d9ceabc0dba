#!/usr/bin/env python3
"""Where do the instructions of a trajectory loop come from? The device assembly of one translation unit, compiled with line tables
(`hipcc -S --cuda-device-only -gline-tables-only`), carries a `.loc file line` in front of every instruction: the innermost inlined
source line. This tool takes one kernel symbol, finds its K loop (the largest backward branch), and tallies the loop's instructions
per SOURCE SECTION — the csrc/*.hpp function the line belongs to, grouped (vector field, RK stage sums, hexagon clip, sin / cos,
2 pi remainder, observation, ...) — split into vector / scalar / memory instructions. Static counts of the loop body, hot and cold
(the out-of-line slow paths are calls; the in-line IEEE-division fall-backs are listed on their own line).
usage: tools/isa_tally.py <device.s> <kernel symbol regex> [--md OUT]"""
import argparse
import collections
import os
import re
import sys

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "exciting-environments_amd", "csrc")
NAMES = ("pymod_two_pi", "wrap_angle", "sincos_fast", "sincos_t", "sincos_defer", "fastq", "div_all_defer", "div_all", "hex_clip", "constraint",
         "observe_defer", "observe", "torque", "post", "rk_step", "env_advance_raw", "env_step", "ahead_time", "normalize_fields",
         "normalize_field", "denormalize", "normalize", "max_nan", "min_nan", "save_row", "load_action", "dma_window", "advance",
         "publish_last", "next_index", "store_stream", "store_flags", "store_v", "load_v", "load_row", "load_ctx", "prep_ctx", "pmsm_reward", "init", "div", "f")


def function_ranges(path):
    """[(first line, last line, name)] of the functions / lambdas of a header that carry one of NAMES, by a brace count."""
    out, lines = [], open(path).read().splitlines()
    for i, l in enumerate(lines):
        st = l.strip()
        if st.startswith("//") or "(" not in l:
            continue
        hit = None
        for n in NAMES:
            if re.search(r"(\b|::)" + re.escape(n) + r"\s*(<[^>]*>)?\s*\(", l) or re.search(r"auto\s+" + re.escape(n) + r"\s*=\s*\[", l):
                hit = n
                break
        if not hit or not ("__device__" in l or "auto " in l or "static" in l):
            continue
        if ";" in l and "{" not in l:
            continue  # a declaration or a call
        depth, j, seen = 0, i, False
        while j < len(lines):
            depth += lines[j].count("{") - lines[j].count("}")
            seen = seen or "{" in lines[j]
            if seen and depth <= 0:
                break
            j += 1
        if seen:
            out.append((i + 1, j + 1, hit))
    return out


SECTION_OF = {
    "f": "vector field (M::f)", "rk_step": "RK stage sums, y + sum a k, k = f dt (rk_step)", "hex_clip": "hexagon clip (sector pick, rotation, clamp)",
    "max_nan": "hexagon clip (sector pick, rotation, clamp)", "min_nan": "hexagon clip (sector pick, rotation, clamp)",
    "constraint": "constraint_denormalization (Park rotations, scaling)", "sincos_fast": "sin / cos", "sincos_t": "sin / cos", "sincos_defer": "sin / cos",
    "pymod_two_pi": "2 pi remainder / wrap_angle", "wrap_angle": "2 pi remainder / wrap_angle",
    "fastq": "invariant division, fast path (InvDiv)", "div_all_defer": "invariant division, fast path (InvDiv)", "div": "invariant division, fast path (InvDiv)",
    "div_all": "invariant division: guard + in-line IEEE fall-back (cold)", "init": "invariant division: set-up",
    "observe": "generate_observation (normalisations)", "observe_defer": "generate_observation (normalisations)", "normalize_fields": "generate_observation (normalisations)",
    "normalize_field": "generate_observation (normalisations)", "normalize": "generate_observation (normalisations)",
    "torque": "post-processing of the saved row (torque)", "post": "post-processing of the saved row (torque)",
    "env_advance_raw": "action path (denormalise, dead time, predicted angle)", "env_step": "action path (denormalise, dead time, predicted angle)",
    "ahead_time": "action path (denormalise, dead time, predicted angle)", "denormalize": "action path (denormalise, dead time, predicted angle)",
    "save_row": "save row (stores, addresses)", "store_stream": "save row (stores, addresses)", "store_v": "save row (stores, addresses)", "store_flags": "save row (stores, addresses)",
    "load_action": "action rows (loads / LDS windows)", "dma_window": "action rows (loads / LDS windows)", "load_v": "action rows (loads / LDS windows)", "load_row": "action rows (loads / LDS windows)",
    "advance": "loop glue (indices, register moves)", "next_index": "loop glue (indices, register moves)", "publish_last": "loop glue (indices, register moves)",
    "load_ctx": "loop glue (indices, register moves)", "prep_ctx": "loop glue (indices, register moves)", "pmsm_reward": "gym outputs",
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("asm")
    ap.add_argument("symbol")
    ap.add_argument("--md", default=None)
    a = ap.parse_args()
    files = {}
    sym_re = re.compile(a.symbol)
    cur, body = None, []
    for line in open(a.asm, errors="replace"):
        m = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', line)
        if m:
            files[int(m.group(1))] = m.group(2)
            continue
        m = re.match(r"^([A-Za-z_][\w$.]*):\s*(;.*)?$", line)
        if m and not m.group(1).startswith(".L"):
            if cur and body:
                break
            cur = m.group(1) if sym_re.search(m.group(1)) else None
            continue
        if cur:
            body.append(line.rstrip("\n"))
    if not cur:
        sys.exit("kernel not found")
    ins, loc, labels = [], (0, 0), {}
    for l in body:
        s = l.strip()
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
        if m:
            loc = (int(m.group(1)), int(m.group(2)))
            continue
        m = re.match(r"^(\.LBB\S+):", s)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        if not s or s.startswith((".", ";", "//")) or s.endswith(":"):
            continue
        op = s.split()[0]
        if re.match(r"^[a-z][a-z_0-9]+$", op):
            ins.append((op, s, loc))
    best = None
    for i, (op, s, _) in enumerate(ins):
        if op.startswith(("s_cbranch", "s_branch")):
            t = s.split()[-1]
            if t in labels and labels[t] <= i and (best is None or i - labels[t] > best[1] - best[0]):
                best = (labels[t], i)
    lo, hi = best
    ranges = {fid: function_ranges(os.path.join(CSRC, name)) for fid, name in files.items() if os.path.exists(os.path.join(CSRC, name))}
    tally = collections.defaultdict(collections.Counter)
    for op, s, (fid, ln) in ins[lo:hi + 1]:
        fn = None
        for a0, b0, name in ranges.get(fid, []):
            if a0 <= ln <= b0 and (fn is None or (b0 - a0) < fn[0]):
                fn = (b0 - a0, name)
        sec = SECTION_OF.get(fn[1], fn[1]) if fn else f"other ({files.get(fid, '?')}: kernel body, library headers)"
        if op.startswith(("v_div_scale", "v_div_fmas", "v_div_fixup")):
            sec = "invariant division: guard + in-line IEEE fall-back (cold)"
        kind = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "mem"
        tally[sec][kind] += 1
        tally[sec]["s_nop"] += op == "s_nop"
        tally[sec]["packed"] += op.startswith("v_pk_")
    rows = sorted(tally.items(), key=lambda kv: -sum(kv[1][k] for k in ("valu", "salu", "mem")))
    tot = collections.Counter()
    out = [f"kernel `{cur}`", "", f"K loop: {hi - lo + 1} instructions of the kernel's {len(ins)}", "",
           "| section (source function of the innermost inlined line) | vector | of which packed (2 lanes' work each) | scalar | of which `s_nop` | memory |", "|---|---|---|---|---|---|"]
    for sec, c in rows:
        out.append(f"| {sec} | {c['valu']} | {c['packed']} | {c['salu']} | {c['s_nop']} | {c['mem']} |")
        tot.update(c)
    out.append(f"| **total** | {tot['valu']} | {tot['packed']} | {tot['salu']} | {tot['s_nop']} | {tot['mem']} |")
    text = "\n".join(out)
    print(text)
    if a.md:
        with open(a.md, "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()
