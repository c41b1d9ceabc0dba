#!/usr/bin/env python3
"""Condense the passes of tools/profile_gym.sh into profiles/<tag>_rocprof_summary.md: per model the plain trajectory launch beside
the launch that also writes reward / terminated / truncated — duration (kernel trace, timed dispatches), bytes written (WRITE_SIZE),
store instructions (SQ_INSTS_VMEM_WR), share of wave time parked on s_waitcnt (SQ_WAIT_ANY / SQ_WAVE_CYCLES).
usage: summarize_gym.py <gpurun_out/prof dir> <profiles dir> <tag>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]


def find(sub, pat):
    return sorted(glob.glob(os.path.join(src, sub, "**", pat), recursive=True))


def variants(path):
    return [json.loads(l) for l in open(path).read().splitlines() if l.startswith("{")] if os.path.exists(path) else []


def is_traj(name):
    return "sim_ahead" in name


def cut(rows, vs):
    """rows: the trajectory dispatches of one process in start order; vs: the probe's variants with their launch counts."""
    total = sum(v["launches"] for v in vs)
    if len(rows) != total:
        return None
    out, at = [], 0
    for v in vs:
        out.append(rows[at:at + v["launches"]])
        at += v["launches"]
    return out


lines = [f"# rocprofv3 summary {tag}: fused gym trajectories beside the plain launch (B = 2^22, K = 100, fp32 Euler, lane-major)", ""]
plain_run = variants(os.path.join(src, "plain_run.jsonl"))
# ---- kernel trace ----------------------------------------------------------------------------------------------------------
vs = variants(os.path.join(src, "trace_run.jsonl"))
trace = []
for f in find("trace", "*kernel_trace.csv"):
    trace += [r for r in csv.DictReader(open(f)) if is_traj(r.get("Kernel_Name", ""))]
trace.sort(key=lambda r: int(r["Start_Timestamp"]))
parts = cut(trace, vs)
dur = {}
if parts is None:
    lines += [f"(kernel trace: {len(trace)} trajectory dispatches, the probe made {sum(v['launches'] for v in vs)} — not cut)", ""]
else:
    for v, rows in zip(vs, parts):
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows[-v["timed"]:]]
        dur[(v["model"], tuple(v["control"]), v["gym"])] = (sum(d) / len(d), min(d), max(d), rows[-1])
# ---- counters ----------------------------------------------------------------------------------------------------------------
cnt = defaultdict(dict)  # (model, control, gym) -> counter -> per-dispatch value
for sub in ("pmc_write", "pmc_fetch", "pmc_sq", "pmc_sq2"):
    pv = variants(os.path.join(src, sub + "_run.jsonl"))
    per = defaultdict(lambda: defaultdict(float))  # dispatch id -> counter -> value
    name = {}
    for f in find(sub, "*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if is_traj(r.get("Kernel_Name", "")):
                per[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r.get("Counter_Value", 0) or 0)
    ids = sorted(per)
    p = cut(ids, pv)
    if p is None:
        lines += [f"({sub}: {len(ids)} trajectory dispatches, the probe made {sum(v['launches'] for v in pv)} — not cut)", ""]
        continue
    for v, dids in zip(pv, p):
        last = per[dids[-1]]  # the last launch of the variant
        for c, val in last.items():
            cnt[(v["model"], tuple(v["control"]), v["gym"])][c] = val

lines += ["## per model: plain launch | with reward / terminated / truncated", "",
          "Durations: rocprofv3 `--kernel-trace`, average of the 10 timed dispatches of each variant (after the placement of its output "
          "set has settled); `un-profiled` = the same probe without the profiler (wall clock around 10 calls incl. the control-column "
          "fill kernel). Counters: own `--pmc` passes, one dispatch. WRITE_SIZE in KiB x 1024 = bytes. `alg. B / env-step` = observations + state "
          "leaves written + actions read by the trajectory kernel (+ reward 4, terminated 1, truncated TW with gym outputs); in brackets "
          "the whole call with the control columns of the observations, which `control_fill_kernel` writes behind it.", "",
          "| model, controls | variant | kernel form | alg. B / env-step: kernel (call) | trace ms (min … max) | un-profiled ms per call | GB/s alg. | of 8 TB/s | WRITE_SIZE bytes | / alg. written | SQ_INSTS_VMEM_WR | SQ_INSTS_VALU | SQ_WAIT_ANY / SQ_WAVE_CYCLES |",
          "|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
ratio_rows = []
for v in vs:
    k = (v["model"], tuple(v["control"]), v["gym"])
    d = dur.get(k)
    c = cnt.get(k, {})
    un = next((p["ms_per_call"] for p in plain_run if (p["model"], tuple(p["control"]), p["gym"]) == k), float("nan"))
    written = v["kernel_written_per_env_step"] * v["env_steps"]
    kbytes = v["kernel_bytes_per_env_step"] * v["env_steps"]
    ws = c.get("WRITE_SIZE", float("nan")) * 1024.0
    gbs = kbytes / (d[0] * 1e-3) / 1e9 if d else float("nan")
    wait = c.get("SQ_WAIT_ANY", float("nan")) / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else float("nan")
    form = v["form"].split("(")[0].strip() + (" (" + v["form"].split("(", 1)[1] if "(" in v["form"] else "")
    lines.append(f"| {v['model']} {v['control']} | {'gym' if v['gym'] else 'plain'} | {form[:60]} | {v['kernel_bytes_per_env_step']} ({v['call_bytes_per_env_step']}) | "
                 + (f"{d[0]:.3f} ({d[1]:.3f} … {d[2]:.3f})" if d else "—") +
                 f" | {un:.3f} | {gbs:.0f} | {gbs / 8000:.3f} | {ws:.4e} | {ws / written:.3f} | {c.get('SQ_INSTS_VMEM_WR', float('nan')):.4e} | "
                 f"{c.get('SQ_INSTS_VALU', float('nan')):.4e} | {wait:.3f} |")
    ratio_rows.append((k, v, d, c))
lines += ["", "## gym / plain", "", "| model, controls | byte ratio, kernel (call) | time ratio, kernel (trace) | time ratio, call (un-profiled) | store-instruction ratio | VALU ratio |", "|---|---|---|---|---|---|"]
byk = {k: (v, d, c) for k, v, d, c in ratio_rows}
for (m, ctl, gym), (v, d, c) in byk.items():
    if not gym or (m, ctl, False) not in byk:
        continue
    pv, pd, pc = byk[(m, ctl, False)]
    tr = d[0] / pd[0] if d and pd else float("nan")
    sr = c.get("SQ_INSTS_VMEM_WR", float("nan")) / pc.get("SQ_INSTS_VMEM_WR", float("nan")) if pc.get("SQ_INSTS_VMEM_WR") else float("nan")
    vr = c.get("SQ_INSTS_VALU", float("nan")) / pc.get("SQ_INSTS_VALU", float("nan")) if pc.get("SQ_INSTS_VALU") else float("nan")
    un = {(p["model"], tuple(p["control"]), p["gym"]): p["ms_per_call"] for p in plain_run}
    cr = un.get((m, ctl, True), float("nan")) / un.get((m, ctl, False), float("nan")) if un else float("nan")
    lines.append(f"| {m} {list(ctl)} | {v['kernel_bytes_per_env_step'] / pv['kernel_bytes_per_env_step']:.3f} "
                 f"({v['call_bytes_per_env_step'] / pv['call_bytes_per_env_step']:.3f}) | {tr:.3f} | {cr:.3f} | {sr:.3f} | {vr:.3f} |")
# dispatch resources of the gym kernels
if dur:
    lines += ["", "## dispatch resources (kernel_trace.csv; VGPR_Count is half the allocated wave64 registers on gfx950)", "",
              "| model, controls | variant | VGPR | SGPR | LDS | scratch | workgroup | kernel |", "|---|---|---|---|---|---|---|---|"]
    for (m, ctl, gym), (_, _, _, r) in dur.items():
        lines.append(f"| {m} {list(ctl)} | {'gym' if gym else 'plain'} | {r.get('VGPR_Count')} | {r.get('SGPR_Count')} | {r.get('LDS_Block_Size')} | "
                     f"{r.get('Scratch_Size')} | {r.get('Workgroup_Size')} | `{r.get('Kernel_Name', '')[:110]}` |")
os.makedirs(dst, exist_ok=True)
open(os.path.join(dst, f"{tag}_rocprof_summary.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
