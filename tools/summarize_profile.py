#!/usr/bin/env python3
"""Condense rocprofv3 CSV output into the small files committed under profiles/.
usage: summarize_profile.py <gpurun_out/prof dir> <profiles dir> <tag> [traffic key]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
key = sys.argv[4] if len(sys.argv) > 4 else None
os.makedirs(dst, exist_ok=True)


def find(sub, pat):
    return sorted(glob.glob(os.path.join(src, sub, "**", pat), recursive=True))


lines = [f"# rocprofv3 summary {tag}", ""]
# ---- the un-profiled bench line of the SAME gpurun session + the roofline fraction recomputed from the profile ---------
bench = None
bpath = os.path.join(src, "bench.json")
if os.path.exists(bpath):
    try:
        bench = json.loads([l for l in open(bpath).read().splitlines() if l.startswith("{")][-1])
    except Exception:
        bench = None
hot_avg_ms = None
for f in find("trace", "*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "sim_ahead" in r.get("Name", "") or "step_kernel" in r.get("Name", ""):
            avg = float(r.get("AverageNs", 0) or 0) / 1e6
            if hot_avg_ms is None or float(r.get("TotalDurationNs", 0) or 0) > hot_avg_ms[1]:
                hot_avg_ms = (avg, float(r.get("TotalDurationNs", 0) or 0), r.get("Name", ""), r.get("Calls"))
if bench is not None:
    rf = bench["roofline"]
    lines += ["## same-session bench line (un-profiled run of `bench.py`, HIP events) and the fraction recomputed from this profile", "",
              f"* workload: {bench['config']['workload']}",
              f"* bench line: value {bench['value']:.4e} {bench['unit']}, ms_per_step {bench['ms_per_step']:.4f}, "
              f"kernel_ms (HIP events) {rf['kernel_ms']:.4f}, roofline.frac {rf['frac']:.4f}",
              f"* algorithmic bytes per launch: {rf['algorithmic_bytes_per_env_step']} B x {bench['config']['batch_per_gpu']} envs x "
              f"{bench['config']['chunk_steps']} steps = {rf['algorithmic_bytes_per_launch']:.4e} B"]
    if hot_avg_ms is not None:
        gbs = rf["algorithmic_bytes_per_launch"] / (hot_avg_ms[0] * 1e-3) / 1e9
        lines += [f"* rocprofv3 --kernel-trace --stats average of the dominant kernel ({hot_avg_ms[3]} calls, profiled run): "
                  f"{hot_avg_ms[0]:.4f} ms -> {gbs:.0f} GB/s algorithmic = **{gbs / 8000.0:.4f} of the 8 TB/s HBM peak** "
                  f"(bench line, un-profiled: {rf['frac']:.4f}; the profiled run is another process: its buffers sit elsewhere in physical memory, which moves this kernel by up to 25 %, DESIGN.md §6, and profiled passes run at a slightly lower clock)"]
        # the timed steps alone: the last `steps` dispatches of that kernel in the trace (the earlier ones are warm-up and the
        # placement probes of new output sets, core_env.py — several of those run in the slow level by design)
        try:
            tb = json.loads([l for l in open(os.path.join(src, "bench_under_trace.json")).read().splitlines() if l.startswith("{")][-1])
            nsteps = int(tb["steps"])
            durs = []
            for f in find("trace", "*kernel_trace.csv"):
                rows = [r for r in csv.DictReader(open(f)) if r.get("Kernel_Name", "") == hot_avg_ms[2]]
                rows.sort(key=lambda r: int(r["Start_Timestamp"]))
                durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows][-nsteps:]
            if durs:
                avg = sum(durs) / len(durs)
                gbs = rf["algorithmic_bytes_per_launch"] / (avg * 1e-3) / 1e9
                lines += [f"* the {len(durs)} timed steps of the profiled run alone (last dispatches of the trace): average {avg:.4f} ms "
                          f"(min {min(durs):.4f}, max {max(durs):.4f}) -> {gbs:.0f} GB/s = **{gbs / 8000.0:.4f}**"]
        except Exception as e:  # noqa: BLE001
            lines += [f"* (timed steps of the profiled run: not available: {e})"]
    lines.append("")
# ---- kernel stats ------------------------------------------------------------------------------
for f in find("trace", "*kernel_stats.csv"):
    lines += [f"## kernel-trace --stats ({os.path.basename(f)})", "", "| kernel | calls | total ms | avg ms | min ms | max ms | % |", "|---|---|---|---|---|---|---|"]
    for r in csv.DictReader(open(f)):
        name = r.get("Name", "?")
        short = name if len(name) < 110 else name[:107] + "..."
        ns = lambda k: float(r.get(k, 0) or 0) / 1e6
        lines.append(f"| `{short}` | {r.get('Calls')} | {ns('TotalDurationNs'):.3f} | {ns('AverageNs'):.4f} | {ns('MinNs'):.4f} | {ns('MaxNs'):.4f} | {r.get('Percentage')} |")
    lines.append("")
# per-dispatch resource usage of the hot kernel
for f in find("trace", "*kernel_trace.csv"):
    seen = {}
    for r in csv.DictReader(open(f)):
        n = r.get("Kernel_Name", "")
        if "sim_ahead" in n or "step_kernel" in n:
            seen[n] = r
    if seen:
        lines += ["## dispatch resources (kernel_trace.csv)", "", "| kernel | VGPR | accum VGPR | SGPR | LDS | scratch | workgroup | grid |", "|---|---|---|---|---|---|---|---|"]
        for n, r in seen.items():
            lines.append(f"| `{n[:90]}` | {r.get('VGPR_Count')} | {r.get('Accum_VGPR_Count')} | {r.get('SGPR_Count')} | {r.get('LDS_Block_Size')} | {r.get('Scratch_Size')} | {r.get('Workgroup_Size')} | {r.get('Grid_Size')} |")
        lines += ["", "Units of this table: rocprofv3's `VGPR_Count` on gfx950 is HALF the registers a wave64 lane is allocated — the "
                  "headline kernel compiles to 198 VGPRs (`-Rpass-analysis=kernel-resource-usage`, `profiles/*_loop_isa.md`), the "
                  "allocation granule rounds that to 208, the trace shows 104; `SGPR_Count` is the allocated block (112), not the 93 "
                  "the kernel uses; `LDS_Block_Size` is in granules. 512 registers per SIMD lane / 208 = two resident waves per "
                  "SIMD; the SQ section below gives the measured residency.", ""]

# ---- PMC ---------------------------------------------------------------------------------------
pmc = {}
for sub, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    per = defaultdict(lambda: defaultdict(float))  # kernel -> dispatch -> value
    for f in find(sub, "*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            per[r.get("Kernel_Name", "?")][r.get("Dispatch_Id")] += float(r.get("Counter_Value", 0) or 0)
    pmc[counter] = {k: (sum(v.values()) / len(v), len(v)) for k, v in per.items()}
if any(pmc.values()):
    lines += ["## PMC (separate passes; counter unit = KiB per dispatch, averaged over dispatches)", "",
              "| kernel | dispatches | FETCH_SIZE KiB | WRITE_SIZE KiB |", "|---|---|---|---|"]
    kernels = sorted(set(pmc.get("FETCH_SIZE", {})) | set(pmc.get("WRITE_SIZE", {})))
    hot = None
    calib = None
    for k in kernels:
        fz, n1 = pmc.get("FETCH_SIZE", {}).get(k, (float("nan"), 0))
        wz, n2 = pmc.get("WRITE_SIZE", {}).get(k, (float("nan"), 0))
        if not (fz + wz >= 1000 or "sim_ahead" in k or "step_kernel" in k):
            continue
        lines.append(f"| `{k[:100]}` | {max(n1, n2)} | {fz:.0f} | {wz:.0f} |")
        if "sim_ahead" in k or "step_kernel" in k:
            hot = (fz, wz, k)
        if "trunc" in k.lower():
            calib = (fz, wz)
    lines.append("")
    gib = float(1 << 20)  # KiB in 1 GiB
    if calib:
        lines += [f"Calibration (`torch.trunc` over 2^28 fp32: 1 GiB read + 1 GiB written, 16 B/lane): FETCH_SIZE reports "
                  f"{calib[0] / gib:.3f} x the bytes read, WRITE_SIZE {calib[1] / gib:.3f} x the bytes written "
                  "(MI355X_MICROARCH.md §HBM: FETCH_SIZE counts exactly 1/2 of a wide coalesced read stream on gfx950, "
                  "WRITE_SIZE is exact).", ""]
    if hot:
        # correction prescribed by the guide: double FETCH_SIZE, take WRITE_SIZE as is
        hbm = (2.0 * hot[0] + hot[1]) * 1024.0
        lines += [f"`{hot[2][:60]}...`: HBM traffic per launch = (2 x {hot[0]:.0f} KiB + {hot[1]:.0f} KiB) x 1024 "
                  f"= {hbm:.4e} bytes.", ""]
        if "rowmajor" in tag or "_em_" in tag:
            lo = (1.0 * hot[0] + hot[1]) * 1024.0
            lines += ["Caveat for this kernel: the x2 is calibrated on wide coalesced reads (128-byte fabric requests tallied at 64 "
                      "bytes). The row-major action array is read as 64-byte windows, four adjacent lanes per window "
                      "(DESIGN.md §4.1b): FETCH_SIZE / 64 B is then the number of fabric requests, one per 128-byte line a window "
                      "touches, and what a request moves is 64 bytes if the L2 fills sectors and 128 if it fills lines (part of the "
                      f"second requests of a line are Infinity-Cache hits, which the counter includes). Bounds: {lo:.4e} ... "
                      f"{hbm:.4e} bytes per launch.", ""]
        if key:
            tpath = os.path.join(dst, "traffic.json")
            tj = json.load(open(tpath)) if os.path.exists(tpath) else {}
            tj[key] = {"hbm_bytes_per_launch": hbm, "fetch_size_kib": hot[0], "write_size_kib": hot[1],
                       "fetch_correction": 2.0, "write_correction": 1.0,
                       "calibration_fetch_ratio": calib[0] / gib if calib else None,
                       "calibration_write_ratio": calib[1] / gib if calib else None,
                       "source": f"profiles/{tag}_rocprof_summary.md"}
            json.dump(tj, open(tpath, "w"), indent=1, sort_keys=True)
# ---- SQ counters of the hot kernel (own passes) ----------------------------------------------------
sq = defaultdict(lambda: defaultdict(float))
nd = defaultdict(set)
for sub in ("pmc_sq", "pmc_sq2"):
    for f in find(sub, "*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            if "sim_ahead" in k or "step_kernel" in k:
                sq[k][r.get("Counter_Name")] += float(r.get("Counter_Value", 0) or 0)
                nd[(k, r.get("Counter_Name"))].add(r.get("Dispatch_Id"))
if sq:
    lines += ["## SQ counters (per dispatch, summed over the chip)", ""]
    for k, cs in sq.items():
        lines += [f"`{k[:100]}`", "", "| counter | value per dispatch |", "|---|---|"]
        per = {c: v / max(1, len(nd[(k, c)])) for c, v in cs.items()}
        for c in sorted(per):
            lines.append(f"| {c} | {per[c]:.4e} |")
        if "SQ_INSTS_VALU" in per and "SQ_WAVES" in per and per["SQ_WAVES"] > 0:
            lines.append(f"| VALU wave-instructions per wave | {per['SQ_INSTS_VALU'] / per['SQ_WAVES']:.1f} |")
        if "SQ_WAIT_ANY" in per and "SQ_WAVE_CYCLES" in per and per["SQ_WAVE_CYCLES"] > 0:
            lines.append(f"| SQ_WAIT_ANY / SQ_WAVE_CYCLES (wave parked on s_waitcnt) | {per['SQ_WAIT_ANY'] / per['SQ_WAVE_CYCLES']:.3f} |")
        if "SQ_ACTIVE_INST_VALU" in per and "SQ_WAVE_CYCLES" in per and per["SQ_WAVE_CYCLES"] > 0:
            lines.append(f"| SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES | {per['SQ_ACTIVE_INST_VALU'] / per['SQ_WAVE_CYCLES']:.3f} |")
        if per.get("GRBM_GUI_ACTIVE", 0) > 0 and per.get("SQ_WAVE_CYCLES", 0) > 0:
            # SQ_WAVE_CYCLES counts quad-cycles of resident waves summed over the chip, GRBM_GUI_ACTIVE busy cycles summed over the
            # 8 XCDs: resident waves per SIMD, averaged over the launch = 4 * SQ_WAVE_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8)
            occ = 4.0 * per["SQ_WAVE_CYCLES"] / (1024.0 * per["GRBM_GUI_ACTIVE"] / 8.0)
            lines.append(f"| measured residency: 4 x SQ_WAVE_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs) = waves per SIMD, launch average | {occ:.2f} |")
        lines.append("")
open(os.path.join(dst, f"{tag}_rocprof_summary.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
