# usage (ON THE GPU BOX): bash tools/placement_pmc.sh — the placement probe under rocprofv3 --pmc (TLB / write-stall counters per launch)
set -u
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/placement_pmc
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE --output-format csv -d "$OUT/p1" -- python3 "$REPO/tools/placement_probe.py" > "$OUT/p1.log" 2>&1 || echo "pass 1 failed"
rocprofv3 --pmc TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum GRBM_GUI_ACTIVE --output-format csv -d "$OUT/p2" -- python3 "$REPO/tools/placement_probe.py" > "$OUT/p2.log" 2>&1 || echo "pass 2 failed"
cd "$REPO"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for p in ("p1", "p2"):
    print("==", p)
    print(open(f"{out}/{p}.log").read().strip().split("\n")[-14:])
    rows = collections.defaultdict(dict)
    for f in glob.glob(f"{out}/{p}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "sim_ahead_kernel" not in r["Kernel_Name"]: continue
            rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(rows)
    # 18 launches per placement (3 warm + 15 timed): print the mean over each group of 18
    names = sorted({k for d in rows.values() for k in d})
    print("launches", len(ids), names)
    for g in range(0, len(ids), 18):
        grp = ids[g:g + 18]
        print("  group", g // 18, " ".join(f"{n}={sum(rows[i].get(n, 0) for i in grp) / len(grp):.3e}" for n in names))
PY
