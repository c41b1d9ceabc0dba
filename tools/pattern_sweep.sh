# usage (ON THE GPU BOX, repo root): bash tools/pattern_sweep.sh — the access-shape sweep of tools/microbench/pattern_sweep.hip (built here by
# hipcc before the push), twice, then memory-side counter passes over a few variants (one rocprofv3 --pmc run per variant and set).
set -u
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/sweep
mkdir -p "$OUT"
BIN=$REPO/tools/microbench/pattern_sweep
timeout -k 10 300 $BIN > "$OUT/sweep_a.txt" 2> "$OUT/err_a.txt" || { echo "sweep failed"; exit 1; }
timeout -k 10 300 $BIN > "$OUT/sweep_b.txt" 2> "$OUT/err_b.txt" || { echo "sweep failed"; exit 1; }
export TMPDIR=/tmp
cd /tmp
i=0
for only in "base wg256" "vpl2 (2 KiB" "wg1024 sync per row" "obs [8][K+1][B] (component" "obs [8][K+1][B] vpl2" "tiled"; do
  i=$((i+1))
  for set in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" "TCC_EA0_WRREQ_LEVEL_sum TCC_TAG_STALL_sum TCC_BUSY_sum GRBM_GUI_ACTIVE"; do
    d="$OUT/pmc_${i}_$(echo $set | cut -c9-20 | tr -d ' ')"
    timeout -k 10 120 rocprofv3 --pmc $set --output-format csv -d "$d" -- $BIN --only "$only" --reps 2 > "$d.log" 2>&1 || echo "pmc pass $i failed"
  done
done
cd "$REPO"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
names = ["base wg256", "vpl2", "wg1024 sync per row", "obs [8][K+1][B]", "obs [8][K+1][B] vpl2", "tiled"]
with open(f"{out}/pmc_summary.txt", "w") as fh:
    for i, n in enumerate(names, 1):
        tot = collections.defaultdict(list)
        for f in glob.glob(f"{out}/pmc_{i}_*/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "pat" not in r["Kernel_Name"] or "fill" in r["Kernel_Name"]: continue
                tot[r["Counter_Name"]].append(float(r["Counter_Value"]))
        line = f"{n:28s} " + "  ".join(f"{k}={sum(v)/len(v):.4e}" for k, v in sorted(tot.items()))
        print(line); fh.write(line + "\n")
PY
