# usage (ON THE GPU BOX, repo root): bash tools/r4_wide_pmc.sh — memory-side and SQ counters of the C2 launch (pendulum Euler fp32,
# B = 2^20, 1000 rows) in its 1024-thread form (one barrier per row) and in the 256-thread form (EXCENV_WIDE=0). One rocprofv3
# --pmc pass per counter set and form over tools/traffic_probe.py.
set -u
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/wide_pmc
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for form in wide narrow; do
  if [ $form = narrow ]; then export EXCENV_WIDE=0; else unset EXCENV_WIDE; fi
  i=0
  for set in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" \
             "TCC_EA0_WRREQ_LEVEL_sum TCC_TAG_STALL_sum TCC_BUSY_sum GRBM_GUI_ACTIVE" \
             "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU"; do
    i=$((i+1))
    timeout -k 10 150 rocprofv3 --pmc $set --output-format csv -d "$OUT/${form}_p$i" -- python3 "$REPO/tools/traffic_probe.py" --workload pendulum_euler_f32 > "$OUT/${form}_p$i.log" 2>&1 || echo "$form pass $i failed"
  done
done
cd "$REPO"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
with open(f"{out}/summary.txt", "w") as fh:
    for form in ("wide", "narrow"):
        tot = collections.defaultdict(list)
        for f in glob.glob(f"{out}/{form}_p*/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "sim_ahead_kernel" not in r["Kernel_Name"]: continue
                tot[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k in sorted(tot):
            line = f"{form:7s} {k:40s} {sum(tot[k]) / len(tot[k]):.4e}  (n={len(tot[k])})"
            print(line); fh.write(line + "\n")
PY
