# usage (ON THE GPU BOX, repo root): bash tools/r4_rowsync_pmc.sh — counters of the headline launch forced to ONE environment per lane
# (4-byte stores) with the rows leaving directly (EXCENV_ROW_SYNC=0), behind a barrier (=1) and through LDS as 16-byte stores (=2).
set -u
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/rowsync_pmc
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for mode in 0 1 2; do
  export EXCENV_ROW_SYNC=$mode
  i=0
  for set in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" \
             "TCC_EA0_WRREQ_LEVEL_sum TCC_TAG_STALL_sum TCC_REQ_sum GRBM_GUI_ACTIVE" \
             "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_LDS"; do
    i=$((i+1))
    timeout -k 10 150 rocprofv3 --pmc $set --output-format csv -d "$OUT/m${mode}_p$i" -- python3 "$REPO/tools/traffic_probe.py" --vec 1 > "$OUT/m${mode}_p$i.log" 2>&1 || echo "mode $mode pass $i failed"
  done
done
cd "$REPO"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
with open(f"{out}/summary.txt", "w") as fh:
    for mode in "012":
        tot = collections.defaultdict(list)
        for f in glob.glob(f"{out}/m{mode}_p*/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "sim_ahead_kernel" not in r["Kernel_Name"]: continue
                tot[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k in sorted(tot):
            line = f"row_sync={mode} {k:40s} {sum(tot[k]) / len(tot[k]):.4e}  (n={len(tot[k])})"
            print(line); fh.write(line + "\n")
PY
