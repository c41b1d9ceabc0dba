# usage (ON THE GPU BOX): bash tools/r4_pmc.sh <tag> "<traffic_probe args>" "<kernel name substring>" [ENV=VAL ...]
# Memory-side + SQ counter passes (one rocprofv3 --pmc run per set) over the launches of tools/traffic_probe.py.
set -u
TAG=$1; ARGS=$2; KSUB=$3; shift 3
for kv in "$@"; do export "$kv"; done
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/pmc_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_READ_sum" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $set --output-format csv -d "$OUT/p$i" -- python3 "$REPO/tools/traffic_probe.py" $ARGS > "$OUT/p$i.log" 2>&1 || echo "pass $i ($set) failed"
done
cd "$REPO"
python3 - "$OUT" "$KSUB" <<'PY'
import csv, glob, sys, collections
out, ksub = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(list)
for f in glob.glob(f"{out}/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if ksub not in r["Kernel_Name"]: continue
        tot[r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(f"{out}/summary.txt", "w") as fh:
    for k in sorted(tot):
        v = tot[k]
        line = f"{k:32s} {sum(v)/len(v):.4e}  (n={len(v)})"
        print(line); fh.write(line + "\n")
PY
