# usage (ON THE GPU BOX): bash tools/emr_pmc.sh <tag> — SQ counter passes over the fused env-major kernels (both forms)
set -u
TAG=${1:-emr}
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/emrpmc_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
ARGS="--traj-layout env_major --action-layout env_major"
i=0
for set in "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $set --output-format csv -d "$OUT/p$i" -- python3 "$REPO/tools/traffic_probe.py" $ARGS > "$OUT/p$i.log" 2>&1 || echo "pass $i failed"
done
cd "$REPO"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tot = collections.defaultdict(list)
for f in glob.glob(f"{out}/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sim_ahead_em" not in r["Kernel_Name"]: continue
        tot[(r["Kernel_Name"][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k in sorted(tot):
    v = tot[k]
    print(f"{k[0]:42s} {k[1]:32s} {sum(v)/len(v):.4e}  (n={len(v)})")
PY
