# usage (ON THE GPU BOX): bash tools/icache_probe.sh "<workload list>"  — instruction-cache and issue-stall counters per kernel
set -u
REPO=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
for w in $1; do
  OUT=$REPO/gpurun_out/icache_$w
  mkdir -p "$OUT"
  cd /tmp
  rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH --output-format csv -d "$OUT/p1" -- python3 "$REPO/tools/traffic_probe.py" --workload $w > "$OUT/p1.log" 2>&1 || echo "pass 1 failed for $w"
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_WR SQ_IFETCH_LEVEL --output-format csv -d "$OUT/p2" -- python3 "$REPO/tools/traffic_probe.py" --workload $w > "$OUT/p2.log" 2>&1 || echo "pass 2 failed for $w"
  cd "$REPO"
  python3 - "$OUT" "$w" <<'PY'
import csv, glob, sys, collections
out, w = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "sim_ahead" not in k: continue
        acc[k[:70]][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k[:70], r["Counter_Name"])] += 1
for k, d in acc.items():
    print(w, k)
    for c, v in sorted(d.items()): print("   %-28s %.4e" % (c, v / max(n[(k, c)], 1)))
PY
done
