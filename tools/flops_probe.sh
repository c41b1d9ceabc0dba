# usage (ON THE GPU BOX): bash tools/flops_probe.sh "<workload list>" — floating-point instruction counters of the trajectory kernel
# (SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F32/F64, wave-level counts) -> flops per env-step and per algorithmic byte
set -u
REPO=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
for w in $1; do
  OUT=$REPO/gpurun_out/flops_$w
  rm -rf "$OUT"; mkdir -p "$OUT"
  cd /tmp
  rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 --output-format csv -d "$OUT/p1" -- python3 "$REPO/tools/traffic_probe.py" --workload $w > "$OUT/p1.log" 2>&1 || echo "pass 1 failed for $w"
  rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 --output-format csv -d "$OUT/p2" -- python3 "$REPO/tools/traffic_probe.py" --workload $w > "$OUT/p2.log" 2>&1 || echo "pass 2 failed for $w"
  rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d "$OUT/p3" -- python3 "$REPO/tools/traffic_probe.py" --workload $w > "$OUT/p3.log" 2>&1 || echo "pass 3 failed for $w"
  cd "$REPO"
  python3 - "$OUT" "$w" <<'PY'
import csv, glob, sys, collections
sys.path.insert(0, ".")
import bench
out, w = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sim_ahead_kernel" not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
c = {k: acc[k] / n[k] for k in acc}
reg, solver, dtype, tau, log2b, chunk = bench.WORKLOADS[w]
steps = (1 << log2b) * chunk
lanes = 64
g = lambda k: c.get(k, 0.0)
f32 = (g("SQ_INSTS_VALU_ADD_F32") + g("SQ_INSTS_VALU_MUL_F32") + 2 * g("SQ_INSTS_VALU_FMA_F32") + g("SQ_INSTS_VALU_TRANS_F32")) * lanes
f64 = (g("SQ_INSTS_VALU_ADD_F64") + g("SQ_INSTS_VALU_MUL_F64") + 2 * g("SQ_INSTS_VALU_FMA_F64") + g("SQ_INSTS_VALU_TRANS_F64")) * lanes
print(w, {k: "%.3e" % v for k, v in sorted(c.items())})
print(f"{w}: fp32 flops/env-step {f32 / steps:.1f}, fp64 flops/env-step {f64 / steps:.1f}, VALU lane-instr/env-step {g('SQ_INSTS_VALU') * lanes / steps:.1f} (packed instructions count once; a packed fp32 op carries two results)")
PY
done
