# usage (ON THE GPU BOX): bash tools/vec_sweep.sh "<workloads>" "<vec list>"  — same-session sweep of environments per lane
set -e
mkdir -p gpurun_out/vec
for w in $1; do
  for v in $2; do
    python bench.py --workload $w --vec $v --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/vec/${w}_v$v.json 2>> gpurun_out/vec/err.txt
    python - <<PY
import json
x=json.load(open("gpurun_out/vec/${w}_v$v.json"))
print("%-22s V=%s  %.3f ms frac %.3f"%("$w","$v",x["ms_per_step"],x["roofline"]["frac"]))
PY
  done
done
