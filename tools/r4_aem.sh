# round 4: row-major actions fused into the lane-major kernel (AEM) — parity tests, then same-session A/B of the headline launch
set -e
mkdir -p gpurun_out/r4a
python -m pytest tests/test_gpu_fused_actions.py -x -q > gpurun_out/r4a/tests.log 2>&1 || { tail -40 gpurun_out/r4a/tests.log; exit 1; }
tail -3 gpurun_out/r4a/tests.log
for rep in 1 2; do
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-live-traffic > gpurun_out/r4a/lane_$rep.json 2>> gpurun_out/r4a/err.txt
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-live-traffic --action-layout env_major > gpurun_out/r4a/aem_$rep.json 2>> gpurun_out/r4a/err.txt
  EXCENV_AEM=0 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-live-traffic --action-layout env_major > gpurun_out/r4a/transp_$rep.json 2>> gpurun_out/r4a/err.txt
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4a/*.json")):
    try:
        r=json.load(open(f)); print(f.split("/")[-1], "ms/step %.3f"%r["ms_per_step"], "frac %.3f"%r["roofline"]["frac"], "kernel_ms", r["roofline"].get("kernel_ms"))
    except Exception as e: print(f, "ERR", e)
PY
