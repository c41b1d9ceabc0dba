# round 4: register-ring env-major kernel — parity tests, then the headline launch in row-major arrays
set -e
mkdir -p gpurun_out/r4e
python -m pytest tests/test_gpu_env_major_ring.py tests/test_gpu_fuzz.py -x -q > gpurun_out/r4e/tests.log 2>&1 || { tail -40 gpurun_out/r4e/tests.log; exit 1; }
tail -2 gpurun_out/r4e/tests.log
for rep in 1 2; do
  for w in ${WL:-pmsm_euler_f32}; do
    python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline --no-live-traffic --traj-layout env_major --action-layout env_major > gpurun_out/r4e/${w}_$rep.json 2>> gpurun_out/r4e/err.txt
    python - $w $rep <<'PY'
import json,sys
r=json.load(open(f"gpurun_out/r4e/{sys.argv[1]}_{sys.argv[2]}.json")); ks=sorted(r["roofline"].get("kernel_ms_per_step") or [0])
print(sys.argv[1], "ms/step %.3f"%r["ms_per_step"], "min %.3f"%ks[0], "frac %.3f"%r["roofline"]["frac"], flush=True)
PY
  done
done
