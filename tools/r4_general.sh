set -e
mkdir -p gpurun_out/r4g
python -m pytest tests/test_gpu_parity.py tests/test_gpu_trajectory_pool.py tests/test_gpu_gym.py tests/test_gpu_fuzz.py -x -q > gpurun_out/r4g/tests.log 2>&1 || { tail -30 gpurun_out/r4g/tests.log; exit 1; }
tail -2 gpurun_out/r4g/tests.log
python tools/general_path_cost.py > gpurun_out/r4g/general_path_cost.txt 2>&1 || true
cat gpurun_out/r4g/general_path_cost.txt
python bench.py --workload pmsm_euler_f64 --steps 20 --warmup 3 --no-cpu-baseline --no-live-traffic --traj-layout env_major --action-layout env_major > gpurun_out/r4g/f64_em.json 2>> gpurun_out/r4g/err.txt
for i in 1 2; do python bench.py --no-cpu-baseline --no-live-traffic --steps 20 > gpurun_out/r4g/bench_$i.json 2>> gpurun_out/r4g/err.txt; done
python - <<'PY'
import json
r=json.load(open("gpurun_out/r4g/f64_em.json")); print("pmsm_euler_f64 env-major", r["ms_per_step"], r["roofline"]["frac"])
for f in ("1","2"):
    r=json.load(open(f"gpurun_out/r4g/bench_{f}.json")); ks=r["roofline"]["kernel_ms_per_step"]
    print(f, round(r["ms_per_step"],3), round(r["roofline"]["frac"],3), "spread %.3f"%((max(ks)-min(ks))/sorted(ks)[len(ks)//2]), r["config"].get("placement_settle_steps"), r["config"].get("pooled_set_steady_ms"), [round(k,2) for k in ks[:6]])
PY
