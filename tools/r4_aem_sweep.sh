# round 4: row-major actions, every workload: lane-major actions | fused (in-tree, NP=4) | fused NP=2 | transposition pass
mkdir -p gpurun_out/r4b
WL=${WL:-"pendulum_euler_f32 msd_tsit5_f64 cartpole_euler_f32 tank_euler_f32 pmsm_euler_f64 pmsm_tsit5_f32 acrobot_euler_f32 msd_euler_f32 pmsm_euler_f32"}
for w in $WL; do
  python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline --no-live-traffic > gpurun_out/r4b/${w}_lane.json 2>> gpurun_out/r4b/err.txt
  python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline --no-live-traffic --action-layout env_major > gpurun_out/r4b/${w}_np4.json 2>> gpurun_out/r4b/err.txt
  EXCENV_HIP_LIB=$PWD/ab_libs/libexcenv_np2.so python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline --no-live-traffic --action-layout env_major > gpurun_out/r4b/${w}_np2.json 2>> gpurun_out/r4b/err.txt
  EXCENV_AEM=0 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline --no-live-traffic --action-layout env_major > gpurun_out/r4b/${w}_transp.json 2>> gpurun_out/r4b/err.txt
  python - $w <<'PY'
import json,sys
w=sys.argv[1]; out=[]
for v in ("lane","np4","np2","transp"):
    try:
        r=json.load(open(f"gpurun_out/r4b/{w}_{v}.json")); ks=sorted(r["roofline"].get("kernel_ms_per_step") or [r["ms_per_step"]])
        out.append("%s %.3f (min %.3f) frac %.3f"%(v, r["ms_per_step"], ks[0], r["roofline"]["frac"]))
    except Exception as e: out.append(f"{v} ERR {e}")
print("%-22s"%w, " | ".join(out), flush=True)
PY
done
