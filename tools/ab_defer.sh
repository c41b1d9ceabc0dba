set -e
mkdir -p gpurun_out/ab_defer
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_gym.py tests/test_gpu_saturated.py -m gpu -x -q > gpurun_out/ab_defer/tests.log 2>&1 || { tail -30 gpurun_out/ab_defer/tests.log; exit 1; }
tail -2 gpurun_out/ab_defer/tests.log
for w in acrobot_tsit5_f32 pmsm_tsit5_f32 pmsm_rk4_f32 pmsm_euler_f32 pendulum_euler_f32 cartpole_euler_f32 acrobot_euler_f32; do
  for rep in 1 2; do
    EXCENV_HIP_LIB=$PWD/build/ab/nodefer/libexcenv_nodefer.so python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/ab_defer/${w}_base_$rep.json 2>> gpurun_out/ab_defer/err.txt
    python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/ab_defer/${w}_new_$rep.json 2>> gpurun_out/ab_defer/err.txt
  done
  python - <<PY
import json
for v in ("base","new"):
    r=[json.load(open(f"gpurun_out/ab_defer/${w}_%s_%d.json"%(v,i))) for i in (1,2)]
    print("$w", v, ["%.3f ms frac %.3f"%(x["ms_per_step"],x["roofline"]["frac"]) for x in r])
PY
done
