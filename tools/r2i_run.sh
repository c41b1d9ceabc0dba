set -e
mkdir -p gpurun_out/r2i
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "invariant_division" > gpurun_out/r2i/tests_div.log 2>&1 || { tail -40 gpurun_out/r2i/tests_div.log; exit 1; }
python -m pytest tests -m gpu -x -q > gpurun_out/r2i/tests.log 2>&1 || { tail -60 gpurun_out/r2i/tests.log; exit 1; }
tail -3 gpurun_out/r2i/tests.log
for w in pmsm_euler_f32 msd_tsit5_f64 pendulum_euler_f32 pmsm_tsit5_f32 pmsm_rk4_f32 msd_euler_f32 cartpole_euler_f32 tank_euler_f32 pmsm_euler_f64 acrobot_tsit5_f32; do
  python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r2i/$w.json 2>> gpurun_out/r2i/err.txt
done
python bench.py --path step --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r2i/step_pmsm.json 2>> gpurun_out/r2i/err.txt
