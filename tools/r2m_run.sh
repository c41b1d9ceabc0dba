set -e
mkdir -p gpurun_out/r2m
PREV=$PWD/build/ab/libexcenv_hip_prev.so
python -m pytest tests -m gpu -x -q > gpurun_out/r2m/tests.log 2>&1 || { tail -60 gpurun_out/r2m/tests.log; exit 1; }
tail -3 gpurun_out/r2m/tests.log
for w in pmsm_euler_f32 pmsm_tsit5_f32 pmsm_rk4_f32 msd_tsit5_f64 cartpole_euler_f32 acrobot_tsit5_f32 pendulum_euler_f32; do
  python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r2m/new_$w.json 2>> gpurun_out/r2m/err.txt
  EXCENV_HIP_LIB=$PREV python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r2m/prev_$w.json 2>> gpurun_out/r2m/err.txt
done
