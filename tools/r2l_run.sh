set -e
mkdir -p gpurun_out/r2l
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r2l/headline.json 2>> gpurun_out/r2l/err.txt
for w in pmsm_tsit5_f32 pmsm_rk4_f32 acrobot_tsit5_f32 pendulum_euler_f32; do for v in 1 2 4; do
  python bench.py --workload $w --vec $v --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r2l/${w}_v$v.json 2>> gpurun_out/r2l/err.txt
done; done
for v in 1 2; do python bench.py --workload msd_tsit5_f64 --vec $v --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r2l/msd_tsit5_f64_v$v.json 2>> gpurun_out/r2l/err.txt; done
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r2l/headline2.json 2>> gpurun_out/r2l/err.txt
