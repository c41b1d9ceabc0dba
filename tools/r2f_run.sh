set -e
mkdir -p gpurun_out/r2f
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_saturated.py -m gpu -x -q > gpurun_out/r2f/tests.log 2>&1 || { tail -50 gpurun_out/r2f/tests.log; exit 1; }
tail -3 gpurun_out/r2f/tests.log
EM="--traj-layout env_major --action-layout env_major --steps 20 --warmup 3 --no-cpu-baseline"
python bench.py $EM > gpurun_out/r2f/em_tk8.json 2> gpurun_out/r2f/err.txt
python bench.py $EM --workload pendulum_euler_f32 > gpurun_out/r2f/em_pend.json 2>> gpurun_out/r2f/err.txt
python bench.py $EM --workload msd_tsit5_f64 > gpurun_out/r2f/em_msd.json 2>> gpurun_out/r2f/err.txt
python bench.py $EM --workload cartpole_euler_f32 > gpurun_out/r2f/em_cartpole.json 2>> gpurun_out/r2f/err.txt
python bench.py $EM --obs-only > gpurun_out/r2f/em_obsonly.json 2>> gpurun_out/r2f/err.txt
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r2f/lane.json 2>> gpurun_out/r2f/err.txt
