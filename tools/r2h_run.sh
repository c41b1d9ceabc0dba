set -e
mkdir -p gpurun_out/r2h
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_saturated.py tests/test_gpu_gym.py -m gpu -x -q > gpurun_out/r2h/tests.log 2>&1 || { tail -50 gpurun_out/r2h/tests.log; exit 1; }
tail -3 gpurun_out/r2h/tests.log
EM="--traj-layout env_major --action-layout env_major --steps 20 --warmup 3 --no-cpu-baseline"
for w in pmsm_euler_f32 pendulum_euler_f32 msd_tsit5_f64 cartpole_euler_f32 pmsm_euler_f64; do python bench.py $EM --workload $w > gpurun_out/r2h/em_$w.json 2>> gpurun_out/r2h/err.txt; done
python bench.py $EM --obs-only > gpurun_out/r2h/em_obsonly.json 2>> gpurun_out/r2h/err.txt
bash tools/profile_gpu.sh r02_c3_pmsm_euler_f32 > gpurun_out/r2h/prof_c3.log 2>&1
bash tools/profile_gpu.sh r02_c2_pendulum_euler_f32 --workload pendulum_euler_f32 > gpurun_out/r2h/prof_c2.log 2>&1
bash tools/profile_gpu.sh r02_c4_msd_tsit5_f64 --workload msd_tsit5_f64 > gpurun_out/r2h/prof_c4.log 2>&1
bash tools/profile_gpu.sh r02_em_pmsm_euler_f32 --traj-layout env_major --action-layout env_major > gpurun_out/r2h/prof_em.log 2>&1
