set -e
mkdir -p gpurun_out/r2g
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_saturated.py -m gpu -x -q > gpurun_out/r2g/tests.log 2>&1 || { tail -50 gpurun_out/r2g/tests.log; exit 1; }
tail -3 gpurun_out/r2g/tests.log
EM="--traj-layout env_major --action-layout env_major --steps 20 --warmup 3 --no-cpu-baseline"
python bench.py $EM > gpurun_out/r2g/em_tk8.json 2> gpurun_out/r2g/err.txt
for v in tk4 tk16 skipobs skipst; do
  EXCENV_HIP_LIB=$PWD/build/ab/libexcenv_hip_$v.so python bench.py $EM > gpurun_out/r2g/em_$v.json 2>> gpurun_out/r2g/err.txt
done
for w in pendulum_euler_f32 msd_tsit5_f64 cartpole_euler_f32; do python bench.py $EM --workload $w > gpurun_out/r2g/em_$w.json 2>> gpurun_out/r2g/err.txt; done
python bench.py $EM --obs-only > gpurun_out/r2g/em_obsonly.json 2>> gpurun_out/r2g/err.txt
bash tools/profile_gpu.sh r02_em_pmsm --traj-layout env_major --action-layout env_major > gpurun_out/r2g/prof.log 2>&1
python tools/summarize_profile.py gpurun_out/prof_r02_em_pmsm gpurun_out/r2g r02_em_pmsm > /dev/null
