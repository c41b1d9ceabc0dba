set -e
mkdir -p gpurun_out/r2e
TK4=$PWD/build/ab/libexcenv_hip_tk4.so
EM="--traj-layout env_major --action-layout env_major --steps 20 --warmup 3 --no-cpu-baseline"
python bench.py $EM > gpurun_out/r2e/em_tk8.json 2> gpurun_out/r2e/err.txt
EXCENV_HIP_LIB=$TK4 python bench.py $EM > gpurun_out/r2e/em_tk4.json 2>> gpurun_out/r2e/err.txt
python bench.py $EM --no-fused > gpurun_out/r2e/em_ws.json 2>> gpurun_out/r2e/err.txt
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r2e/lane.json 2>> gpurun_out/r2e/err.txt
for v in 1 2; do python bench.py --workload msd_tsit5_f64 --vec $v --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r2e/msd_v$v.json 2>> gpurun_out/r2e/err.txt; done
for v in 1 2 4; do python bench.py --workload pmsm_tsit5_f32 --vec $v --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r2e/pmsm_tsit5_v$v.json 2>> gpurun_out/r2e/err.txt; done
for v in 1 2 4; do python bench.py --workload acrobot_tsit5_f32 --vec $v --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r2e/acro_tsit5_v$v.json 2>> gpurun_out/r2e/err.txt; done
for v in 1 2 4; do python bench.py --workload pmsm_rk4_f32 --vec $v --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r2e/pmsm_rk4_v$v.json 2>> gpurun_out/r2e/err.txt; done
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/r2e/tests.log 2>&1 || { tail -50 gpurun_out/r2e/tests.log; exit 1; }
tail -3 gpurun_out/r2e/tests.log
