set -e
mkdir -p gpurun_out/r2b
python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -x -q --durations=8 > gpurun_out/r2b/tests.log 2>&1 || { tail -60 gpurun_out/r2b/tests.log; exit 1; }
tail -25 gpurun_out/r2b/tests.log
python bench.py --steps 30 --warmup 5 > gpurun_out/r2b/bench.json 2> gpurun_out/r2b/bench.err
for w in msd_tsit5_f64 pendulum_euler_f32 pmsm_tsit5_f32; do
  python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r2b/bench_$w.json 2>> gpurun_out/r2b/bench.err
done
EXCENV_BENCH_ONE_GPU=1 EXCENV_BENCH_BACKEND=gloo python bench.py --gpus 2 --batch 1048576 --steps 10 --warmup 2 > gpurun_out/r2b/bench_2rank_gloo.json 2> gpurun_out/r2b/bench_2rank.err
cat gpurun_out/r2b/bench_2rank_gloo.json
