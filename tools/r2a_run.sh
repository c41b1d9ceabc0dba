set -e
mkdir -p gpurun_out/r2a
PP=$PWD/build/ab/libexcenv_hip_pp.so
python bench.py --steps 30 --warmup 5 > gpurun_out/r2a/bench_sp.json 2> gpurun_out/r2a/bench_sp.err
EXCENV_HIP_LIB=$PP python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r2a/bench_pp.json 2> gpurun_out/r2a/bench_pp.err
python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r2a/bench_sp2.json 2>> gpurun_out/r2a/bench_sp.err
python tools/batch_sweep.py --sim-only > gpurun_out/r2a/sweep_sp_auto.md 2>&1
python tools/batch_sweep.py --sim-only --vec 1 --log2 10 14 16 17 18 19 20 > gpurun_out/r2a/sweep_sp_v1.md 2>&1
python tools/batch_sweep.py --sim-only --vec 4 --log2 10 14 16 17 18 19 20 > gpurun_out/r2a/sweep_sp_v4.md 2>&1
EXCENV_HIP_LIB=$PP python tools/batch_sweep.py --sim-only > gpurun_out/r2a/sweep_pp_auto.md 2>&1
EXCENV_HIP_LIB=$PP python tools/batch_sweep.py --sim-only --vec 1 --log2 10 14 16 17 18 19 20 > gpurun_out/r2a/sweep_pp_v1.md 2>&1
for w in msd_tsit5_f64 pendulum_euler_f32 pmsm_tsit5_f32 acrobot_tsit5_f32; do
  python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r2a/bench_sp_$w.json 2>> gpurun_out/r2a/bench_sp.err
  EXCENV_HIP_LIB=$PP python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r2a/bench_pp_$w.json 2>> gpurun_out/r2a/bench_pp.err
done
python -m pytest tests -m gpu -x -q > gpurun_out/r2a/tests.log 2>&1
tail -3 gpurun_out/r2a/tests.log
