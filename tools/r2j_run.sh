set -e
mkdir -p gpurun_out/r2j
python -m pytest tests -m gpu -x -q > gpurun_out/r2j/tests.log 2>&1 || { tail -60 gpurun_out/r2j/tests.log; exit 1; }
tail -3 gpurun_out/r2j/tests.log
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r2j/headline.json 2>> gpurun_out/r2j/err.txt
for v in 1 2 4; do python bench.py --path step --steps 50 --warmup 5 --no-cpu-baseline --vec $v > gpurun_out/r2j/step_pmsm_v$v.json 2>> gpurun_out/r2j/err.txt; done
for w in pendulum_euler_f32 msd_tsit5_f64 cartpole_euler_f32; do python bench.py --path step --workload $w --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r2j/step_$w.json 2>> gpurun_out/r2j/err.txt; done
python tools/batch_sweep.py --log2 10 14 16 18 20 22 24 > gpurun_out/r2j/sweep.md 2>&1
python tools/host_overhead.py 2>&1 | grep "us per" > gpurun_out/r2j/host.txt
