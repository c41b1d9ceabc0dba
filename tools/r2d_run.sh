set -e
mkdir -p gpurun_out/r2d
python -m pytest tests/test_gpu_gym.py tests/test_gpu_stepper.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r2d/tests.log 2>&1 || { tail -80 gpurun_out/r2d/tests.log; exit 1; }
tail -5 gpurun_out/r2d/tests.log
python tools/host_overhead.py 2>&1 | grep -E "us per" > gpurun_out/r2d/host_overhead.txt
cat gpurun_out/r2d/host_overhead.txt
