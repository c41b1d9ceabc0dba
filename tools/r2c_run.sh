set -e
mkdir -p gpurun_out/r2c
python -m pytest tests/test_gpu_stepper.py tests/test_gpu_gym.py tests/test_gpu_parity.py tests/test_gpu_saturated.py -m gpu -x -q > gpurun_out/r2c/tests.log 2>&1 || { tail -60 gpurun_out/r2c/tests.log; exit 1; }
tail -5 gpurun_out/r2c/tests.log
python tools/host_overhead.py > gpurun_out/r2c/host_overhead.txt 2>&1
cat gpurun_out/r2c/host_overhead.txt | grep -v "^ \|^$" | head -60
