set -e
mkdir -p gpurun_out/r2n
for b in 4194304 4195328 4198400 4210688 4259840 4194304; do
  python bench.py --batch $b --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r2n/b_$b.json 2>> gpurun_out/r2n/err.txt
  python3 -c "
import json
d=json.load(open('gpurun_out/r2n/b_$b.json'))
print('$b', 'ms %.4f'%d['roofline']['kernel_ms'], 'frac %.4f'%d['roofline']['frac'])"
done
for b in 1048576 1049600 1114112; do
  python bench.py --workload pendulum_euler_f32 --batch $b --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r2n/pend_$b.json 2>> gpurun_out/r2n/err.txt
  python3 -c "
import json
d=json.load(open('gpurun_out/r2n/pend_$b.json'))
print('pend $b', 'ms %.4f'%d['roofline']['kernel_ms'], 'frac %.4f'%d['roofline']['frac'])"
done
