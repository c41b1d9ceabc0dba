for w in pmsm_tsit5_f32 pmsm_rk4_f32 pmsm_euler_f64 cartpole_euler_f32 acrobot_euler_f32 msd_tsit5_f64 pendulum_tsit5_f32; do
  for v in 0 2 1; do
    python bench.py --workload $w --obs-only --vec $v --steps 30 --warmup 3 --no-cpu-baseline --no-live-traffic --no-calibration 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$w', 'vec', $v, round(d['ms_per_step'],3), round(d['roofline']['frac'],3))"
  done
done
