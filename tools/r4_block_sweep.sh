# round 4: workgroup size / per-row barrier per workload, same-buffers A/B (tools/ab_same_buffers.py; libs by tools/build_variant.sh)
mkdir -p gpurun_out/blk
for w in ${WL:-pendulum_euler_f32 msd_euler_f32 tank_euler_f32 cartpole_euler_f32 acrobot_euler_f32 msd_tsit5_f64 pmsm_euler_f64 pmsm_tsit5_f32 pmsm_rk4_f32 pmsm_euler_f32}; do
  timeout -k 10 240 python tools/ab_same_buffers.py --rounds 2 --workload $w ${LIBS:-rb512 b512 rb1024} > gpurun_out/blk/$w.txt 2>&1
  echo "== $w: $(tail -n 5 gpurun_out/blk/$w.txt | tr '\n' '|')"
done
