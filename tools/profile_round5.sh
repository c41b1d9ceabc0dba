# Round-5 measurement session (run ON THE GPU BOX from the repo root: gpurun -- bash tools/profile_round5.sh <part>): per
# BASELINE config the un-profiled bench line + rocprofv3 kernel trace + PMC passes in ONE session. Summaries:
# python tools/summarize_profile.py gpurun_out/prof_<tag> profiles <tag> "<traffic key>".
set -e
PART=${1:-a}
mkdir -p gpurun_out/profile_round5
if [ "$PART" = a ]; then
  bash tools/profile_gpu.sh r05_c3_pmsm_euler_f32 > gpurun_out/profile_round5/prof_c3.log 2>&1
  bash tools/profile_gpu.sh r05_c2_pendulum_euler_f32 --workload pendulum_euler_f32 > gpurun_out/profile_round5/prof_c2.log 2>&1
  bash tools/profile_gpu.sh r05_c4_msd_tsit5_f64 --workload msd_tsit5_f64 > gpurun_out/profile_round5/prof_c4.log 2>&1
  python bench.py > gpurun_out/profile_round5/bench_default.json 2> gpurun_out/profile_round5/bench_default.err
elif [ "$PART" = b ]; then
  bash tools/profile_gpu.sh r05_pmsm_tsit5_f32 --workload pmsm_tsit5_f32 > gpurun_out/profile_round5/prof_pt.log 2>&1
  bash tools/profile_gpu.sh r05_pmsm_rk4_f32 --workload pmsm_rk4_f32 > gpurun_out/profile_round5/prof_rk4.log 2>&1
  bash tools/profile_gpu.sh r05_pmsm_step --path step > gpurun_out/profile_round5/prof_step.log 2>&1
  bash tools/profile_gpu.sh r05_em_pmsm_euler_f32 --traj-layout env_major --action-layout env_major > gpurun_out/profile_round5/prof_em.log 2>&1
  # the reference-shaped INPUT with the library's default outputs: a plain row-major actions[B, K, A] tensor read by the trajectory
  # kernel itself (round 4: 64-byte windows through LDS, no transposition pass)
  bash tools/profile_gpu.sh r05_c3_rowmajor_actions --action-layout env_major > gpurun_out/profile_round5/prof_c3_rm.log 2>&1
else
  for w in msd_euler_f32 tank_euler_f32 cartpole_euler_f32 acrobot_euler_f32 pmsm_euler_f64 pmsm_rk4_f32 pmsm_tsit5_f32 acrobot_tsit5_f32 pendulum_euler_f32 msd_tsit5_f64 pmsm_sat_euler_f32 pmsm_sat_tsit5_f32 pmsm_sat_euler_f64; do
    python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/profile_round5/$w.json 2>> gpurun_out/profile_round5/err.txt
    echo "$w done" >> gpurun_out/profile_round5/progress.txt
  done
  python bench.py --obs-only --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/profile_round5/pmsm_obsonly.json 2>> gpurun_out/profile_round5/err.txt
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pool > gpurun_out/profile_round5/pmsm_nopool.json 2>> gpurun_out/profile_round5/err.txt
  for w in pendulum_euler_f32 msd_tsit5_f64 pmsm_euler_f64 pmsm_tsit5_f32; do
    python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline --action-layout env_major > gpurun_out/profile_round5/${w}_rowmajor_actions.json 2>> gpurun_out/profile_round5/err.txt
  done
  python bench.py --path step --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/profile_round5/pmsm_step_bench.json 2>> gpurun_out/profile_round5/err.txt
  python tools/general_path_cost.py > gpurun_out/profile_round5/general_path_cost.txt 2>&1
  python tools/gym_cost.py > gpurun_out/profile_round5/gym_cost.txt 2>&1
  python tools/host_overhead.py > gpurun_out/profile_round5/host_overhead.txt 2>&1
  python tools/batch_sweep.py > gpurun_out/profile_round5/batch_sweep.txt 2>&1
fi
