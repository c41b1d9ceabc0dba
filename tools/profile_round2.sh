# Round-2 measurement session (run ON THE GPU BOX from the repo root: gpurun -- bash tools/profile_round2.sh): per BASELINE
# config the un-profiled bench line + rocprofv3 kernel trace + PMC passes in ONE session, then the workload table. Summaries:
# python tools/summarize_profile.py gpurun_out/prof_<tag> profiles <tag> "<traffic key>".
set -e
mkdir -p gpurun_out/profile_round2
bash tools/profile_gpu.sh r02_c3_pmsm_euler_f32 > gpurun_out/profile_round2/prof_c3.log 2>&1
bash tools/profile_gpu.sh r02_c2_pendulum_euler_f32 --workload pendulum_euler_f32 > gpurun_out/profile_round2/prof_c2.log 2>&1
bash tools/profile_gpu.sh r02_c4_msd_tsit5_f64 --workload msd_tsit5_f64 > gpurun_out/profile_round2/prof_c4.log 2>&1
bash tools/profile_gpu.sh r02_pmsm_tsit5_f32 --workload pmsm_tsit5_f32 > gpurun_out/profile_round2/prof_pt.log 2>&1
bash tools/profile_gpu.sh r02_acrobot_tsit5_f32 --workload acrobot_tsit5_f32 > gpurun_out/profile_round2/prof_at.log 2>&1
bash tools/profile_gpu.sh r02_pmsm_step --path step > gpurun_out/profile_round2/prof_step.log 2>&1
python bench.py > gpurun_out/profile_round2/bench_default.json 2> gpurun_out/profile_round2/bench_default.err
for w in msd_euler_f32 tank_euler_f32 cartpole_euler_f32 acrobot_euler_f32 pmsm_euler_f64 pmsm_rk4_f32 pmsm_sat_euler_f32 pmsm_sat_tsit5_f32; do
  python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/profile_round2/$w.json 2>> gpurun_out/profile_round2/err.txt
done
python bench.py --obs-only --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/profile_round2/pmsm_obsonly.json 2>> gpurun_out/profile_round2/err.txt
bash tools/profile_gpu.sh r02_em_pmsm_euler_f32 --traj-layout env_major --action-layout env_major > gpurun_out/profile_round2/prof_em.log 2>&1
for w in pmsm_tsit5_f32 acrobot_tsit5_f32 pendulum_euler_f32 msd_tsit5_f64 pmsm_sat_euler_f64; do
  python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/profile_round2/$w.json 2>> gpurun_out/profile_round2/err.txt
done
python tools/host_overhead.py > gpurun_out/profile_round2/host_overhead.txt 2>&1
python tools/batch_sweep.py > gpurun_out/profile_round2/batch_sweep.txt 2>&1
