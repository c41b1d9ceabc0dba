#!/bin/bash
# Run ON THE GPU BOX from the repo root (gpurun -- bash tools/profile_gym.sh <tag>): the fused gym trajectories of all six models,
# plain launch beside gym launch — un-profiled timing, rocprofv3 kernel trace, and PMC passes (own runs, no trace domains) for the
# bytes written, the store instructions and the share of wave time parked on s_waitcnt.
# Summary: python tools/summarize_gym.py gpurun_out/prof_<tag> profiles <tag>
set -u
TAG=${1:-r05_gym}
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
python3 "$REPO/tools/gym_probe.py" "$OUT/plain_run.jsonl" 10 1 > "$OUT/plain_run.log" 2>&1 || echo "un-profiled run failed"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/tools/gym_probe.py" "$OUT/trace_run.jsonl" 10 1 > "$OUT/trace.log" 2>&1 || echo "trace run failed"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$REPO/tools/gym_probe.py" "$OUT/pmc_write_run.jsonl" 2 0 > "$OUT/pmc_write.log" 2>&1 || echo "pmc write run failed"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$REPO/tools/gym_probe.py" "$OUT/pmc_fetch_run.jsonl" 2 0 > "$OUT/pmc_fetch.log" 2>&1 || echo "pmc fetch run failed"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT/pmc_sq" -- python3 "$REPO/tools/gym_probe.py" "$OUT/pmc_sq_run.jsonl" 2 0 > "$OUT/pmc_sq.log" 2>&1 || echo "pmc sq run failed"
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq2" -- python3 "$REPO/tools/gym_probe.py" "$OUT/pmc_sq2_run.jsonl" 2 0 > "$OUT/pmc_sq2.log" 2>&1 || echo "pmc sq2 run failed"
cd "$REPO"
find "$OUT" -name "*.csv" | head -20
