#!/bin/bash
# Run ON THE GPU BOX from the repo root: tools/profile_gpu.sh <tag> [extra bench/probe args...]
# kernel-trace/--stats and the two PMC counters are collected in separate rocprofv3 runs (gpurun refuses combined
# trace domains with --pmc). Raw output stays under gpurun_out/ (scratch); the summary goes to profiles/.
set -u
TAG=${1:-r01}; shift || true
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
# the un-profiled bench line of the same session (same box, same minute) — bench.py's own --steps / --warmup defaults
python3 "$REPO/bench.py" --no-cpu-baseline "$@" > "$OUT/bench.json" 2> "$OUT/bench.err" || echo "bench run failed"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/bench.py" --steps 20 --warmup 2 --no-cpu-baseline "$@" > "$OUT/bench_under_trace.json" 2> "$OUT/trace.err" || echo "trace run failed"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$REPO/tools/traffic_probe.py" "$@" > "$OUT/pmc_fetch.log" 2>&1 || echo "pmc fetch run failed"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$REPO/tools/traffic_probe.py" "$@" > "$OUT/pmc_write.log" 2>&1 || echo "pmc write run failed"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT/pmc_sq" -- python3 "$REPO/tools/traffic_probe.py" "$@" > "$OUT/pmc_sq.log" 2>&1 || echo "pmc sq run failed"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d "$OUT/pmc_sq2" -- python3 "$REPO/tools/traffic_probe.py" "$@" > "$OUT/pmc_sq2.log" 2>&1 || echo "pmc sq2 run failed"
cd "$REPO"
find "$OUT" -name "*.csv" | head -20
