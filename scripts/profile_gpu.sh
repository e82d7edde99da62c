#!/bin/bash
# Collect the rocprofv3 evidence for bench.py's default workload:
#   1. kernel trace + stats, 2..n. one PMC pass each (counters cannot share passes freely).
# Usage (on the GPU box, from the repo root): bash scripts/profile_gpu.sh <tag> [bench args]
set -u
TAG=${1:-r01}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
# the profiler starts HIP before Python runs: the queue count must be in the environment already
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-8}
cd /tmp
ARGS="--steps 10 --warmup 2 --sustain-s 0 --no-cpu-baseline --hw-queues 0 $*"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/stats_bench.json" 2> "$OUT/stats.err" || { echo "stats pass failed"; tail -5 "$OUT/stats.err"; exit 1; }
if [ "${STATS_ONLY:-0}" = "1" ]; then echo done; exit 0; fi
for PASS in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "GRBM_GUI_ACTIVE"; do
  NAME=$(echo $PASS | tr ' ' '_' | cut -c1-40)
  timeout -s KILL 300 rocprofv3 --pmc $PASS --output-format csv -d "$OUT/pmc_$NAME" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_$NAME.json" 2> "$OUT/pmc_$NAME.err" || { echo "pmc pass $PASS failed"; tail -5 "$OUT/pmc_$NAME.err"; exit 1; }
done
echo done
