#!/bin/bash
# one PMC pass: bash scripts/pmc_once.sh <tag> "<counters>" [bench args]
TAG=$1; CNT=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"; export TMPDIR=/tmp
# the profiler starts HIP before Python runs: the queue count must be in the environment already
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-8}; cd /tmp
timeout -s KILL 200 rocprofv3 --pmc $CNT --output-format csv -d "$OUT" -- python3 "$ROOT/bench.py" --steps 5 --warmup 1 --sustain-s 0 --no-cpu-baseline --hw-queues 0 "$@" > "$OUT/bench.json" 2> "$OUT/err.txt" || { echo FAILED; tail -3 "$OUT/err.txt"; exit 1; }
python3 - "$OUT" <<'PY'
import csv,glob,collections,sys
f=glob.glob(sys.argv[1]+"/*/*_counter_collection.csv")[0]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"].split("(")[0][:40]
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    if "k_wf_level" in k or "k_render" in k:
        print(k, {c:"%.4g"%(sum(x)/len(x)) for c,x in v.items()}, "launches", len(list(v.values())[0]))
PY
