#!/bin/bash
# host SAH builder variants: bash scripts/sah_ab.sh  (WORKLOADS="..." to choose)
run() {  # label, env...
  L=$1; shift
  for W in ${WORKLOADS:-pawn1080 teaset1080}; do
    env "$@" timeout -k 10 200 python bench.py --workload $W --bvh sah --no-cpu-baseline --sustain-s 0 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L $W', round(d['ms_per_step'],3), round(d['single_frame']['ms'],3), d['config']['frame_crc32'], d['counters_per_frame']['nodes_visited'], d['counters_per_frame']['tri_tests'], round(d['config']['commit_ms'],1), flush=True)" || exit 1
  done
}
run base RT_X=0
run bottomup RT_SAH_BOTTOMUP=1
run bottomup+sweep64 RT_SAH_BOTTOMUP=1 RT_SAH_SWEEP=64
run bottomup+sweepall RT_SAH_BOTTOMUP=1 RT_SAH_SWEEP=100000000
run bottomup+sweepall+trav1.5 RT_SAH_BOTTOMUP=1 RT_SAH_SWEEP=100000000 RT_SAH_TRAV=1.5
run bottomup+sweepall+trav2.5 RT_SAH_BOTTOMUP=1 RT_SAH_SWEEP=100000000 RT_SAH_TRAV=2.5
run bottomup+sweepall+trav3 RT_SAH_BOTTOMUP=1 RT_SAH_SWEEP=100000000 RT_SAH_TRAV=3
run sweepall RT_SAH_SWEEP=100000000
run base RT_X=0
