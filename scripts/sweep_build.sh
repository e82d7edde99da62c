#!/bin/bash
# sweep of the SAH builder's knobs on the headline workloads (wavefront pipeline)
for W in ${WORKLOADS:-pawn1080 synthetic4k}; do
for L in ${LEAVES:-4 6 8 12 15}; do
for T in ${TRAVS:-1 2 3 5}; do
  RT_MAX_LEAF=$L RT_SAH_TRAV=$T timeout -k 10 120 python -u bench.py --workload $W --pipeline wavefront --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$W leaf $L trav $T', 'ms %.3f' % d['ms_per_step'], flush=True)"
done
done
done
