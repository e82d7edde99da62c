#!/bin/bash
# Device builder knobs on one workload: leaf size x node cost of the leaf decision
for L in ${LEAVES:-8 10 12 15}; do for T in ${TRAVS:-1.5 2 3}; do
  RT_MAX_LEAF=$L RT_LBVH_TRAV=$T timeout -k 10 200 python -u bench.py --workload ${W:-pawn1080} --steps 20 --warmup 3 --sustain-s 0 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['counters_per_frame']; print('${W:-pawn1080} leaf=$L trav=$T', 'ms/frame %.3f' % d['ms_per_step'], 'nodes/frame %.1fM' % (c['nodes_visited']/1e6), 'tris/frame %.1fM' % (c['tri_tests']/1e6), 'bvh nodes', d['config']['bvh_nodes'], flush=True)" || exit 1
done; done
