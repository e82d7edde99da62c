#!/bin/bash
# SAH builder knobs on one workload: leaf size x node cost
for L in ${LEAVES:-8 12 15}; do for T in ${TRAVS:-2 3 5 8}; do
  RT_MAX_LEAF=$L RT_SAH_TRAV=$T timeout -k 10 200 python -u bench.py --workload ${W:-synthetic4k} --steps 10 --warmup 2 --sustain-s 0 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['counters_per_frame']; print('leaf=$L trav=$T', 'ms/frame %.3f' % d['ms_per_step'], 'nodes/frame %.1fM' % (c['nodes_visited']/1e6), 'tris/frame %.1fM' % (c['tri_tests']/1e6), 'bvh nodes', d['config']['bvh_nodes'], flush=True)" || exit 1
done; done
