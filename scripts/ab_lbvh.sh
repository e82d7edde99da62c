#!/bin/bash
# Device-built tree with / without the rebuilt large subtrees (RT_LBVH_BIG_TREELETS) against the host SAH
# build: ms per frame (four in flight), commit time.  Usage (GPU box): scripts/ab_lbvh.sh [workloads...]
WLS=${@:-pawn1080 teaset1080 synthetic4k}
for wl in $WLS; do for cfg in "sah 1" "lbvh 1" "lbvh 0"; do
  set -- $cfg
  RT_LBVH_BIG_TREELETS=$2 timeout -k 10 150 python bench.py --workload $wl --bvh $1 --no-cpu-baseline --sustain-s 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['counters_per_frame']; print('$wl $1 big=$2: x4 %.3f ms  sustained %.3f  x1 %.3f ms  commit %.1f ms  nodes %d  node visits %.1fM tri tests %.1fM crc %d' % (d['ms_per_step'], d['sustained']['ms_per_step'], d['single_frame']['ms'], d['config']['commit_ms'], d['config']['bvh_nodes'], c['nodes_visited']/1e6, c['tri_tests']/1e6, d['config']['frame_crc32']), flush=True)" || exit 1
done; done
