#!/bin/bash
# A/B of the shadow pass (RT_SHADOW_PASS=1/0) on the bench workloads: ms per frame with four frames
# in flight, Mrays/s, one frame at a time, sustained.  Usage (on the GPU box): scripts/ab_shadow.sh [workloads...]
WLS=${@:-pawn1080 synthetic4k teaset1080}
for sp in 1 0; do for wl in $WLS; do
  echo "== RT_SHADOW_PASS=$sp $wl"
  RT_SHADOW_PASS=$sp timeout -k 10 150 python bench.py --workload $wl --no-cpu-baseline --sustain-s 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('x4 %.3f ms  %.0f Mrays/s  x1 %.3f ms  sustained %.3f ms  crc %d' % (d['ms_per_step'], d['value'], d['single_frame']['ms'], d['sustained']['ms_per_step'], d['config']['frame_crc32']))" || exit 1
done; done
