#!/bin/bash
# Coherence sort of the ray queues on/off (RT_SORT_RAYS, RT_SORT_MIN_LEVEL): ms per frame with four
# frames in flight and one at a time.  Usage (GPU box): scripts/ab_sort.sh [workloads...]
WLS=${@:-pawn8k synthetic4k pawn1080}
for wl in $WLS; do for cfg in "0 2" "1 2" "1 1"; do
  set -- $cfg
  RT_SORT_RAYS=$1 RT_SORT_MIN_LEVEL=$2 timeout -k 10 150 python bench.py --workload $wl --no-cpu-baseline --sustain-s 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl sort=$1 from level $2: x4 %.3f ms  %.0f Mrays/s  x1 %.3f ms  crc %d' % (d['ms_per_step'], d['value'], d['single_frame']['ms'], d['config']['frame_crc32']), flush=True)" || exit 1
done; done
