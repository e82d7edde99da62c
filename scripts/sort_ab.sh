#!/bin/bash
# coherence sort on/off and from which level, one frame at a time and four in flight
for CFG in "0 1" "1 1" "1 2" "1 3"; do
  set -- $CFG
  for F in 1 4; do
    RT_SORT_RAYS=$1 RT_SORT_MIN_LEVEL=$2 timeout -k 10 120 python -u bench.py --workload ${W:-pawn1080} --inflight $F --steps 20 --warmup 3 --sustain-s 0 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('sort=$1 from_level=$2 inflight=$F', 'ms/frame %.3f' % d['ms_per_step'], 'single %.3f' % d['single_frame']['ms'], 'crc', d['config']['frame_crc32'], flush=True)"
  done
done
