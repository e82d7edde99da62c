#!/bin/bash
# sort-key variants (variants/*.so) x first sorted level, one frame at a time and four in flight.
# Stops at the first run that fails or renders another frame than the unsorted one.
WANT=${CRC:-1906925002}
for V in "$@"; do
  if [ "$V" = base ]; then L=""; else L=$(pwd)/variants/$V.so; fi
  for ML in ${LEVELS:-1 2}; do
  for F in 1 4; do
    OUT=$(RT_LIB=$L RT_SORT_MIN_LEVEL=$ML timeout -k 10 120 python -u bench.py --workload ${W:-pawn1080} --inflight $F --steps 20 --warmup 3 --sustain-s 0 --no-cpu-baseline 2>/dev/null) || { echo "$V level $ML inflight $F: bench failed"; exit 1; }
    echo "$OUT" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$V from_level=$ML inflight=$F', 'ms/frame %.3f' % d['ms_per_step'], 'single %.3f' % d['single_frame']['ms'], 'crc', d['config']['frame_crc32'], flush=True); sys.exit(0 if d['config']['frame_crc32'] == $WANT else 3)" || { echo "wrong frame"; exit 1; }
  done; done
done
