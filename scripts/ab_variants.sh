#!/bin/bash
# A/B of library builds: bash scripts/ab_variants.sh NAME...   ("base" = the in-tree library)
for W in ${WORKLOADS:-pawn1080 synthetic4k}; do
for V in "$@"; do
  if [ "$V" = base ]; then L=""; else L=$(pwd)/variants/$V.so; fi
  RT_LIB=$L timeout -k 10 120 python -u bench.py --workload $W --pipeline ${PIPE:-wavefront} --steps ${STEPS:-20} --warmup 3 --inflight ${INFLIGHT:-4} --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$W $V', 'ms %.3f' % d['ms_per_step'], 'crc', d['config'].get('frame_crc32'), flush=True)"
done
done
