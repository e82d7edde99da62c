#!/bin/bash
# A/B of library builds over workloads: WORKLOADS="..." bash scripts/ab_configs.sh NAME... ("base" = in-tree)
for W in ${WORKLOADS:-pawn1080 teaset1080 synthetic4k cover400}; do
for V in "$@"; do
  if [ "$V" = base ]; then L=""; else L=$(pwd)/variants/$V.so; fi
  RT_LIB=$L timeout -k 10 200 python -u bench.py --workload $W --steps ${STEPS:-20} --warmup 3 --sustain-s 0 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$W $V', 'ms/frame %.3f' % d['ms_per_step'], 'single %.3f' % d.get('single_frame',{}).get('ms', d['config']['frame_latency_ms']), 'crc', d['config'].get('frame_crc32'), flush=True)" || exit 1
done
done
