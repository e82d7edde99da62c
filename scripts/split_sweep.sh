#!/bin/bash
# frame split x frames in flight on the headline workloads
for W in ${WORKLOADS:-pawn1080 teaset1080 synthetic4k}; do
for F in 1 3; do
for S in 1 2 3 4; do
  RT_FRAME_SPLIT=$S timeout -k 10 120 python -u bench.py --workload $W --inflight $F --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$W inflight=$F split=$S', 'ms/frame %.3f' % d['ms_per_step'], 'latency %.3f' % d['config']['frame_latency_ms'], 'crc', d['config'].get('frame_crc32'), flush=True)"
done; done; done
