#!/bin/bash
# hardware queues x frames in flight x frame split on the headline workload
for Q in 4 8; do
for F in 2 3 4 6; do
for S in 1 2; do
  GPU_MAX_HW_QUEUES=$Q RT_FRAME_SPLIT=$S timeout -k 10 120 python -u bench.py --workload ${W:-pawn1080} --hw-queues $Q --inflight $F --steps 24 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('queues=$Q inflight=$F split=$S', 'ms/frame %.3f' % d['ms_per_step'], 'latency %.3f' % d['config']['frame_latency_ms'], flush=True)"
done; done; done
