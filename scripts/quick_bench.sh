#!/bin/bash
# quick A/B: both pipelines on the headline workload (+ optional others)
for W in ${WORKLOADS:-pawn1080}; do
for P in ${PIPES:-megakernel wavefront}; do
  timeout -k 10 120 python -u bench.py --workload $W --pipeline $P --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$W $P', 'Mrays/s %.1f' % d['value'], 'ms %.3f' % d['ms_per_step'])"
done
done
