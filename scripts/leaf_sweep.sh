#!/bin/bash
# leaf size x SAH node cost on the headline workload (both are read at commit, no rebuild)
for L in 8 10 12 14 15; do for T in 2 3 4; do
RT_MAX_LEAF=$L RT_SAH_TRAV=$T timeout -k 10 120 python -u bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('leaf $L trav $T', 'ms %.3f' % d['ms_per_step'], flush=True)"
done; done
