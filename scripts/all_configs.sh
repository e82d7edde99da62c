#!/bin/bash
# every benchmark configuration, both pipelines, one frame at a time and three frames in flight
for W in cover400 pawn1080 teaset1080 pawn8k synthetic4k; do
for P in wavefront megakernel; do
for F in 1 4; do
  timeout -k 10 300 python -u bench.py --workload $W --pipeline $P --inflight $F --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$W $P inflight=$F', 'ms/frame %.3f' % d['ms_per_step'], 'Mrays/s %.0f' % d['value'], 'latency %.3f' % d['config']['frame_latency_ms'], 'rays %d' % d['config']['rays_per_frame'])"
done; done; done
for W in pawn1080 teaset1080 synthetic4k; do
  timeout -k 10 300 python -u bench.py --workload $W --mode f32_fast --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$W f32_fast inflight=4', 'ms/frame %.3f' % d['ms_per_step'], 'Mrays/s %.0f' % d['value'], d['pixel_match_vs_f64_parity'])"
  timeout -k 10 300 python -u bench.py --workload $W --bvh lbvh --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$W lbvh inflight=4', 'ms/frame %.3f' % d['ms_per_step'], 'Mrays/s %.0f' % d['value'], 'commit_ms %.0f' % d['config']['commit_ms'])"
done
