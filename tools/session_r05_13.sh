#!/bin/bash
# Round 5, GPU session 13: the north-star sweep at the shard sizes of a strong-scaling run over 1e8 points (what ONE rank of 2, 4, 8 sweeps), on one GPU:
# the per-rank kernel and step times that DESIGN §8's strong-scaling expectation is built on (no 8-GPU node is available to this builder).
set -u
for pts in 100000000 50000000 25000000 12500000; do
  python bench.py --points $pts --steps 200 --warmup 20 --no-cpu-baseline --no-cold-probes --no-telemetry --rotate 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('points %d: step %.4f ms  kernel %.4f ms  value %.4e points/s  frac %.3f' % ($pts, d['ms_per_step'], d['roofline']['kernel_ms'], d['value'], d['roofline']['frac']))"
done | tee gpurun_out/shard_sizes_r05.txt
echo finished
