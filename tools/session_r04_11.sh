#!/bin/bash
# Round 4, GPU session 11: bare-copy ceilings of the other HBM-bound kernels' column counts (tools/stream_probe shapes) next to their bench lines.
set -u
mkdir -p gpurun_out/profiles
timeout 300 tools/stream_probe 100000000 20 slab pattern shapes 2>&1 | tee gpurun_out/profiles/r04_probe_stream_shapes.txt
for wd in icenuc:f32 mp0m:f32 arg2000:f32 mp1m:f32 sb2006_fields:f32; do
  wl=${wd%%:*}; dt=${wd##*:}
  python bench.py --workload $wl --dtype $dt --steps 100 --warmup 10 --no-cpu-baseline --no-cold-probes 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl $dt kernel_ms %.4f frac %.3f rotating %.4f' % (d['roofline']['kernel_ms'], d['roofline']['frac'], d['ranks_kernel_ms']['rotating'][0]))"
done 2>&1 | tee -a gpurun_out/profiles/r04_probe_stream_shapes.txt
echo finished
