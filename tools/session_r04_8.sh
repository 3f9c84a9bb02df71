#!/bin/bash
# Round 4, GPU session 8 (and 9): Float64 ice integrands (2M + P3, self-collection) — the merged non-spherical block (slower: reverted), then log_pos — parity, then same-box A/B;
# bench telemetry smoke test; distribution-tools tests after the rain-PSD fix.
set -u
mkdir -p gpurun_out/bench gpurun_out/profiles
timeout 1500 python -m pytest tests/test_distribution_tools.py tests/test_mp2m_p3_gpu.py tests/test_p3_collisions_gpu.py tests/test_p3_gpu.py -q -m gpu 2>&1 | tail -8
L=cloudmicrophysics.jl_amd/csrc
EXTRA="--no-cold-probes --rotate 1 --points 1000000 --no-telemetry" REPS=2 STEPS=3 tools/ab_bench.sh "mp2m_p3:f64 p3_selfcol:f64" $L/libcmx_prev.so $L/libcmx.so 2>&1 | tee gpurun_out/ab_r04_9.txt
EXTRA="--no-cold-probes --rotate 1 --points 10000000 --no-telemetry" REPS=2 STEPS=5 tools/ab_bench.sh "p3:f64" $L/libcmx_prev.so $L/libcmx.so 2>&1 | tee -a gpurun_out/ab_r04_9.txt
python bench.py --workload sb2006 --dtype f64 --steps 20 --warmup 3 --no-cpu-baseline --no-cold-probes 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('telemetry', d['telemetry']); print('valu', d['roofline'].get('valu'))"
echo finished
