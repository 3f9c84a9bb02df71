#!/bin/bash
# Round 4, GPU session 20: XCD-aware tile order of the collision kernel — tests, same-box A/B, FETCH / WRITE traffic of the Float32 2M + P3 step; bench.py's time-based settle phase.
set -u
mkdir -p gpurun_out/profiles
L=cloudmicrophysics.jl_amd/csrc
timeout 1500 python -m pytest tests/test_mp2m_p3_gpu.py tests/test_p3_collisions_gpu.py tests/test_layouts_gpu.py tests/test_nan_inputs_gpu.py -q -m gpu 2>&1 | tail -3
EXTRA="--no-cold-probes --rotate 1 --no-telemetry --points 1000000" REPS=2 STEPS=3 tools/ab_bench.sh "mp2m_p3:f32 mp2m_p3:f64" $L/libcmx_prev.so $L/libcmx.so 2>&1 | tee gpurun_out/ab_r04_20.txt
KT_STEPS=10 tools/profile.sh mp2m_p3 f32 1000000 r04x > gpurun_out/prof_x.log 2>&1; python -c "
import json; d=json.load(open('gpurun_out/profiles/r04x_pmc_traffic_mp2m_p3_f32.json')); print('mp2m_p3 f32 traffic/algorithmic', d['traffic_over_algorithmic'], 'fetch', d['fetch_bytes_corrected'], 'write', d['write_bytes'])"
for i in 1 2 3; do python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-cold-probes --no-telemetry 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('default-like steps 20: kernel_ms %.4f same %.4f rot %.4f settle_steps %d' % (d['roofline']['kernel_ms'], d['same_buffer_ms_per_step'], d['rotating_ms_per_step'], d['settle_steps']))"; done
echo finished
