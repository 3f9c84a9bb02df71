#!/bin/bash
# Round 5, GPU session 11: incomplete-gamma loops of the Float64 shape solve — cf0 = forward continued fraction with closed-form coefficients (session 10's build), cf1 = forward with
# the coefficients by differences + series denominators by decrement, libcmx = continued fraction from the tail inwards (one recurrence instead of two) + the same series.
set -u
L=cloudmicrophysics.jl_amd/csrc
timeout 1700 python -m pytest tests/test_p3_gpu.py tests/test_p3_collisions_gpu.py tests/test_mp2m_p3_gpu.py tests/test_distribution_tools.py tests/test_row_g.py -q -m gpu --tb=short 2>&1 | grep -E "Assert|assert|Error|passed|failed|FAILED" | head -20
EXTRA="--no-cold-probes --rotate 1 --no-telemetry" REPS=2 STEPS=20 tools/ab_bench.sh "p3:f64 p3_split:f64 mp2m_p3:f64" $L/libcmx_cf0.so $L/libcmx_cf1.so $L/libcmx.so 2>&1 | tee gpurun_out/ab_r05_11.txt
echo finished
