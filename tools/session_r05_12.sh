#!/bin/bash
# Round 5, GPU session 12: ln x of the incomplete gammas of the shape solve from ln λ + ln D_threshold (lx) against session 11's build (libcmx)
set -u
L=cloudmicrophysics.jl_amd/csrc
CMX_LIB=$PWD/$L/libcmx_lx.so timeout 1700 python -m pytest tests/test_p3_gpu.py tests/test_mp2m_p3_gpu.py -q -m gpu --tb=short 2>&1 | grep -E "Assert|assert|Error|passed|failed|FAILED" | head -20
EXTRA="--no-cold-probes --rotate 1 --no-telemetry" REPS=2 STEPS=20 tools/ab_bench.sh "p3:f64 p3_split:f64 mp2m_p3:f64 p3:f32" $L/libcmx.so $L/libcmx_lx.so 2>&1 | tee gpurun_out/ab_r05_12.txt
echo finished
