#!/bin/bash
# Round 5, GPU session 14: wave-uniform fast paths of the small / large switch of the Chen-2022 ice fall speed in the fall-speed quadrature (libcmx) against the evidence build (ev)
set -u
L=cloudmicrophysics.jl_amd/csrc
timeout 1700 python -m pytest tests/test_p3_gpu.py tests/test_nan_inputs_gpu.py -q -m gpu --tb=short 2>&1 | grep -E "Assert|assert|Error|passed|failed|FAILED" | head -20
EXTRA="--no-cold-probes --rotate 1 --no-telemetry" REPS=2 STEPS=20 tools/ab_bench.sh "p3:f64 p3:f32 p3_split:f64" $L/libcmx_ev.so $L/libcmx.so 2>&1 | tee gpurun_out/ab_r05_14.txt
echo finished
