#!/bin/bash
# Round 5, GPU session 8: column kernels with ONE-WAVE tiles (neighbour flux by wave shuffle: no LDS halo, no workgroup barrier) — parity suites, then same-box A/B:
# wg = workgroup tiles (rounds 2-4), libcmx = wave tiles in 256-lane workgroups, wt128 / wt64 = wave tiles in 128- / 64-lane workgroups.
set -u
L=cloudmicrophysics.jl_amd/csrc
timeout 1500 python -m pytest tests/test_column_gpu.py tests/test_mp1m_column.py tests/test_nan_inputs_gpu.py -q -m gpu -x --tb=short 2>&1 | grep -E "Assert|assert|Error|passed|failed" | head -8
EXTRA="--no-cold-probes --rotate 1 --no-telemetry" REPS=3 STEPS=200 tools/ab_bench.sh "sb2006_column:f32 sb2006_column:f64 mp1m_column:f32 mp1m_column_lin:f32 mp1m_column:f64" $L/libcmx_wg.so $L/libcmx.so $L/libcmx_wt128.so $L/libcmx_wt64.so 2>&1 | tee gpurun_out/ab_r05_8.txt
echo finished
