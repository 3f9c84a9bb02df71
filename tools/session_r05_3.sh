#!/bin/bash
# Round 5, GPU session 3: packed Float32 arithmetic in every SB2006 kernel (tendencies, column step, layout adapters) — parity suites, then same-box A/B of
# scalar (rounds 1-4) / packed with phase-local constants (default) / the same with four waves requested for the tendencies kernel / packed with the constants in SGPRs.
set -u
L=cloudmicrophysics.jl_amd/csrc
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_sb2006_gpu.py tests/test_column_gpu.py tests/test_layouts_gpu.py tests/test_nan_inputs_gpu.py tests/test_graphs_gpu.py -q -m gpu -x 2>&1 | tail -4 | tee gpurun_out/r05_s3_tests.txt
EXTRA="--no-cold-probes --rotate 1 --no-telemetry" REPS=4 STEPS=200 tools/ab_bench.sh "sb2006:f32 sb2006_chen:f32 sb2006_column:f32 sb2006_fields:f32 sb2006_aos:f32" $L/libcmx_scalar.so $L/libcmx.so $L/libcmx_pkw4.so $L/libcmx_nophase.so 2>&1 | tee gpurun_out/ab_r05_3.txt
echo finished
