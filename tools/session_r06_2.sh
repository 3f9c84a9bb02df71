#!/bin/bash
# Round 6, GPU session 2: ARG-2000 — sink terms in front of the mode loop, both sums pinned per mode in the per-element kernel (157 → 96 VGPRs at 8 Float64 modes),
# workgroup-base addressing there, phase-local constants for the Float32 number+mass instantiation.  A/B against round 5's final tree:
#   libcmx.so        make -C cloudmicrophysics.jl_amd/csrc
#   libcmx_r05.so    tools/build_ref_variant.sh r05 c85d362
set -u
L=cloudmicrophysics.jl_amd/csrc
timeout 1500 python -m pytest tests/test_arg2000_gpu.py tests/test_row_g.py tests/test_nan_inputs_gpu.py -q -m gpu --tb=short 2>&1 | tail -30
EXTRA="--no-cold-probes --rotate 1 --no-telemetry" REPS=3 STEPS=20 tools/ab_bench.sh "arg2000:f32 arg2000:f64 arg2000_columns:f32 arg2000_columns:f64" $L/libcmx_r05.so $L/libcmx.so 2>&1 | tee gpurun_out/ab_r06_2.txt
cp gpurun_out/parity_report.json gpurun_out/parity_report_r06_2.json 2>/dev/null
echo finished
