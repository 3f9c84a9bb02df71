#!/bin/bash
# Round 5, GPU session 10: the P3 changes as built by default — contraction by pragma around the P3 code only (the pointwise part of the 2M + P3 entry stays
# bit-identical to the warm-rain entry), shared D^(σ/2) exponential and phase-local exp / log constants in the self-collection / melting sweeps.
# Parity suites of every family that includes cmx_p3.hpp, then same-box A/B against the evidence build (base, digest 7e4565c4d4720fce).
set -u
L=cloudmicrophysics.jl_amd/csrc
timeout 1700 python -m pytest tests/test_p3_gpu.py tests/test_p3_collisions_gpu.py tests/test_mp2m_p3_gpu.py tests/test_nan_inputs_gpu.py tests/test_row_g.py tests/test_distribution_tools.py tests/test_abi.py tests/test_layouts_gpu.py -q -m gpu --tb=short 2>&1 | grep -E "Assert|assert|Error|passed|failed|FAILED" | head -20
EXTRA="--no-cold-probes --rotate 1 --no-telemetry" REPS=1 STEPS=20 tools/ab_bench.sh "mp2m_p3:f64 mp2m_p3:f32 p3:f64 p3:f32 p3_split:f64 p3_selfcol:f64 p3_selfcol:f32" $L/libcmx_base.so $L/libcmx.so 2>&1 | tee gpurun_out/ab_r05_10.txt
echo finished
