#!/bin/bash
# Round 5, GPU session 4: packed Float32 arithmetic in the 1-moment kernels (tendencies, LinearizedAverage pairs, column step, layout adapters) and the final
# SB2006 choices (north star one point at a time, Chen / column / layouts packed) — parity suites of both families, then same-box A/B scalar vs packed.
set -u
L=cloudmicrophysics.jl_amd/csrc
mkdir -p gpurun_out
timeout 1700 python -m pytest tests/test_sb2006_gpu.py tests/test_column_gpu.py tests/test_layouts_gpu.py tests/test_nan_inputs_gpu.py tests/test_mp1m_gpu.py tests/test_mp1m_linearized.py tests/test_mp1m_column.py tests/test_abi_caller.py -q -m gpu -x 2>&1 | tail -4 | tee gpurun_out/r05_s4_tests.txt
EXTRA="--no-cold-probes --rotate 1 --no-telemetry" REPS=3 STEPS=200 tools/ab_bench.sh "mp1m:f32 mp1m_lin:f32 mp1m_column:f32 mp1m_column_lin:f32 sb2006:f32 sb2006_chen:f32" $L/libcmx_scalar.so $L/libcmx.so 2>&1 | tee gpurun_out/ab_r05_4.txt
echo finished
