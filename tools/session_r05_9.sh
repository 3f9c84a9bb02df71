#!/bin/bash
# Round 5, GPU session 9: P3 integrands — (a) one D^(σ/2) exponential for both non-spherical regimes, (b) exp / log coefficients pinned in registers for the
# self-collection / melting sweeps only (loc1: the second polynomial coefficient; loc2: all eighteen constants), (c) floating-point contraction in the P3
# translation units (loc2fast) and in every translation unit (allfast: measurement of what the other Float64 kernels would gain).  base = the round's
# evidence build (digest 7e4565c4d4720fce).  Parity suites of the P3 families on the contracted build first, then same-box A/B.
set -u
L=cloudmicrophysics.jl_amd/csrc
CMX_LIB=$PWD/$L/libcmx_loc2fast.so timeout 1700 python -m pytest tests/test_p3_gpu.py tests/test_p3_collisions_gpu.py tests/test_mp2m_p3_gpu.py tests/test_nan_inputs_gpu.py tests/test_row_g.py -q -m gpu --tb=short 2>&1 | grep -E "Assert|assert|Error|passed|failed|FAILED" | head -20
EXTRA="--no-cold-probes --rotate 1 --no-telemetry" REPS=2 STEPS=30 tools/ab_bench.sh "mp2m_p3:f64 mp2m_p3:f32 p3:f64 p3:f32 p3_selfcol:f64 p3_selfcol:f32" $L/libcmx_base.so $L/libcmx_loc1.so $L/libcmx_loc2.so $L/libcmx_loc2fast.so 2>&1 | tee gpurun_out/ab_r05_9.txt
EXTRA="--no-cold-probes --rotate 1 --no-telemetry" REPS=2 STEPS=100 tools/ab_bench.sh "sb2006:f64 mp1m:f64 arg2000:f64 mp1m_lin:f64 sb2006_column:f64 cloud_diag:f64 sb2006:f32 mp1m_lin:f32" $L/libcmx_base.so $L/libcmx_allfast.so 2>&1 | tee -a gpurun_out/ab_r05_9.txt
echo finished
