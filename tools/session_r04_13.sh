#!/bin/bash
# Round 4, GPU session 13/14: lanes per workgroup of the 1-moment kernels — same-box A/B.
#   13: tendencies kernel 256 (shipped until then) vs 128 lanes (-DCMX_1M_BLOCK=128): f32 0.804 -> 0.758 ms, f64 2.375 -> 2.396 => 128 for Float32 only
#   14: the per-type choice (libcmx) against the previous build; LinearizedAverage kernel 256 vs 128 lanes (-DCMX_1M_LIN_BLOCK=128)
set -u
L=cloudmicrophysics.jl_amd/csrc
timeout 900 python -m pytest tests/test_mp1m_gpu.py tests/test_mp1m_linearized.py -q -m gpu 2>&1 | tail -3
EXTRA="--no-cold-probes --rotate 1 --no-telemetry" REPS=3 STEPS=100 tools/ab_bench.sh "mp1m:f32 mp1m:f64" $L/libcmx_prev.so $L/libcmx.so 2>&1 | tee gpurun_out/ab_r04_14.txt
EXTRA="--no-cold-probes --rotate 1 --no-telemetry" REPS=3 STEPS=60 tools/ab_bench.sh "mp1m_lin:f32 mp1m_lin:f64" $L/libcmx.so $L/libcmx_lin128.so 2>&1 | tee -a gpurun_out/ab_r04_14.txt
echo finished
