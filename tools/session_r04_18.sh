#!/bin/bash
# Round 4, GPU session 18: the adopted workgroup sizes of the column kernels (SB2006 column 256; ARG columns 64 Float32 / 256 Float64) — tests, then same-box
# A/B against the 128-lane build; 1-moment column kernel 128 (shipped) vs 256 / 64 lanes (-DCMX_COLUMN1M_BS).
set -u
L=cloudmicrophysics.jl_amd/csrc
timeout 1200 python -m pytest tests/test_column_gpu.py tests/test_arg2000_gpu.py tests/test_mp1m_column.py -q -m gpu 2>&1 | tail -3
EXTRA="--no-cold-probes --rotate 1 --no-telemetry" REPS=2 STEPS=100 tools/ab_bench.sh "sb2006_column:f32 sb2006_column:f64 arg2000_columns:f32 arg2000_columns:f64" $L/libcmx_prev.so $L/libcmx.so 2>&1 | tee gpurun_out/ab_r04_18.txt
EXTRA="--no-cold-probes --rotate 1 --no-telemetry" REPS=2 STEPS=60 tools/ab_bench.sh "mp1m_column:f32 mp1m_column:f64 mp1m_column_lin:f32" $L/libcmx.so $L/libcmx_c1m256.so $L/libcmx_c1m64.so 2>&1 | tee -a gpurun_out/ab_r04_18.txt
echo finished
