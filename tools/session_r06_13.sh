#!/bin/bash
# Round 6, GPU session 13: column-step tiles of 512 and 1024 lanes (CMX_COLUMN_BS / CMX_COLUMN1M_BS; round 4 measured 64 / 128 / 256 and the trend was still
# falling).  libcmx_bs512.so / libcmx_bs1024.so: tools/build_variant.sh bs512 -DCMX_COLUMN_BS=512 -DCMX_COLUMN1M_BS=512 (and 1024).
set -u
L=cloudmicrophysics.jl_amd/csrc
CMX_LIB=$L/libcmx_bs512.so timeout 900 python -m pytest tests/test_column_gpu.py tests/test_mp1m_column.py -q -m gpu --tb=short 2>&1 | grep -E "Error|error|assert|passed|failed|FAILED|^E " | head -10
EXTRA="--no-cold-probes --rotate 1 --no-telemetry" REPS=3 STEPS=40 tools/ab_bench.sh "sb2006_column:f32 sb2006_column:f64 mp1m_column:f32 mp1m_column:f64" $L/libcmx.so $L/libcmx_bs512.so $L/libcmx_bs1024.so 2>&1 | tee gpurun_out/ab_r06_13.txt
echo finished
