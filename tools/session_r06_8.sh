#!/bin/bash
# Round 6, GPU session 8: XCD-aware tile order of the column kernels against tile = workgroup index.  NOT KEPT (HISTORY.md R6): the variant was this patch of
# csrc/cmx_sb2006_column.hip and csrc/cmx_mp1m_column.hip, with xcd_tile() moved from cmx_p3_collisions.hip to cmx_launch.hpp:
#   tile0 = xcd_tile(blockIdx.x, gridDim.x) * (BS - 1);  if (tile0 >= nvec) return;      (instead of blockIdx.x * (BS - 1))
#   grid rounded up to a multiple of 8 at the two launch sites
# built as libcmx.so, and the unpatched tree as libcmx_colxcd0.so.
set -u
L=cloudmicrophysics.jl_amd/csrc
timeout 1500 python -m pytest tests/test_column_gpu.py tests/test_mp1m_column.py -q -m gpu --tb=short 2>&1 | grep -E "Error|error|assert|passed|failed|FAILED|^E " | head -20
EXTRA="--no-cold-probes --rotate 1 --no-telemetry" REPS=3 STEPS=40 tools/ab_bench.sh "sb2006_column:f32 sb2006_column:f64 mp1m_column:f32 mp1m_column_lin:f32 mp1m_column:f64" $L/libcmx_colxcd0.so $L/libcmx.so 2>&1 | tee gpurun_out/ab_r06_8.txt
echo finished
