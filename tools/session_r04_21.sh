#!/bin/bash
# Round 4, GPU session 21: column kernels at 512 lanes per workgroup (-DCMX_COLUMN_BS=512 -DCMX_COLUMN1M_BS=512) vs the shipped 256 — same-box A/B.
set -u
L=cloudmicrophysics.jl_amd/csrc
EXTRA="--no-cold-probes --rotate 1 --no-telemetry" REPS=2 STEPS=100 tools/ab_bench.sh "sb2006_column:f32 sb2006_column:f64 mp1m_column:f32 mp1m_column:f64" $L/libcmx.so $L/libcmx_col512.so 2>&1 | tee gpurun_out/ab_r04_21.txt
echo finished
