#!/bin/bash
# Round 4, GPU session 17: lanes per workgroup of the SB2006 column and ARG columns kernels (-DCMX_COLUMN_BS / -DCMX_ARGCOL_BS = 64, 256) vs the shipped 128.
set -u
L=cloudmicrophysics.jl_amd/csrc
EXTRA="--no-cold-probes --rotate 1 --no-telemetry" REPS=2 STEPS=100 tools/ab_bench.sh "sb2006_column:f32 sb2006_column:f64 arg2000_columns:f32 arg2000_columns:f64" $L/libcmx.so $L/libcmx_col64.so $L/libcmx_col256.so 2>&1 | tee gpurun_out/ab_r04_17.txt
echo finished
