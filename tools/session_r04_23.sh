#!/bin/bash
# Round 4, GPU session 23: SB2006 and ARG sweeps at 256 lanes per workgroup (-DCMX_TEND_BS=256 -DCMX_ARG_BS=256) vs the shipped 128 — same-box A/B, both float types.
set -u
L=cloudmicrophysics.jl_amd/csrc
EXTRA="--no-cold-probes --rotate 1 --no-telemetry" REPS=3 STEPS=100 tools/ab_bench.sh "sb2006:f64 arg2000:f64 sb2006:f32 arg2000:f32 sb2006_chen:f32" $L/libcmx.so $L/libcmx_tend256.so 2>&1 | tee gpurun_out/ab_r04_23.txt
echo finished
