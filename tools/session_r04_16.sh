#!/bin/bash
# Round 4, GPU session 16: 64-lane workgroups for the layout (fields / AoS) kernels (-DCMX_LAYOUT_BS=64) vs the shipped 128 — same-box A/B.
set -u
L=cloudmicrophysics.jl_amd/csrc
EXTRA="--no-cold-probes --rotate 1 --no-telemetry" REPS=3 STEPS=100 tools/ab_bench.sh "sb2006_fields:f32 sb2006_aos:f32 sb2006_fields:f64 mp1m:f32" $L/libcmx.so $L/libcmx_lay64.so 2>&1 | tee gpurun_out/ab_r04_16.txt
echo finished
