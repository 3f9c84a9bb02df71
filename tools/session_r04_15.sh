#!/bin/bash
# Round 4, GPU session 15: 64-lane workgroups for the 1-moment, ARG and SB2006 sweeps (-DCMX_1M_BLOCK=64 -DCMX_ARG_BS=64 -DCMX_TEND_BS=64) vs shipped — same-box A/B.
set -u
L=cloudmicrophysics.jl_amd/csrc
EXTRA="--no-cold-probes --rotate 1 --no-telemetry" REPS=3 STEPS=100 tools/ab_bench.sh "mp1m:f32 arg2000:f32 sb2006:f32 sb2006_chen:f32 arg2000:f64 sb2006:f64" $L/libcmx.so $L/libcmx_bs64.so 2>&1 | tee gpurun_out/ab_r04_15.txt
echo finished
