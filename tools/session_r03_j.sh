#!/bin/bash
# round-3 GPU session J: Float64 SB2006 column kernel compiled for 3 waves per SIMD (141 VGPRs, no spills) vs 2 (182 VGPRs) — same-box A/B
set -u
mkdir -p gpurun_out/r03j
L=cloudmicrophysics.jl_amd/csrc
REPS=3 STEPS=20 timeout 1500 tools/ab_bench.sh "sb2006_column:f64" $L/libcmx.so $L/libcmx_colw3.so 2>&1 | tee gpurun_out/r03j/ab.log
