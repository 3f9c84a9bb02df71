#!/bin/bash
# Round 4, GPU session 19: lanes per workgroup of the P3 kernels (one state per lane: -DCMX_P3_BS; collision kernel: -DCMX_COL_THREADS) 256 (shipped) vs 128 vs 64.
set -u
L=cloudmicrophysics.jl_amd/csrc
EXTRA="--no-cold-probes --rotate 1 --no-telemetry --points 10000000" REPS=2 STEPS=5 tools/ab_bench.sh "p3:f64 p3:f32" $L/libcmx.so $L/libcmx_p3bs128.so $L/libcmx_p3bs64.so 2>&1 | tee gpurun_out/ab_r04_19.txt
EXTRA="--no-cold-probes --rotate 1 --no-telemetry --points 1000000" REPS=2 STEPS=3 tools/ab_bench.sh "mp2m_p3:f64 mp2m_p3:f32 p3_selfcol:f64 p3_selfcol:f32" $L/libcmx.so $L/libcmx_p3bs128.so $L/libcmx_p3bs64.so 2>&1 | tee -a gpurun_out/ab_r04_19.txt
echo finished
