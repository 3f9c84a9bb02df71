#!/bin/bash
# Round 6, GPU session 7: how much of the 2M + P3 step is the crossover solve?  The reference's budget (libcmx) against 4 and 1 zeroin iterations.
#   libcmx_cross4.so   tools/build_variant.sh cross4 -DCMX_P3_CROSSOVER_ITERS=4
#   libcmx_cross1.so   tools/build_variant.sh cross1 -DCMX_P3_CROSSOVER_ITERS=1
set -u
L=cloudmicrophysics.jl_amd/csrc
EXTRA="--no-cold-probes --rotate 1 --no-telemetry" REPS=2 STEPS=10 tools/ab_bench.sh "mp2m_p3:f64 mp2m_p3:f32" $L/libcmx.so $L/libcmx_cross4.so $L/libcmx_cross1.so 2>&1 | tee gpurun_out/ab_r06_7.txt
echo finished
