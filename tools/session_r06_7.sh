#!/bin/bash
# Round 6, GPU session 7: how much of the 2M + P3 step is the crossover solve?  The reference's budget (libcmx) against 4 and 1 zeroin iterations.
# The two variants are one-line patches of csrc/cmx_p3_collisions.hip (not kept in the tree: the evidence of the round is tied to the kernel sources' digest):
#   sed -i 's/k.brent_iters = sizeof(FT) == 4 ? 8 : 10;/k.brent_iters = 4;/' cloudmicrophysics.jl_amd/csrc/cmx_p3_collisions.hip && tools/build_variant.sh cross4
#   (… = 1; for cross1), then `git checkout` the file
set -u
L=cloudmicrophysics.jl_amd/csrc
EXTRA="--no-cold-probes --rotate 1 --no-telemetry" REPS=2 STEPS=10 tools/ab_bench.sh "mp2m_p3:f64 mp2m_p3:f32" $L/libcmx.so $L/libcmx_cross4.so $L/libcmx_cross1.so 2>&1 | tee gpurun_out/ab_r06_7.txt
echo finished
