#!/bin/bash
# Round 6, GPU session 14 (NOT KEPT, HISTORY.md R6; the variant is the patch described here): the rain curve's fall speed at the two ends of the crossover bracket is evaluated once per state (two more passes of the rain-node cache
# loop, S[28], S[29]) instead of once per outer node inside the solve — same expression on the same values, same iterates.  libcmx_head.so = HEAD, libcmx.so = tree.
set -u
L=cloudmicrophysics.jl_amd/csrc
timeout 1500 python -m pytest tests/test_p3_collisions_gpu.py tests/test_mp2m_p3_gpu.py tests/test_reference_suites_gpu.py -q -m gpu --tb=short 2>&1 | grep -E "Error|error|assert|passed|failed|FAILED|^E " | head -20
EXTRA="--no-cold-probes --rotate 1 --no-telemetry" REPS=3 STEPS=20 tools/ab_bench.sh "mp2m_p3:f64 mp2m_p3:f32" $L/libcmx_head.so $L/libcmx.so 2>&1 | tee gpurun_out/ab_r06_14.txt
echo finished
