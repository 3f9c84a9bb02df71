#!/bin/bash
# Round 5, GPU session 15: the partially rimed regime's area^(−½) as the reciprocal root of its collision radius instead of a logarithm; the area-law / aspect constants and D_th
# pinned in registers for the fused sweeps (libcmx) against the evidence build (ev, digest f07ffba876d5d0f7).  Parity suites of the P3 families first.
set -u
L=cloudmicrophysics.jl_amd/csrc
timeout 1700 python -m pytest tests/test_p3_gpu.py tests/test_p3_collisions_gpu.py tests/test_mp2m_p3_gpu.py tests/test_nan_inputs_gpu.py tests/test_utilities_gpu.py -q -m gpu --tb=short 2>&1 | grep -E "Assert|assert|Error|passed|failed|FAILED" | head -20
EXTRA="--no-cold-probes --rotate 1 --no-telemetry" REPS=2 STEPS=20 tools/ab_bench.sh "mp2m_p3:f64 mp2m_p3:f32 p3:f64 p3:f32 p3_selfcol:f64 p3_selfcol:f32" $L/libcmx_ev.so $L/libcmx.so 2>&1 | tee gpurun_out/ab_r05_15.txt
echo finished
