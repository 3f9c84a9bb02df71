#!/bin/bash
# Round 5, GPU session 16 (prototype, not in the sources of this round): Float32 ice evaluator of the collision kernel in base-2 units (no multiply in front of v_exp_f32 / behind
# v_log_f32) — l2u, built from a scratch copy of csrc — against the round's library.
set -u
L=cloudmicrophysics.jl_amd/csrc
CMX_LIB=$PWD/$L/libcmx_l2u.so timeout 900 python -m pytest tests/test_mp2m_p3_gpu.py tests/test_p3_collisions_gpu.py -q -m gpu --tb=short 2>&1 | grep -E "Assert|assert|Error|passed|failed|FAILED" | head -10
EXTRA="--no-cold-probes --rotate 1 --no-telemetry" REPS=2 STEPS=20 tools/ab_bench.sh "mp2m_p3:f32 mp2m_p3:f64" $L/libcmx.so $L/libcmx_l2u.so 2>&1 | tee gpurun_out/ab_r05_16.txt
echo finished
