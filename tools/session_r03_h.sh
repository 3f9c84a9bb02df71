#!/bin/bash
# round-3 GPU session H: kernel-argument constants pinned in the per-lane selects of the P3 quadrature loops (no vector loads from the
# kernel-argument segment inside the loops) — P3 tests, then same-box A/B against the previous commit
set -u
mkdir -p gpurun_out/r03h
timeout 3000 python -m pytest tests/test_p3_gpu.py tests/test_p3_collisions_gpu.py tests/test_mp2m_p3_gpu.py -q -m gpu > gpurun_out/r03h/tests.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/r03h/tests.log
L=cloudmicrophysics.jl_amd/csrc
REPS=2 STEPS=5 EXTRA="--points 1000000" timeout 1500 tools/ab_bench.sh "mp2m_p3:f64 mp2m_p3:f32 p3_selfcol:f64 p3_selfcol:f32" $L/libcmx.so $L/libcmx_prev.so 2>&1 | tee gpurun_out/r03h/ab_p3.log
REPS=1 STEPS=5 EXTRA="--points 10000000" timeout 900 tools/ab_bench.sh "p3:f64 p3:f32" $L/libcmx.so $L/libcmx_prev.so 2>&1 | tee gpurun_out/r03h/ab_p3shape.log
