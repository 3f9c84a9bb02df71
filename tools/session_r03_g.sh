#!/bin/bash
# round-3 GPU session G: the 2M + P3 entry as ONE launch (pointwise part in the collision kernel's epilogue) — full -m gpu suite, same-box A/B
# against the two-launch build (-DCMX_MP2M_P3_ONE_LAUNCH=0), and the PMC traffic of the Float64 step
set -u
mkdir -p gpurun_out/r03g
timeout 3000 python -m pytest tests -q -m gpu > gpurun_out/r03g/tests.log 2>&1
echo "tests rc=$?"; tail -6 gpurun_out/r03g/tests.log
L=cloudmicrophysics.jl_amd/csrc
REPS=2 STEPS=5 EXTRA="--points 1000000" timeout 1500 tools/ab_bench.sh "mp2m_p3:f64 mp2m_p3:f32" $L/libcmx.so $L/libcmx_twolaunch.so 2>&1 | tee gpurun_out/r03g/ab_p3.log
REPS=1 STEPS=20 EXTRA="--points 20000" timeout 600 tools/ab_bench.sh "mp2m_p3:f64 mp2m_p3:f32" $L/libcmx.so $L/libcmx_twolaunch.so 2>&1 | tee gpurun_out/r03g/ab_p3_small.log
KT_STEPS=10 tools/profile.sh mp2m_p3 f64 1000000 r03g > gpurun_out/r03g/prof.log 2>&1; tail -2 gpurun_out/r03g/prof.log | cut -c1-600
