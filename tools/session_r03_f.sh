#!/bin/bash
# round-3 GPU session F: full -m gpu suite (collision kernel without scratch, ARG / erfc changes, one-step rcp_nz), then same-box A/B of
#   libcmx.so (this tree) | libcmx_prev.so (previous commit) | libcmx_nofin.so (this tree, full Float64 forms)
set -u
mkdir -p gpurun_out/r03f
timeout 3000 python -m pytest tests -q -m gpu > gpurun_out/r03f/tests.log 2>&1
echo "tests rc=$?"; tail -8 gpurun_out/r03f/tests.log
L=cloudmicrophysics.jl_amd/csrc
REPS=2 STEPS=20 timeout 1500 tools/ab_bench.sh "sb2006:f64 mp1m:f64 arg2000:f64 mp1m_lin:f64 arg2000:f32" $L/libcmx.so $L/libcmx_prev.so $L/libcmx_nofin.so 2>&1 | tee gpurun_out/r03f/ab_f64.log
REPS=2 STEPS=5 EXTRA="--points 1000000" timeout 1500 tools/ab_bench.sh "mp2m_p3:f64 mp2m_p3:f32" $L/libcmx.so $L/libcmx_prev.so 2>&1 | tee gpurun_out/r03f/ab_p3.log
