#!/bin/bash
# round-3 GPU session E: full -m gpu suite with the Float64 finite-argument forms, then same-box A/B of
#   libcmx.so (finite forms, 3 waves) | libcmx_nofin.so (full forms) | libcmx_n1.so (one Newton step) | libcmx_w2.so (collision kernel at 2 waves)
set -u
mkdir -p gpurun_out/r03e
timeout 3000 python -m pytest tests -q -m gpu > gpurun_out/r03e/tests.log 2>&1
echo "tests rc=$?"; tail -8 gpurun_out/r03e/tests.log
L=cloudmicrophysics.jl_amd/csrc
REPS=2 STEPS=20 timeout 1500 tools/ab_bench.sh "sb2006:f64 mp1m:f64 arg2000:f64 mp1m_lin:f64 sb2006_column:f64 icenuc:f64" $L/libcmx.so $L/libcmx_nofin.so $L/libcmx_n1.so 2>&1 | tee gpurun_out/r03e/ab_f64.log
REPS=2 STEPS=5 EXTRA="--points 1000000" timeout 1500 tools/ab_bench.sh "mp2m_p3:f64 p3_selfcol:f64" $L/libcmx.so $L/libcmx_nofin.so $L/libcmx_w2.so 2>&1 | tee gpurun_out/r03e/ab_p3.log
REPS=1 STEPS=5 EXTRA="--points 10000000" timeout 900 tools/ab_bench.sh "p3:f64 p3_split:f64 p3:f32 p3_split:f32" $L/libcmx.so $L/libcmx_nofin.so 2>&1 | tee gpurun_out/r03e/ab_p3shape.log
