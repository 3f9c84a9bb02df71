#!/bin/bash
# Round 4, GPU session 4: full suite on the build with overlapping column tiles + shorter Float64 polynomials + fixed ARG-columns constant;
# then a same-box A/B of that build against the previous commit's (libcmx_prev.so).
set -u
mkdir -p gpurun_out/bench gpurun_out/profiles
timeout 2700 python -m pytest tests -q -m gpu > gpurun_out/gpu_tests.log 2>&1
echo "gpu tests rc=$?"; tail -6 gpurun_out/gpu_tests.log
L=cloudmicrophysics.jl_amd/csrc
EXTRA="--no-cold-probes --rotate 1" REPS=2 STEPS=30 tools/ab_bench.sh "sb2006:f64 mp1m:f64 arg2000:f64 sb2006_column:f64 sb2006_column:f32 mp1m_column:f32 mp1m_column:f64 mp1m_lin:f64 sb2006_fields:f64 icenuc:f64 arg2000_columns:f64 arg2000_columns:f32" $L/libcmx_prev.so $L/libcmx.so 2>&1 | tee gpurun_out/ab_r04_4.txt
EXTRA="--no-cold-probes --rotate 1 --points 10000000" REPS=1 STEPS=5 tools/ab_bench.sh "p3:f64 p3:f32" $L/libcmx_prev.so $L/libcmx.so 2>&1 | tee -a gpurun_out/ab_r04_4.txt
EXTRA="--no-cold-probes --rotate 1 --points 1000000" REPS=1 STEPS=3 tools/ab_bench.sh "mp2m_p3:f64 p3_selfcol:f64" $L/libcmx_prev.so $L/libcmx.so 2>&1 | tee -a gpurun_out/ab_r04_4.txt
cp gpurun_out/parity_report.json gpurun_out/profiles/r04c_parity_report.json 2>/dev/null
