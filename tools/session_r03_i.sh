#!/bin/bash
# round-3 GPU session I: second polynomial coefficients shared in VGPRs (CMX_LEAN_VGPR_C1) — same-box A/B
set -u
mkdir -p gpurun_out/r03i
L=cloudmicrophysics.jl_amd/csrc
REPS=3 STEPS=20 timeout 1500 tools/ab_bench.sh "sb2006:f64 mp1m:f64 arg2000:f64 mp1m_lin:f64" $L/libcmx.so $L/libcmx_novc.so 2>&1 | tee gpurun_out/r03i/ab.log
