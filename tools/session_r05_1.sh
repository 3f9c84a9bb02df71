#!/bin/bash
# Round 5, GPU session 1: full -m gpu suite on the ADVICE-r04 build; VALU probe with the packed-Float32 / scalar-operand modes; same-box A/B of the
# Float64 experiments (t* by reciprocal = default build vs base; normalised exp2 polynomial; the same with four waves requested; three-address first Horner step).
set -u
L=cloudmicrophysics.jl_amd/csrc
mkdir -p gpurun_out
timeout 1700 python -m pytest tests -q -m gpu -x 2>&1 | tail -5 | tee gpurun_out/r05_s1_tests.txt
timeout 300 tools/valu_probe > gpurun_out/r05_probe_valu.txt 2>&1
tail -14 gpurun_out/r05_probe_valu.txt | head -12
EXTRA="--no-cold-probes --rotate 1 --no-telemetry" REPS=3 STEPS=100 tools/ab_bench.sh "sb2006:f64 mp1m:f64 arg2000:f64" $L/libcmx_base.so $L/libcmx.so $L/libcmx_norm.so $L/libcmx_normw.so $L/libcmx_asm.so 2>&1 | tee gpurun_out/ab_r05_1.txt
echo finished
