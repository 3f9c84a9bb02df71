#!/bin/bash
# Round 6, GPU session 1: ARG-2000 with the relative-accuracy Float32 erfc, the log2-domain S_max sum (three transcendentals fewer per state) and packed f32x2 states.
# Builds (all from this commit's sources unless stated):
#   libcmx.so            make -C cloudmicrophysics.jl_amd/csrc
#   libcmx_r05.so        tools/build_ref_variant.sh r05 c85d362          (round 5's final tree)
#   libcmx_argpk0.so     tools/build_variant.sh argpk0 -DCMX_ARG_F32_PACKED=0
#   libcmx_argnoslp.so   NOSLP_EXTRA=cmx_arg_kernels.o tools/build_variant.sh argnoslp
set -u
L=cloudmicrophysics.jl_amd/csrc
timeout 1500 python -m pytest tests/test_arg2000_gpu.py -q -m gpu --tb=short -x 2>&1 | tail -30
EXTRA="--no-cold-probes --rotate 1 --no-telemetry" REPS=3 STEPS=20 tools/ab_bench.sh "arg2000:f32 arg2000:f64 arg2000_columns:f32 arg2000_columns:f64" $L/libcmx_r05.so $L/libcmx.so $L/libcmx_argpk0.so $L/libcmx_argnoslp.so 2>&1 | tee gpurun_out/ab_r06_1.txt
cp gpurun_out/parity_report.json gpurun_out/parity_report_r06_1.json 2>/dev/null
echo finished
