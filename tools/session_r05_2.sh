#!/bin/bash
# Round 5, GPU session 2: packed Float32 arithmetic (f32x2 value type) in the SB2006 tendencies kernel — parity suite of the SB2006 family, then same-box A/B:
# scalar (rounds 1-4) vs packed with phase-local constants (default) vs the same with four waves requested vs packed with all constants in SGPRs.
set -u
L=cloudmicrophysics.jl_amd/csrc
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_sb2006_gpu.py tests/test_nan_inputs_gpu.py tests/test_abi_caller.py -q -m gpu -x 2>&1 | tail -4 | tee gpurun_out/r05_s2_tests.txt
EXTRA="--no-cold-probes --rotate 1 --no-telemetry" REPS=3 STEPS=100 tools/ab_bench.sh "sb2006:f32 sb2006_chen:f32" $L/libcmx_scalar.so $L/libcmx.so $L/libcmx_pkw4.so $L/libcmx_pknophase.so 2>&1 | tee gpurun_out/ab_r05_2.txt
echo finished
