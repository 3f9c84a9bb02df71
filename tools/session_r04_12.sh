#!/bin/bash
# Round 4, GPU session 12: 128-lane workgroups for the ice-nucleation and 0-moment sweeps — tests, then same-box A/B against the 256-lane build.
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ice_nucleation_gpu.py tests/test_mp0m.py tests/test_row_g.py tests/test_nan_inputs_gpu.py -q -m gpu 2>&1 | tail -4
L=cloudmicrophysics.jl_amd/csrc
EXTRA="--no-cold-probes --rotate 1 --no-telemetry" REPS=3 STEPS=100 tools/ab_bench.sh "icenuc:f32 icenuc:f64 mp0m:f32 mp0m:f64" $L/libcmx_prev.so $L/libcmx.so 2>&1 | tee gpurun_out/ab_r04_12.txt
echo finished
