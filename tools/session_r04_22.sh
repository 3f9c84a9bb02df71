#!/bin/bash
# Round 4, GPU session 22: the pointwise constants of the one-launch 2M + P3 kernel read through the kernel-argument segment at the point of use (Float64) — parity, same-box A/B.
set -u
L=cloudmicrophysics.jl_amd/csrc
timeout 1500 python -m pytest tests/test_mp2m_p3_gpu.py tests/test_layouts_gpu.py tests/test_nan_inputs_gpu.py -q -m gpu 2>&1 | tail -3
EXTRA="--no-cold-probes --rotate 1 --no-telemetry --points 1000000" REPS=3 STEPS=3 tools/ab_bench.sh "mp2m_p3:f64" $L/libcmx_prev.so $L/libcmx.so 2>&1 | tee gpurun_out/ab_r04_22.txt
echo finished
