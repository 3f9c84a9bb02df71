#!/bin/bash
# Round 6, GPU session 3: Brent's zeroin on the device (shape solve and crossover solve).  The P3 test families, the mirrored reference suites, the exposure
# measurement, and an A/B of the P3 workloads against round 5's final tree.
#   libcmx.so        make -C cloudmicrophysics.jl_amd/csrc
#   libcmx_r05.so    tools/build_ref_variant.sh r05 c85d362
set -u
L=cloudmicrophysics.jl_amd/csrc
timeout 2400 python -m pytest tests/test_reference_suites_gpu.py tests/test_p3_gpu.py tests/test_p3_collisions_gpu.py tests/test_mp2m_p3_gpu.py tests/test_arg2000_gpu.py -q -m gpu --tb=short -s 2>&1 | grep -v Warning | grep -E "^\[|Error|error|assert|passed|failed|FAILED|^E " | head -80
EXTRA="--no-cold-probes --rotate 1 --no-telemetry" REPS=2 STEPS=20 tools/ab_bench.sh "p3:f64 p3:f32 mp2m_p3:f64 mp2m_p3:f32 p3_split:f64" $L/libcmx_r05.so $L/libcmx.so 2>&1 | tee gpurun_out/ab_r06_3.txt
cp gpurun_out/parity_report.json gpurun_out/parity_report_r06_3.json 2>/dev/null
echo finished
