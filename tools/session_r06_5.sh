#!/bin/bash
# Round 6, GPU session 5: the Float32 fall-speed / melting node loops of p3_velocity_kernel two nodes at a time in packed arithmetic (libcmx) against one node at
# a time (p3pk0) and round 5's final tree (r05).  P3 parity suites first.
#   libcmx.so          make -C cloudmicrophysics.jl_amd/csrc
#   libcmx_p3pk0.so    tools/build_variant.sh p3pk0 -DCMX_P3_F32_PACKED_NODES=0
#   libcmx_r05.so      tools/build_ref_variant.sh r05 c85d362
set -u
L=cloudmicrophysics.jl_amd/csrc
timeout 1500 python -m pytest tests/test_p3_gpu.py tests/test_mp2m_p3_gpu.py tests/test_layouts_gpu.py tests/test_mp1m_linearized.py -q -m gpu --tb=short 2>&1 | grep -E "Error|error|assert|passed|failed|FAILED|^E " | head -20
EXTRA="--no-cold-probes --rotate 1 --no-telemetry" REPS=3 STEPS=20 tools/ab_bench.sh "p3:f32 p3_split:f32 p3:f64" $L/libcmx_r05.so $L/libcmx_p3pk0.so $L/libcmx.so 2>&1 | tee gpurun_out/ab_r06_5.txt
echo finished
