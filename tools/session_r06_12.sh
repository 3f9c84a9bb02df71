#!/bin/bash
# Round 6, GPU session 12: the Chen fall speeds of the 1-moment column step share one floored log2 ρ and take the finite-argument exponentials / positive-normal
# logarithms (cmx_mp1m_vel.hpp).  libcmx_head.so = eb54717 (the tree of the round's evidence), libcmx.so = the working tree.
set -u
L=cloudmicrophysics.jl_amd/csrc
timeout 1500 python -m pytest tests/test_column_gpu.py tests/test_mp1m_column.py tests/test_mp1m_gpu.py tests/test_nan_inputs_gpu.py tests/test_layouts_gpu.py -q -m gpu --tb=short 2>&1 | grep -E "Error|error|assert|passed|failed|FAILED|^E " | head -30
EXTRA="--no-cold-probes --rotate 1 --no-telemetry" REPS=3 STEPS=40 tools/ab_bench.sh "mp1m_column:f64 mp1m_column_lin:f64 mp1m_column:f32" $L/libcmx_head.so $L/libcmx.so 2>&1 | tee gpurun_out/ab_r06_12.txt
echo finished
