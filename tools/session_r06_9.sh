#!/bin/bash
# Round 6, GPU session 9: the logarithms of ρ·q of an absent species (q = 0: half the points of the bench's field, as of a real one) sent nearly every wave of
# the Float64 1-moment kernels through the rescue block of lean::log2 (55 instructions per logarithm).  log2_floored (cmx_math.hpp): floor at the smallest
# normal number + the main path alone; the logistic integral's logarithm likewise (its argument is ≥ 1 − e^{−k}).  libcmx_base.so = HEAD before the change
# (tools/build_ref_variant.sh base a06c737), libcmx.so = the working tree.
set -u
L=cloudmicrophysics.jl_amd/csrc
timeout 1500 python -m pytest tests/test_mp1m_gpu.py tests/test_mp1m_linearized.py tests/test_column_gpu.py tests/test_mp1m_column.py tests/test_nan_inputs_gpu.py tests/test_layouts_gpu.py tests/test_reference_suites_gpu.py -q -m gpu --tb=short -x 2>&1 | grep -E "Error|error|assert|passed|failed|FAILED|^E " | head -30
EXTRA="--no-cold-probes --rotate 1 --no-telemetry" REPS=2 STEPS=40 tools/ab_bench.sh "mp1m:f64 mp1m_lin:f64 mp1m_column:f64 mp1m_column_lin:f64 mp1m:f32 mp1m_column:f32" $L/libcmx_base.so $L/libcmx.so 2>&1 | tee gpurun_out/ab_r06_9.txt
echo finished
