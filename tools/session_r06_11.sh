#!/bin/bash
# Round 6, GPU session 11: session 9's change plus — Float64 1-moment point: positive-argument roots and one-step reciprocals (ρ ≤ 0 is poisoned), finite-form
# exponentials in the 1-moment fall speeds, floored logarithms in the cloud diagnostics.  Whole GPU suite, then same-box A/B against HEAD a06c737.
set -u
L=cloudmicrophysics.jl_amd/csrc
timeout 2400 python -m pytest tests -q -m gpu --tb=short -x 2>&1 | grep -E "Error|error|assert|passed|failed|FAILED|^E " | head -30
EXTRA="--no-cold-probes --rotate 1 --no-telemetry" REPS=2 STEPS=40 tools/ab_bench.sh "mp1m:f64 mp1m_lin:f64 mp1m_column:f64 mp1m_column_lin:f64 cloud_diag:f64 mp1m:f32" $L/libcmx_base.so $L/libcmx.so 2>&1 | tee gpurun_out/ab_r06_11.txt
echo finished
