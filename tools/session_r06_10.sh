#!/bin/bash
# Round 6, GPU session 10: after session 9 — ρ ≤ 0 poisons every output of the Float64 1-moment entries; tests, then the instruction counters of the
# Float64 1-moment kernels (expected ≈ 540 per point, 727 before).
set -u
timeout 1500 python -m pytest tests/test_mp1m_gpu.py tests/test_mp1m_linearized.py tests/test_column_gpu.py tests/test_mp1m_column.py tests/test_nan_inputs_gpu.py tests/test_layouts_gpu.py tests/test_reference_suites_gpu.py -q -m gpu --tb=short 2>&1 | grep -E "Error|error|assert|passed|failed|FAILED|^E " | head -30
for wl in mp1m mp1m_column; do tools/profile.sh $wl f64 100000000 r06x valu 2>&1 | tail -3; done
cat gpurun_out/profiles/r06x_pmc_valu_mp1m_f64.json gpurun_out/profiles/r06x_pmc_valu_mp1m_column_f64.json 2>/dev/null
echo finished
