#!/bin/bash
# round-3 GPU session K: 2M + P3 fields entry, default parity fractions on the ragged column shapes
set -u
mkdir -p gpurun_out/r03k
timeout 3000 python -m pytest tests/test_mp2m_p3_gpu.py tests/test_column_gpu.py tests/test_mp1m_column.py tests/test_p3_collisions_gpu.py -q -m gpu > gpurun_out/r03k/tests.log 2>&1
echo "tests rc=$?"; tail -15 gpurun_out/r03k/tests.log
