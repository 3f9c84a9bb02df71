#!/bin/bash
# round-3 GPU session M: degenerate / edge states of the 1-moment and ARG kernels (finite-argument Float64 forms)
set -u
mkdir -p gpurun_out/r03m
timeout 1500 python -m pytest tests/test_mp1m_gpu.py::test_degenerate_states tests/test_arg2000_gpu.py::test_edge_states_updraft_and_cold -q -m gpu > gpurun_out/r03m/tests.log 2>&1
echo "tests rc=$?"; tail -40 gpurun_out/r03m/tests.log | cut -c1-220
