#!/bin/bash
# round-3 GPU session D: the full -m gpu suite
set -u
mkdir -p gpurun_out/r03d
timeout 3000 python -m pytest tests -q -m gpu > gpurun_out/r03d/tests.log 2>&1
echo "tests rc=$?"; tail -15 gpurun_out/r03d/tests.log
