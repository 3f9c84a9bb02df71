#!/bin/bash
# Round 5, GPU session 5: the full -m gpu suite on the packed build + the new cloud-diagnostics entry.
set -u
mkdir -p gpurun_out
timeout 2000 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 | tee gpurun_out/r05_s5_tests.txt
echo finished
