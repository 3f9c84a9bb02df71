#!/bin/bash
# round-3 GPU session L: segmented-column strides of the collision kernel through an LDS table — P3 tests + the 2M + P3 step
set -u
mkdir -p gpurun_out/r03l
timeout 3000 python -m pytest tests/test_mp2m_p3_gpu.py tests/test_p3_collisions_gpu.py -q -m gpu > gpurun_out/r03l/tests.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/r03l/tests.log
for dt in f64 f32; do for r in 1 2; do python bench.py --workload mp2m_p3 --dtype $dt --points 1000000 --steps 5 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mp2m_p3 $dt %.3f ms' % d['roofline']['kernel_ms'])"; done; done
