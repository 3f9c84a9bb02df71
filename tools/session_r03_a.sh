#!/bin/bash
# round-3 GPU session A: parity of the restructured 1-moment kernels, instruction-issue probe, same-box A/B of the 1-moment variants
set -u
mkdir -p gpurun_out/r03a
L=cloudmicrophysics.jl_amd/csrc
timeout 1500 python -m pytest tests/test_mp1m_gpu.py tests/test_mp1m_linearized.py tests/test_layouts_gpu.py tests/test_abi_caller.py tests/test_nan_inputs_gpu.py tests/test_abi.py -q -m gpu -x > gpurun_out/r03a/tests.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/r03a/tests.log
timeout 300 tools/valu_probe > gpurun_out/r03a/valu_probe.txt 2>&1; echo "probe rc=$?"
REPS=3 STEPS=30 tools/ab_bench.sh "mp1m:f32 mp1m_lin:f32 sb2006:f32" $PWD/$L/libcmx.so $PWD/$L/libcmx_m0.so $PWD/$L/libcmx_h.so $PWD/$L/libcmx_m0h.so 2>&1 | tee gpurun_out/r03a/ab_f32.txt
REPS=2 STEPS=15 tools/ab_bench.sh "mp1m:f64 mp1m_lin:f64" $PWD/$L/libcmx.so $PWD/$L/libcmx_m0.so 2>&1 | tee gpurun_out/r03a/ab_f64.txt
python bench.py --workload mp1m --dtype f32 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03a/bench_mp1m_f32.json 2>/dev/null
tail -c 600 gpurun_out/r03a/bench_mp1m_f32.json
