#!/bin/bash
# round-3 GPU session C: the column step + fitted Chen-Γ (parity), bench lines of the kernels they touch
set -u
mkdir -p gpurun_out/r03c
timeout 2000 python -m pytest tests/test_mp1m_column.py tests/test_sb2006_gpu.py tests/test_column_gpu.py tests/test_mp1m_gpu.py tests/test_mp2m_p3_gpu.py tests/test_layouts_gpu.py -q -m gpu -x > gpurun_out/r03c/tests.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/r03c/tests.log
b() { n=$1; shift; timeout 600 python bench.py "$@" --no-cpu-baseline > gpurun_out/r03c/bench_$n.json 2> gpurun_out/r03c/bench_$n.err
  python - gpurun_out/r03c/bench_$n.json <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); r=d['roofline']
    print(sys.argv[1].split('/')[-1], 'kern %.3f ms cold %.3f frac %.3f bound %s'%(r['kernel_ms'], d['cold_ms_first5'], r['frac'], r['bound']))
except Exception as e: print(sys.argv[1], 'NO RESULT', e)
PY
}
b mp1m_column_f32 --workload mp1m_column --dtype f32 --steps 10 --warmup 2
b mp1m_column_f64 --workload mp1m_column --dtype f64 --steps 10 --warmup 2
b mp1m_column_lin_f32 --workload mp1m_column_lin --dtype f32 --steps 10 --warmup 2
b sb2006_chen_f32 --workload sb2006_chen --dtype f32 --steps 20 --warmup 3
b sb2006_f32 --workload sb2006 --dtype f32 --steps 20 --warmup 5
b sb2006_f64 --workload sb2006 --dtype f64 --steps 20 --warmup 5
b sb2006_column_f32 --workload sb2006_column --dtype f32 --steps 20 --warmup 3
