#!/bin/bash
# round-3 GPU session B: the fused 1-moment column step (parity + bench), 128-lane A/B of the 1-moment tendencies kernel
set -u
mkdir -p gpurun_out/r03b
L=$PWD/cloudmicrophysics.jl_amd/csrc
timeout 1500 python -m pytest tests/test_mp1m_column.py tests/test_mp1m_gpu.py tests/test_layouts_gpu.py -q -m gpu -x > gpurun_out/r03b/tests.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/r03b/tests.log
for wl in mp1m_column mp1m_column_lin; do for dt in f32 f64; do
  timeout 600 python bench.py --workload $wl --dtype $dt --steps 10 --warmup 2 > gpurun_out/r03b/bench_${wl}_${dt}.json 2> gpurun_out/r03b/bench_${wl}_${dt}.err
  python - gpurun_out/r03b/bench_${wl}_${dt}.json <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); r=d['roofline']
    print(sys.argv[1], 'kern %.3f ms cold %.3f frac %.3f bound %s cpu %.3e'%(r['kernel_ms'], d['cold_ms_first5'], r['frac'], r['bound'], d.get('cpu_baseline',{}).get('value',0)))
except Exception as e: print(sys.argv[1], 'NO RESULT', e)
PY
done; done
REPS=3 STEPS=30 tools/ab_bench.sh "mp1m:f32 mp1m:f64" $L/libcmx.so $L/libcmx_b128.so 2>&1 | tee gpurun_out/r03b/ab_block.txt
