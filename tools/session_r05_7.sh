#!/bin/bash
# Round 5, GPU session 7: Float32 ARG activated number with the relative-accuracy erfc (libcmx) vs Abramowitz-Stegun 7.1.26 (libcmx_as): parity suite + same-box A/B.
set -u
L=cloudmicrophysics.jl_amd/csrc
timeout 1200 python -m pytest tests/test_arg2000_gpu.py -q -m gpu -x --tb=short 2>&1 | grep -E "Assert|assert|Error|passed|failed" | head -8
EXTRA="--no-cold-probes --rotate 1 --no-telemetry" REPS=3 STEPS=200 tools/ab_bench.sh "arg2000:f32 arg2000_columns:f32" $L/libcmx_as.so $L/libcmx.so 2>&1 | tee gpurun_out/ab_r05_7.txt
python - <<'PY'
import json
d=json.load(open('gpurun_out/parity_report.json'))
rows=[r for r in d['rows'] if 'ARG' in r['what'] and r['ft']=='f32' and 'N_act' in r['output']]
for r in sorted(rows,key=lambda r:r['frac_within'])[:5]:
    print(r['what'][:40], r['output'], 'frac', round(r['frac_within'],4), 'worst_wc', '%.2e'%r['worst_wellcond'])
PY
echo finished
