#!/bin/bash
# Round 6, GPU session 4: the tests that failed in session 3 after their fixes (sink amplification, two-accumulation equality, exposure thresholds), the diagnostics
# fix, and the DYNAMIC instruction cost of every Float64 elementary function (per-dispatch SQ_INSTS_VALU of cmx_lean_eval_literal_f64, tools/lean_cost_run.py).
set -u
ROOT=$(pwd)
timeout 1500 python -m pytest tests/test_reference_suites_gpu.py tests/test_p3_collisions_gpu.py tests/test_arg2000_gpu.py tests/test_cloud_diagnostics.py tests/test_lean_math.py -q -m gpu --tb=short -s 2>&1 | grep -v Warning | grep -E "^\[|by root|by budget|Error|error|assert|passed|failed|FAILED|^E " | head -60
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d "$ROOT/gpurun_out/lean_cost" -o p -- python3 "$ROOT/tools/lean_cost_run.py" > "$ROOT/gpurun_out/lean_cost.log" 2>&1
cd "$ROOT"
tail -2 gpurun_out/lean_cost.log
python3 - <<'PY'
import csv, glob
rows = []
for f in glob.glob('gpurun_out/lean_cost/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'lean_eval_kernel' in r['Kernel_Name']:
            rows.append((int(r['Dispatch_Id']), r['Counter_Name'], float(r['Counter_Value'])))
rows.sort()
with open('gpurun_out/lean_cost_counters.csv', 'w') as o:
    o.write('dispatch,counter,value\n')
    for d, c, v in rows:
        o.write(f'{d},{c},{v}\n')
print(len(rows), 'counter rows')
PY
rm -rf gpurun_out/lean_cost
cp gpurun_out/parity_report.json gpurun_out/parity_report_r06_4.json 2>/dev/null
echo finished
