#!/bin/bash
# Round 4, GPU session 7: engine clock under each workload (rocm-smi while the kernel loops), GRBM_GUI_ACTIVE with the dispatch's own
# timestamps, and the distribution-tools tests after the Float32-underflow fix of the test.
set -u
mkdir -p gpurun_out/profiles
timeout 600 python -m pytest tests/test_distribution_tools.py -q -m gpu 2>&1 | tail -3
rocm-smi --showclocks 2>&1 | head -20
STEPS=6000 tools/clock_probe.sh "sb2006:f32 sb2006:f64 mp1m:f64 arg2000:f64 arg2000:f32 sb2006_column:f32 mp1m_lin:f64 icenuc:f32" 2>&1 | tee gpurun_out/profiles/r04_clock_probe.txt
STEPS=300 tools/clock_probe.sh "p3:f64:10000000" 2>&1 | tee -a gpurun_out/profiles/r04_clock_probe.txt
STEPS=400 LEAD=9 tools/clock_probe.sh "mp2m_p3:f64:1000000" 2>&1 | tee -a gpurun_out/profiles/r04_clock_probe.txt
export TMPDIR=/tmp
ROOT=$(pwd)
cd /tmp
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $ROOT/gpurun_out/grbm -o g -- python3 $ROOT/bench.py --workload sb2006 --dtype f64 --steps 30 --warmup 5 --no-cpu-baseline --rotate 1 --no-cold-probes > $ROOT/gpurun_out/grbm.log 2>&1
cd $ROOT
find gpurun_out/grbm -name "*.csv" | head; for f in $(find gpurun_out/grbm -name "*.csv"); do echo "== $f"; head -3 $f | cut -c1-600; done
python3 - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/grbm/**/*counter_collection.csv', recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if 'sb2006_tendencies' in r['Kernel_Name']]
    print(len(rows), 'rows; columns', list(rows[0].keys()) if rows else None)
    g = [float(r['Counter_Value']) for r in rows if r['Counter_Name'] == 'GRBM_GUI_ACTIVE']
    if g: print('GRBM_GUI_ACTIVE per launch: first 5', g[:5], 'last 5', g[-5:])
for f in glob.glob('gpurun_out/grbm/**/*kernel_trace.csv', recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if 'sb2006_tendencies' in r['Kernel_Name']]
    d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) for r in rows]
    print('durations ns under --pmc: first 5', d[:5], 'last 5', d[-5:])
PY
rm -rf gpurun_out/grbm
echo finished
