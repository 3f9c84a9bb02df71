#!/bin/bash
# SQ counter passes for one workload (stall analysis of the compute-bound kernels): tools/sq_pass.sh <workload> <dtype> [lib]
#   → gpurun_out/sq_<wl>_<dt>/pass*.csv (+ counters.txt: what rocprofv3 lists on this box)
WL=$1; DT=$2; LIB=${3:-}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/sq_${WL}_${DT}; mkdir -p "$OUT"
export TMPDIR=/tmp
[ -n "$LIB" ] && export CMX_LIB=$ROOT/$LIB
ARGS="$ROOT/bench.py --workload $WL --dtype $DT --steps 3 --warmup 1 --no-cpu-baseline --no-telemetry --no-cold-probes --rotate 1"
cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z0-9_]*" | sort -u > "$OUT/counters.txt"
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" \
           "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d "$OUT/p$i" -o p -- python3 $ARGS > "$OUT/p$i.log" 2>&1 || echo "pass $i failed (see $OUT/p$i.log)"
done
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][:70]
        tot[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
for k, d in tot.items():
    if 'cmx::' not in k:
        continue
    print(k)
    m = {c: v / n[(k, c)] for c, v in d.items()}
    for c, v in sorted(m.items()):
        print(f'   {c:28s} {v:16.1f} per launch')
    wc = m.get('SQ_WAVE_CYCLES')
    if wc:      # MI355X_MICROARCH.md: ACTIVE_INST_ANY + WAIT_INST_ANY + WAIT_ANY ~ WAVE_CYCLES (disjoint states of a resident wave)
        print('   -- share of resident-wave time: issuing %.2f, issue-stalled (pipe busy / dependency) %.2f, parked on s_waitcnt / barrier %.2f'
              % (m.get('SQ_ACTIVE_INST_ANY', 0) / wc, m.get('SQ_WAIT_INST_ANY', 0) / wc, m.get('SQ_WAIT_ANY', 0) / wc))
    if m.get('SQ_INSTS_VALU') and m.get('SQ_ACTIVE_INST_VALU'):
        print('   -- VALU-active quad-cycles per VALU instruction %.2f (x 4 = clocks of issue per instruction: %.2f)'
              % (m['SQ_ACTIVE_INST_VALU'] / m['SQ_INSTS_VALU'], 4 * m['SQ_ACTIVE_INST_VALU'] / m['SQ_INSTS_VALU']))
    if m.get('SQ_BUSY_CYCLES') and wc:
        print('   -- resident waves per SIMD while busy: %.1f  (SQ_WAVE_CYCLES / (SQ_BUSY_CYCLES / 32 x 1024 SIMDs / 4))' % (wc / (m['SQ_BUSY_CYCLES'] / 32 * 1024 / 4)))
PY
# the raw per-dispatch CSVs stay on the box (gpurun merges at most 64 MiB back)
rm -rf "$OUT"/p1 "$OUT"/p2 "$OUT"/p3
