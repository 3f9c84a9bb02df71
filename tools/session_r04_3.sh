#!/bin/bash
# Round 4, GPU session 3: full suite on the build with the 32-bit index forms (layout + column kernels), the log-domain Chen-2022 coefficients
# and the two-pass ARG-columns kernel; then PMC + kernel stats + bench of the kernels those changes touch.
set -u
TAG=r04b
mkdir -p gpurun_out/bench gpurun_out/profiles
timeout 2700 python -m pytest tests -q -m gpu > gpurun_out/gpu_tests.log 2>&1
echo "gpu tests rc=$?"; tail -8 gpurun_out/gpu_tests.log
prof() { KT_STEPS=${KT_STEPS:-200} tools/profile.sh "$1" "$2" "$3" "$TAG" "${4:-}" > gpurun_out/prof_${1}_${2}.log 2>&1 || echo "profile $1 $2 FAILED"; }
prof sb2006_column f32 100000000 valu
prof sb2006_chen f32 100000000 valu
prof sb2006_fields f32 100000000 valu
prof sb2006_fields f64 100000000 valu
prof sb2006_column f64 100000000 valu
prof mp1m_column f32 100000000 valu
prof arg2000 f32 100000000 valu
prof mp1m f32 100000000 valu
prof sb2006 f32 100000000 valu
for wl in sb2006 sb2006_chen sb2006_column sb2006_fields mp1m arg2000 mp1m_column; do for dt in f32 f64; do
  timeout 600 python bench.py --workload $wl --dtype $dt --steps 20 --warmup 3 --no-cpu-baseline --no-cold-probes > gpurun_out/bench/${wl}_${dt}.json 2> gpurun_out/bench/${wl}_${dt}.err
  python - $wl $dt <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(f'gpurun_out/bench/{sys.argv[1]}_{sys.argv[2]}.json') if l.startswith('{')][-1])
    print(sys.argv[1], sys.argv[2], 'same %.4f rot %s kern %.4f frac %.3f' % (d['same_buffer_ms_per_step'], d['rotating_ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))
except Exception as e:
    print(sys.argv[1], 'no line', e)
PY
done; done
python - <<'PY'
import json, glob, csv
for f in sorted(glob.glob('gpurun_out/profiles/r04b_kernel_stats_*.csv')):
    rows = list(csv.reader(open(f)))[1:]
    for r in rows[:2]:
        print(f.split('kernel_stats_')[1], r[0][:70], 'calls', r[1], 'avg', r[3], 'min', r[5], 'max', r[6])
for f in sorted(glob.glob('gpurun_out/profiles/r04b_pmc_valu_*.json')):
    d = json.load(open(f))
    for k, v in d['kernels'].items():
        if v.get('valu_instructions_per_point', 0) > 1:
            print(f.split('pmc_valu_')[1], k[:60], 'instr/pt %.1f' % v['valu_instructions_per_point'], 'util4 %.3f' % v['valu_issue_utilisation'])
PY
cp gpurun_out/parity_report.json gpurun_out/profiles/${TAG}_parity_report.json 2>/dev/null
du -sh gpurun_out
