#!/bin/bash
# One GPU-box session: full parity suite FIRST (a failing suite aborts the session: no numbers are produced from a build whose
# parity is red), then the rocprofv3 passes of every workload (kernel-trace over 1000 launches of the sweep kernels, FETCH_SIZE, WRITE_SIZE, SQ counters — the
# bench lines of the compute-bound workloads read their instruction counts from THESE passes), then every workload's bench line
# (f32 + f64), then the probes.  Each step's rc is checked; a failed bench or profile is reported and skipped, never summarised.
#   usage: tools/gpu_round.sh <round-tag, e.g. r04> [notests]
set -u
set -o pipefail
TAG=${1:-r06}
mkdir -p gpurun_out/bench
FAILED=0

if [ "${2:-}" != "notests" ]; then
  timeout 2400 python -m pytest tests -q -m gpu -x > gpurun_out/gpu_tests.log 2>&1
  rc=$?
  echo "gpu tests rc=$rc"; tail -3 gpurun_out/gpu_tests.log
  if [ $rc -ne 0 ]; then
    echo "ABORT: the GPU parity suite failed — no benchmark or profile is taken from this build"
    exit $rc
  fi
fi

summ() {  # one-line summary of a bench JSON file; prints nothing useful (and says so) if the file holds no JSON line
  python - "$1" <<'PY'
import json, sys
try:
    lines = [l for l in open(sys.argv[1]).read().splitlines() if l.strip().startswith("{")]
    d = json.loads(lines[-1])
    print('%.3e pts/s  step %.3f ms  kern %.3f ms  frac %.3f  cpu %.3e' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'],
          d['roofline']['frac'], d.get('cpu_baseline', {}).get('value', 0)))
except Exception as e:
    print('NO RESULT (%s)' % type(e).__name__)
PY
}

bench() {  # bench <name> <bench.py args…>
  local name=$1; shift
  timeout 900 python bench.py "$@" > gpurun_out/bench/${name}.json 2> gpurun_out/bench/${name}.err
  local rc=$?
  if [ $rc -ne 0 ]; then echo "$name: bench FAILED rc=$rc (see gpurun_out/bench/${name}.err)"; FAILED=$((FAILED+1)); return $rc; fi
  echo "$name: $(summ gpurun_out/bench/${name}.json)"
}

prof() {  # prof <workload> <dtype> <points> [valu]
  KT_STEPS=${KT_STEPS:-1000} tools/profile.sh "$1" "$2" "$3" "$TAG" "${4:-}" > gpurun_out/prof_${1}_${2}.log 2>&1
  local rc=$?
  if [ $rc -ne 0 ]; then echo "profile $1 $2 FAILED rc=$rc"; FAILED=$((FAILED+1)); fi
}

# ---- profiles first: the bench lines of the compute-bound workloads take their VALU instruction count from the PMC pass of THIS build
# (bench.py pmc_valu: same size, same source digest), so the summaries are copied into profiles/ of this box before any line is printed
prof sb2006 f32 100000000 valu
prof sb2006 f64 100000000 valu
prof sb2006_chen f32 100000000 valu
prof sb2006_column f32 100000000 valu
prof sb2006_column f64 100000000 valu
prof sb2006_fields f32 100000000 valu
prof sb2006_fields f64 100000000 valu
prof mp0m f32 100000000
prof cloud_diag f32 100000000
prof cloud_diag f64 100000000
prof icenuc f32 100000000
prof icenuc f64 100000000
prof mp1m f32 100000000 valu
prof mp1m f64 100000000 valu
prof mp1m_lin f32 100000000 valu
prof mp1m_lin f64 100000000 valu
prof mp1m_column f32 100000000 valu
prof mp1m_column f64 100000000 valu
prof mp1m_column_lin f32 100000000 valu
prof arg2000 f32 100000000 valu
prof arg2000 f64 100000000 valu
prof arg2000_columns f32 100000000 valu
prof arg2000_columns f64 100000000 valu
KT_STEPS=10 prof p3 f32 10000000 valu
KT_STEPS=10 prof p3 f64 10000000 valu
KT_STEPS=10 prof p3_split f64 10000000
KT_STEPS=6 prof p3_selfcol f32 1000000 valu
KT_STEPS=6 prof p3_selfcol f64 1000000 valu
KT_STEPS=10 prof mp2m_p3 f32 1000000 valu
KT_STEPS=10 prof mp2m_p3 f64 1000000 valu
prof sb2006_aos f32 100000000
# the stall / occupancy counters of the Float32 kernels VERDICT r03 item 4 names (three more PMC passes each; tools/sq_pass.sh)
for wl in sb2006_column sb2006_chen sb2006_fields arg2000 mp1m mp1m_lin; do
  tools/sq_pass.sh $wl f32 > gpurun_out/profiles/${TAG}_sq_${wl}_f32.txt 2>&1 || { echo "sq_pass $wl FAILED"; FAILED=$((FAILED+1)); }
done
mkdir -p profiles
cp gpurun_out/profiles/${TAG}_pmc_*.json gpurun_out/profiles/${TAG}_kernel_stats_*.csv profiles/ 2>/dev/null

bench default_driver --steps 20 --warmup 5
bench default
for wl in sb2006 sb2006_column icenuc mp0m cloud_diag mp1m arg2000 arg2000_columns mp1m_lin mp1m_column sb2006_aos sb2006_fields; do
  for dt in f32 f64; do
    bench ${wl}_${dt} --workload $wl --dtype $dt --steps 100 --warmup 20
  done
done
bench sb2006_chen_f32 --workload sb2006_chen --dtype f32 --steps 100 --warmup 20
bench mp1m_column_lin_f32 --workload mp1m_column_lin --dtype f32 --steps 100 --warmup 20
for dt in f32 f64; do
  bench p3_${dt} --workload p3 --dtype $dt --points 10000000 --steps 5 --warmup 1
  bench p3_split_${dt} --workload p3_split --dtype $dt --points 10000000 --steps 5 --warmup 1
  bench p3_selfcol_${dt} --workload p3_selfcol --dtype $dt --points 1000000 --steps 3 --warmup 1
  bench mp2m_p3_${dt} --workload mp2m_p3 --dtype $dt --points 1000000 --steps 3 --warmup 1
done
cp gpurun_out/parity_report.json gpurun_out/profiles/${TAG}_parity_report.json 2>/dev/null
python tools/kernel_resources.py > gpurun_out/profiles/${TAG}_kernel_resources.txt 2>&1
# instruction issue rates and dependent-issue latency (tools/valu_probe.hip), stream ceilings (tools/stream_probe.hip quick)
[ -x tools/traffic_calib ] && tools/traffic_calib.sh ${TAG} > /dev/null 2>&1
[ -x tools/valu_probe ] && timeout 120 tools/valu_probe > gpurun_out/profiles/${TAG}_probe_valu.txt 2>&1
[ -x tools/stream_probe ] && timeout 300 tools/stream_probe 100000000 20 slab pattern quick > gpurun_out/profiles/${TAG}_probe_streams.txt 2>&1
ls gpurun_out/profiles
echo "failed steps: $FAILED"
exit $([ $FAILED -eq 0 ] && echo 0 || echo 1)
