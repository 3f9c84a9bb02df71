#!/bin/bash
# Profile one bench.py workload on the GPU box: kernel-trace stats + two separate PMC passes (FETCH_SIZE, WRITE_SIZE)
# as MI355X_MICROARCH.md §HBM prescribes; summaries land in gpurun_out/profiles/ (copy them into profiles/ to commit).
#   usage: tools/profile.sh <workload> <dtype> <points> [round-tag] [valu]
set -u
WL=$1; DT=$2; N=$3; R=${4:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_${WL}_${DT}
mkdir -p "$OUT" "$ROOT/gpurun_out/profiles"
export TMPDIR=/tmp
# counter passes: one buffer set, no probes (a per-launch counter does not depend on which buffers the launch sweeps)
ARGS="$ROOT/bench.py --workload $WL --dtype $DT --points $N --steps 5 --warmup 1 --no-cpu-baseline --rotate 1 --no-cold-probes --no-telemetry"
# the kernel-trace pass runs enough launches for its average to be the steady-state duration bench.py reports
# KT_ROTATE (default 1): with K > 1 the traced launches include the rotating region of bench.py (K disjoint buffer sets)
KTARGS="$ROOT/bench.py --workload $WL --dtype $DT --points $N --steps ${KT_STEPS:-40} --warmup 5 --no-cpu-baseline --rotate ${KT_ROTATE:-1} --no-cold-probes --no-telemetry"
run() {  # run <what> <cmd…>: a failed or timed-out profiler pass ends the script BEFORE any summary is written from partial data
  local what=$1; shift
  "$@"
  local rc=$?
  if [ $rc -ne 0 ]; then echo "profile.sh: $what failed (rc=$rc) for $WL $DT — no summary written" >&2; exit $rc; fi
}
cd /tmp
run kernel-trace timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o kt -- python3 $KTARGS > "$OUT/kt.log" 2>&1
run pmc-fetch timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o fetch -- python3 $ARGS > "$OUT/fetch.log" 2>&1
run pmc-write timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o write -- python3 $ARGS > "$OUT/write.log" 2>&1
cd "$ROOT"
run summary python3 tools/pmc_summary.py "$WL" "$DT" "$N" "$OUT" "$R"
# compute-bound workloads: one more pass with the SQ instruction/cycle counters (VALU issue utilisation)
if [ "${5:-}" = "valu" ]; then
  cd /tmp
  run pmc-sq timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_INST_ANY SQ_INSTS_VALU_TRANS GRBM_GUI_ACTIVE --output-format csv -d "$OUT/sq" -o sq -- python3 $ARGS > "$OUT/sq.log" 2>&1
  cd "$ROOT"
  run valu-summary python3 tools/pmc_summary.py "$WL" "$DT" "$N" "$OUT" "$R" valu
fi
# the raw rocprofv3 output (per-dispatch CSVs, several MB per pass) stays on the box: gpurun merges at most 64 MiB back, and a session
# profiles two dozen workloads; the summaries above and the pass logs are what is kept
rm -rf "$OUT/kt" "$OUT/fetch" "$OUT/write" "$OUT/sq"
