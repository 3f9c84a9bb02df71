#!/bin/bash
# Run tools/traffic_calib under rocprofv3, one counter set per pass (FETCH_SIZE and WRITE_SIZE cannot share a pass; the raw request counters they
# derive from go in passes of their own), and tabulate counter ÷ known bytes per access pattern:
#   usage: tools/traffic_calib.sh [round-tag]     →  gpurun_out/profiles/<round>_traffic_calibration.txt
set -u
R=${1:-r04}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/traffic_calib
mkdir -p "$OUT" "$ROOT/gpurun_out/profiles"
export TMPDIR=/tmp
[ -x tools/traffic_calib ] || make -s -C tools traffic_calib
cd /tmp
pass() { local name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o "$name" -- "$ROOT/tools/traffic_calib" > "$OUT/$name.log" 2>&1 || echo "pass $name failed"; }
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass rdreq TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum
pass wrreq TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
cd "$ROOT"
python3 tools/traffic_calib_summary.py "$OUT" | tee "gpurun_out/profiles/${R}_traffic_calibration.txt"
rm -rf "$OUT"/fetch "$OUT"/write "$OUT"/rdreq "$OUT"/wrreq
