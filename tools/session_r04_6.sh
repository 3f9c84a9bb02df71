#!/bin/bash
# Round 4, GPU session 6: the DistributionTools / PSD entries on the device; FETCH_SIZE / WRITE_SIZE calibrated on known byte counts in the access
# patterns of the collision kernels (tools/traffic_calib.hip); the raw TCC request counters of the 2M + P3 and self-collection workloads.
set -u
mkdir -p gpurun_out/bench gpurun_out/profiles
timeout 1500 python -m pytest tests/test_distribution_tools.py tests/test_mp2m_p3_gpu.py tests/test_lean_math.py tests/test_utilities_gpu.py -q -m gpu 2>&1 | tail -8
tools/traffic_calib.sh r04 2>&1 | tail -20
export TMPDIR=/tmp
ROOT=$(pwd)
raw() {  # raw <workload> <dtype> <points>: the request counters FETCH_SIZE / WRITE_SIZE are derived from, per launch of the cmx kernels
  local out=$ROOT/gpurun_out/raw_$1_$2; mkdir -p "$out"
  local args="$ROOT/bench.py --workload $1 --dtype $2 --points $3 --steps 3 --warmup 1 --no-cpu-baseline --rotate 1 --no-cold-probes"
  ( cd /tmp
    timeout 600 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum --output-format csv -d "$out/rd" -o rd -- python3 $args > "$out/rd.log" 2>&1
    timeout 600 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d "$out/wr" -o wr -- python3 $args > "$out/wr.log" 2>&1
    timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d "$out/l2" -o l2 -- python3 $args > "$out/l2.log" 2>&1 )
  python3 - "$out" "$1 $2" <<'PY'
import csv, glob, sys, collections
out, what = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'cmx::' in r['Kernel_Name']:
            acc[r['Kernel_Name'].split('(')[0][:90]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(what, k)
    for c, v in sorted(d.items()):
        print(f'   {c:26s} {sum(v) / len(v):16.1f} per launch ({len(v)} launches)')
PY
  rm -rf "$out/rd" "$out/wr" "$out/l2"
}
raw mp2m_p3 f64 1000000 2>&1 | tee gpurun_out/profiles/r04_raw_requests_mp2m_p3_f64.txt
raw mp2m_p3 f32 1000000 2>&1 | tee gpurun_out/profiles/r04_raw_requests_mp2m_p3_f32.txt
raw p3_selfcol f64 1000000 2>&1 | tee gpurun_out/profiles/r04_raw_requests_p3_selfcol_f64.txt
raw sb2006 f32 100000000 2>&1 | tee gpurun_out/profiles/r04_raw_requests_sb2006_f32.txt
for dt in f64 f32; do python bench.py --workload mp2m_p3 --dtype $dt --points 1000000 --steps 3 --warmup 1 --no-cpu-baseline --no-cold-probes --rotate 1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('mp2m_p3 $dt kernel_ms', d['roofline']['kernel_ms'])"; done
echo finished
