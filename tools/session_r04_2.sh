#!/bin/bash
# Round 4, GPU session 2: the two tests that failed in session 1 (verbose), the new ARG-columns kernel (tests + bench + resources), and the
# counters of VERDICT r03 item 4 again (session 1's output exceeded the 64-MiB merge limit: tools/sq_pass.sh now prunes its raw CSVs).
set -u
TAG=r04a
mkdir -p gpurun_out/bench gpurun_out/profiles
timeout 900 python -m pytest tests/test_utilities_gpu.py tests/test_sb2006_gpu.py::test_column_sums tests/test_sb2006_gpu.py::test_column_sums_are_deterministic tests/test_arg2000_gpu.py tests/test_abi_caller.py -q -m gpu -x 2>&1 | tail -40 > gpurun_out/gpu_tests_subset.log; cat gpurun_out/gpu_tests_subset.log
prof() { KT_STEPS=${KT_STEPS:-40} tools/profile.sh "$1" "$2" "$3" "$TAG" "${4:-}" > gpurun_out/prof_${1}_${2}.log 2>&1 || echo "profile $1 $2 FAILED"; }
prof sb2006_column f32 100000000 valu
prof sb2006_chen f32 100000000 valu
prof sb2006_fields f32 100000000 valu
prof arg2000 f32 100000000 valu
prof arg2000 f64 100000000 valu
prof mp1m f32 100000000 valu
prof arg2000_columns f32 100000000 valu
prof arg2000_columns f64 100000000 valu
for wl in arg2000 mp1m sb2006_column sb2006_chen; do tools/sq_pass.sh $wl f32 > gpurun_out/sq_${wl}_f32.txt 2>&1; done
for wl in arg2000 arg2000_columns; do for dt in f32 f64; do
  timeout 600 python bench.py --workload $wl --dtype $dt --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench/${wl}_${dt}.json 2> gpurun_out/bench/${wl}_${dt}.err
  python - $wl $dt <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(f'gpurun_out/bench/{sys.argv[1]}_{sys.argv[2]}.json') if l.startswith('{')][-1])
    print(sys.argv[1], sys.argv[2], 'same %.4f rot %s kern %.4f frac %.3f' % (d['same_buffer_ms_per_step'], d['rotating_ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))
except Exception as e:
    print(sys.argv[1], 'no line', e)
PY
done; done
du -sh gpurun_out
