#!/bin/bash
# A/B several builds of libcmx.so on ONE box (box-to-box variance is larger than most kernel changes):
#   tools/ab_bench.sh "<workload dtype> ..." libA.so libB.so ...      (interleaved, REPS rounds; prints kernel ms per build)
WLS=$1; shift
for rep in $(seq 1 ${REPS:-2}); do
  for wd in $WLS; do
    wl=${wd%%:*}; dt=${wd##*:}
    line="$wl $dt:"
    for lib in "$@"; do
      ms=$(CMX_LIB=$lib python bench.py --workload $wl --dtype $dt --steps ${STEPS:-20} --warmup 5 --no-cpu-baseline ${EXTRA:-} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f' % d['roofline']['kernel_ms'])")
      line="$line  $(basename $lib .so)=$ms"
    done
    echo "$line"
  done
done
