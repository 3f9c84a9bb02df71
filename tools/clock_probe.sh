#!/bin/bash
# What engine clock do the kernels run at?  One bench.py workload loops in the background while rocm-smi samples sclk / mclk / power:
#   usage: tools/clock_probe.sh "<workload>:<dtype>[:points] ..."  →  stdout (one block per workload)
# (the VALU-issue estimates of tools/isa_cost.py and bench.py's valu ceiling assume the 2.4 GHz spec clock)
for wd in $1; do
  IFS=: read wl dt pts <<< "$wd"
  pts=${pts:-100000000}
  steps=${STEPS:-4000}
  python bench.py --workload $wl --dtype $dt --points $pts --steps $steps --warmup 5 --no-cpu-baseline --no-cold-probes --rotate 1 > /tmp/clock_probe_${wl}_${dt}.json 2>/dev/null &
  pid=$!
  sleep ${LEAD:-6}          # import torch, build inputs
  echo "== $wl $dt ($pts points)"
  for i in 1 2 3 4 5 6; do
    kill -0 $pid 2>/dev/null || break
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|socclk|Power" | sed 's/^GPU\[0\]\s*: //' | tr '\n' ';'
    echo
    sleep 0.7
  done
  wait $pid
  python - /tmp/clock_probe_${wl}_${dt}.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
    print('   kernel_ms %.4f over %d steps' % (d['roofline']['kernel_ms'], d['steps']))
except Exception as e:
    print('   no bench line', e)
PY
done
