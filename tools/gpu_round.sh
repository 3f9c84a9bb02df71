#!/bin/bash
# One GPU-box session: full parity suite, every workload's bench line (f32 + f64), then profiles.
mkdir -p gpurun_out/bench
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 gpurun_out/gpu_tests.log
for wl in sb2006 icenuc mp0m mp1m arg2000; do
  for dt in f32 f64; do
    timeout 600 python bench.py --workload $wl --dtype $dt --steps 20 --warmup 3 > gpurun_out/bench/${wl}_${dt}.json 2> gpurun_out/bench/${wl}_${dt}.err
    echo "$wl $dt: $(python -c "import json,sys; d=json.loads(open('gpurun_out/bench/${wl}_${dt}.json').read().strip().splitlines()[-1]); print('%.3e pts/s  kern %.3f ms  frac %.3f  cpu %.3e' % (d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'], d.get('cpu_baseline',{}).get('value',0)))" 2>&1 | tail -1)"
  done
done
for dt in f32 f64; do
  timeout 600 python bench.py --workload p3 --dtype $dt --points 10000000 --steps 5 --warmup 1 > gpurun_out/bench/p3_${dt}.json 2> gpurun_out/bench/p3_${dt}.err
  timeout 600 python bench.py --workload p3_selfcol --dtype $dt --points 1000000 --steps 3 --warmup 1 > gpurun_out/bench/p3_selfcol_${dt}.json 2> gpurun_out/bench/p3_selfcol_${dt}.err
  timeout 600 python bench.py --workload mp1m_lin --dtype $dt --steps 10 --warmup 2 > gpurun_out/bench/mp1m_lin_${dt}.json 2> gpurun_out/bench/mp1m_lin_${dt}.err
  timeout 600 python bench.py --workload mp2m_p3 --dtype $dt --points 1000000 --steps 3 --warmup 1 > gpurun_out/bench/mp2m_p3_${dt}.json 2> gpurun_out/bench/mp2m_p3_${dt}.err
  timeout 600 python bench.py --workload sb2006_aos --dtype $dt --steps 20 --warmup 3 > gpurun_out/bench/sb2006_aos_${dt}.json 2> gpurun_out/bench/sb2006_aos_${dt}.err
  timeout 600 python bench.py --workload sb2006_fields --dtype $dt --steps 20 --warmup 3 > gpurun_out/bench/sb2006_fields_${dt}.json 2> gpurun_out/bench/sb2006_fields_${dt}.err
done
timeout 900 python bench.py > gpurun_out/bench/default.json 2> gpurun_out/bench/default.err
tools/profile.sh sb2006 f32 100000000 > /dev/null
tools/profile.sh mp0m f32 100000000 > /dev/null
tools/profile.sh sb2006 f64 100000000 > /dev/null
tools/profile.sh icenuc f32 100000000 > /dev/null
tools/profile.sh mp1m f32 100000000 > /dev/null
tools/profile.sh arg2000 f32 100000000 > /dev/null
tools/profile.sh p3 f64 10000000 > /dev/null
tools/profile.sh mp2m_p3 f64 1000000 r01 valu > /dev/null
tools/profile.sh sb2006_aos f32 100000000 > /dev/null
ls gpurun_out/profiles
