#!/bin/bash
# Round 4, GPU session 1: the whole -m gpu suite on the round's first build (new: gamma_inc entries, deterministic column sums, parity
# rows for every family, argument validation), the rho = 0 probe, the driver-form bench line with the rotating region and the cold
# probes, and the counters VERDICT r03 item 4 asks for BEFORE any kernel is touched (sb2006_column / sb2006_chen / sb2006_fields / arg2000 /
# mp1m Float32: kernel stats + traffic + SQ counters; arg2000_columns both float types).
set -u
TAG=r04a
mkdir -p gpurun_out/bench gpurun_out/profiles
timeout 2700 python -m pytest tests -q -m gpu > gpurun_out/gpu_tests.log 2>&1
echo "gpu tests rc=$?"; tail -15 gpurun_out/gpu_tests.log
timeout 300 python tools/rho_zero_probe.py > gpurun_out/rho_zero_probe.txt 2>&1; echo "rho probe rc=$?"; cat gpurun_out/rho_zero_probe.txt
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench/default_driver.json 2> gpurun_out/bench/default_driver.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    d = json.loads([l for l in open('gpurun_out/bench/default_driver.json') if l.startswith('{')][-1])
    print({k: d[k] for k in ('value', 'ms_per_step', 'same_buffer_ms_per_step', 'rotating_ms_per_step', 'value_uses', 'cold_ms_first5', 'cold', 'ranks_kernel_ms')})
    print(d['roofline']); print(d.get('cpu_baseline'))
except Exception as e:
    print('no bench line', e)
PY
prof() { KT_STEPS=${KT_STEPS:-40} tools/profile.sh "$1" "$2" "$3" "$TAG" "${4:-}" > gpurun_out/prof_${1}_${2}.log 2>&1 || echo "profile $1 $2 FAILED"; }
prof sb2006_column f32 100000000 valu
prof sb2006_chen f32 100000000 valu
prof sb2006_fields f32 100000000 valu
prof arg2000 f32 100000000 valu
prof mp1m f32 100000000 valu
prof arg2000_columns f32 100000000 valu
prof arg2000_columns f64 100000000 valu
for wl in arg2000 mp1m sb2006_column; do tools/sq_pass.sh $wl f32 > gpurun_out/sq_${wl}_f32.txt 2>&1; done
for wl in sb2006_chen sb2006_column arg2000 mp1m sb2006_fields arg2000_columns; do
  timeout 600 python bench.py --workload $wl --dtype f32 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench/${wl}_f32.json 2> gpurun_out/bench/${wl}_f32.err
  python - $wl <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(f'gpurun_out/bench/{sys.argv[1]}_f32.json') if l.startswith('{')][-1])
    print(sys.argv[1], 'same %.4f rot %s kern %.4f frac %.3f cold5 %.3f' % (d['same_buffer_ms_per_step'], d['rotating_ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['cold_ms_first5']), d['cold']['probes'])
except Exception as e:
    print(sys.argv[1], 'no line', e)
PY
done
cp gpurun_out/parity_report.json gpurun_out/profiles/${TAG}_parity_report.json 2>/dev/null
python tools/kernel_resources.py > gpurun_out/profiles/${TAG}_kernel_resources.txt 2>&1
ls gpurun_out/profiles | head -40
