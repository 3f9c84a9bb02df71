#!/bin/bash
# Round 4, GPU session 10: bench.py telemetry from sysfs (no child process) against rocm-smi read from the shell; the bench contract tests.
set -u
ls /sys/class/drm/ | head; for d in /sys/class/drm/card*/device; do echo "$d vendor $(cat $d/vendor 2>/dev/null) $(readlink -f $d | xargs basename)"; ls $d | grep -E "pp_dpm_sclk|pp_dpm_mclk|hwmon|gpu_metrics" | tr '\n' ' '; echo; cat $d/pp_dpm_sclk 2>/dev/null | head -4; ls $d/hwmon/*/ 2>/dev/null | grep power | tr '\n' ' '; echo; done
python bench.py --workload sb2006 --dtype f64 --steps 20 --warmup 3 --no-cpu-baseline --no-cold-probes 2>gpurun_out/tel.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('telemetry', d['telemetry'])"
tail -3 gpurun_out/tel.err
( python bench.py --workload sb2006 --dtype f64 --steps 3000 --warmup 3 --no-cpu-baseline --no-cold-probes --no-telemetry > /dev/null 2>&1 & ); sleep 9; rocm-smi --showclocks --showpower | grep -E "sclk|Power"; sleep 4
timeout 900 python -m pytest tests/test_bench_gpu.py tests/test_nan_inputs_gpu.py -q -m gpu 2>&1 | tail -5
cat gpurun_out/.graft_exec_refused 2>/dev/null | wc -l
echo finished
