#!/bin/bash
# Round 5, GPU session 6: is the north star's one-point-at-a-time choice box-dependent?  (The round's evidence session landed on a box where the packed Chen sweep,
# which does strictly more work, ran FASTER than the one-point SB2006 sweep.)  sb2006 / sb2006_chen Float32: libcmx (north star one point at a time) vs packed at
# four waves per SIMD with phase-local constants (pkw4) and with the constants in SGPRs (pkw4np); engine clock and package power printed per line.
set -u
L=cloudmicrophysics.jl_amd/csrc
mkdir -p gpurun_out
for rep in 1 2 3 4; do
  for wl in sb2006 sb2006_chen; do
    for lib in libcmx libcmx_pkw4 libcmx_pkw4np; do
      CMX_LIB=$L/$lib.so python bench.py --workload $wl --dtype f32 --steps 200 --warmup 20 --no-cpu-baseline --no-cold-probes --rotate 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d.get('telemetry',{})
print('$wl $lib kern %.4f ms  sclk %s  W %s  sustained %s' % (d['roofline']['kernel_ms'], t.get('sclk_mhz'), t.get('package_power_w'), ['%.4f'%x for x in (t.get('sustained_ms_per_step') or [])]))"
    done
  done
done 2>&1 | tee gpurun_out/ab_r05_6.txt
# the two profiles the evidence session could not summarise (pmc_summary.py did not know the new workload)
KT_STEPS=1000 tools/profile.sh cloud_diag f32 100000000 r05 > gpurun_out/prof_cloud_diag_f32.log 2>&1 || echo "profile cloud_diag f32 FAILED"
KT_STEPS=1000 tools/profile.sh cloud_diag f64 100000000 r05 > gpurun_out/prof_cloud_diag_f64.log 2>&1 || echo "profile cloud_diag f64 FAILED"
ls gpurun_out/profiles | grep cloud_diag
echo finished
