#!/bin/bash
# Round 4, GPU session 5: the 2M + P3 one-launch form with its raw inputs in LDS (tests + FETCH/WRITE traffic), and a same-box A/B of the
# whole library compiled with -mllvm -amdgpu-sched-strategy=max-ilp (libcmx_ilp.so) against the default scheduler.
set -u
mkdir -p gpurun_out/bench gpurun_out/profiles
timeout 1200 python -m pytest tests/test_mp2m_p3_gpu.py tests/test_p3_collisions_gpu.py tests/test_lean_math.py tests/test_layouts_gpu.py -q -m gpu 2>&1 | tail -5
prof() { KT_STEPS=${KT_STEPS:-10} tools/profile.sh "$1" "$2" "$3" r04c "${4:-}" > gpurun_out/prof_${1}_${2}.log 2>&1 || echo "profile $1 $2 FAILED"; }
prof mp2m_p3 f64 1000000
prof mp2m_p3 f32 1000000
python - <<'PY'
import json
for dt in ('f64', 'f32'):
    d = json.load(open(f'gpurun_out/profiles/r04c_pmc_traffic_mp2m_p3_{dt}.json'))
    print('mp2m_p3', dt, 'fetch', d['fetch_bytes_corrected'], 'write', d['write_bytes'], 'ratio', d['traffic_over_algorithmic'], 'avg ms', d['avg_ns'] * 1e-6)
PY
L=cloudmicrophysics.jl_amd/csrc
EXTRA="--no-cold-probes --rotate 1" REPS=2 STEPS=30 tools/ab_bench.sh "sb2006:f64 mp1m:f64 arg2000:f64 sb2006_column:f64 mp1m_column:f64 mp1m_lin:f64 sb2006_fields:f64 icenuc:f64 arg2000_columns:f64 sb2006:f32 sb2006_chen:f32 sb2006_column:f32 sb2006_fields:f32 mp1m:f32 arg2000:f32 mp1m_column:f32 mp1m_lin:f32 icenuc:f32" $L/libcmx.so $L/libcmx_ilp.so 2>&1 | tee gpurun_out/ab_r04_5.txt
EXTRA="--no-cold-probes --rotate 1 --points 10000000" REPS=1 STEPS=5 tools/ab_bench.sh "p3:f64 p3:f32" $L/libcmx.so $L/libcmx_ilp.so 2>&1 | tee -a gpurun_out/ab_r04_5.txt
EXTRA="--no-cold-probes --rotate 1 --points 1000000" REPS=1 STEPS=3 tools/ab_bench.sh "mp2m_p3:f64 mp2m_p3:f32 p3_selfcol:f64 p3_selfcol:f32" $L/libcmx.so $L/libcmx_ilp.so 2>&1 | tee -a gpurun_out/ab_r04_5.txt
