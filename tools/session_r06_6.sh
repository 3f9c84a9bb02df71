#!/bin/bash
# Round 6, GPU session 6: the fall-speed / melting kernel deals a wave's non-empty (state, segment) pairs out 64 per round (libcmx) against one lane per state
# (p3seg0) and round 5's final tree (r05).  Bit-identity of the two schedules first, then the P3 parity suites, then the A/B.
#   libcmx.so          make -C cloudmicrophysics.jl_amd/csrc
#   libcmx_p3seg0.so   tools/build_variant.sh p3seg0 -DCMX_P3_COMPACT_SEGMENTS=0
#   libcmx_r05.so      tools/build_ref_variant.sh r05 c85d362
set -u
L=cloudmicrophysics.jl_amd/csrc
python - <<'PY'
import subprocess, sys, os, json
code = r"""
import sys, torch
sys.path.insert(0, 'cloudmicrophysics.jl_amd')
import cmx
from cmx import parameters as P, synthetic
out = {}
for ft, dt in (('f64', torch.float64), ('f32', torch.float32)):
    n = 300_007
    st = synthetic.p3_state(n, dtype=dt, device='cuda', seed=5); rho_a = synthetic.p3_air_density(n, dtype=dt, device='cuda', seed=6)
    p, vel = P.ParametersP3(ft), P.Chen2022VelTypeIce(ft)
    r = cmx.p3_shape_and_terminal_velocities(p, vel, rho_a, *st)
    ll = cmx.p3_shape(p, *st, want=('log_lambda',)).log_lambda
    v = cmx.p3_terminal_velocities(p, vel, rho_a, *st, ll)
    T = 268.0 + 10 * torch.rand(n, dtype=dt, device='cuda')
    m = cmx.p3_ice_melt(p, vel, P.AirProperties(ft), P.ThermodynamicsParameters(ft), P.VentilationFactorP3(ft), T, rho_a, *st, torch.where(torch.isfinite(ll), ll, torch.zeros_like(ll)))
    torch.save([x.cpu() for x in (r.v_n, r.v_m, r.log_lambda, r.D_m, v.v_n, v.v_m, m[0], m[1])], sys.argv[1] + ft + '.pt')
"""
for tag in ("", "_p3seg0"):
    env = dict(os.environ, CMX_LIB=os.path.abspath(f"cloudmicrophysics.jl_amd/csrc/libcmx{tag}.so"))
    subprocess.run([sys.executable, "-c", code, f"/tmp/out{tag}_"], check=True, env=env)
import torch
for ft in ("f64", "f32"):
    a, b = torch.load(f"/tmp/out_{ft}.pt"), torch.load(f"/tmp/out_p3seg0_{ft}.pt")
    same = [bool(torch.equal(x, y) or torch.equal(torch.nan_to_num(x, nan=-1.0), torch.nan_to_num(y, nan=-1.0))) for x, y in zip(a, b)]
    print(ft, "bit-identical to the one-lane-per-state schedule (v_n, v_m, log λ, D_m fused; v_n, v_m split; melt dN, dL):", same)
    assert all(same)
PY
timeout 1500 python -m pytest tests/test_p3_gpu.py tests/test_mp2m_p3_gpu.py tests/test_reference_suites_gpu.py tests/test_nan_inputs_gpu.py -q -m gpu --tb=short 2>&1 | grep -E "Error|error|assert|passed|failed|FAILED|^E " | head -20
EXTRA="--no-cold-probes --rotate 1 --no-telemetry" REPS=3 STEPS=20 tools/ab_bench.sh "p3:f64 p3:f32 p3_split:f64" $L/libcmx_r05.so $L/libcmx_p3seg0.so $L/libcmx.so 2>&1 | tee gpurun_out/ab_r06_6.txt
echo finished
