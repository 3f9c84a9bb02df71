#!/usr/bin/env python3
"""One-off GPU probe (round 4): what the SB2006 and 1-moment entries return for ρ = 0 and ρ < 0 (clamped to 0), both float types,
next to the oracle — the finite / NaN / ±Inf pattern per output.  Output → stdout (tools/session_r04_*.sh redirects it)."""
import sys
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parent.parent
for p in (REPO / "cloudmicrophysics.jl_amd", REPO / "oracle", REPO / "tests"):
    sys.path.insert(0, str(p))
import cmx  # noqa: E402
import oracle_binding as ob  # noqa: E402
from cmx import _abi  # noqa: E402
from cmx import parameters as P  # noqa: E402


def cls(a):
    a = np.asarray(a, dtype=np.float64)
    return "".join("n" if np.isnan(v) else ("+" if v == np.inf else ("-" if v == -np.inf else ("0" if v == 0 else "f"))) for v in a)


dev = torch.device("cuda:0")
rows2 = [(rho, T, qt, ql, nl, qr, nr) for rho in (0.0, -1.0) for (T, qt, ql, nl, qr, nr) in
         ((290.0, 7e-3, 1e-3, 1e8, 5e-3, 1e4), (290.0, 7e-3, 0.0, 0.0, 0.0, 0.0), (250.0, 1e-4, 1e-3, 1e8, 0.0, 0.0), (300.0, 3e-2, 0.0, 0.0, 2e-3, 5e3))]
arr2 = np.array(rows2).T
rows1 = [(rho, T, qt, ql, qi, qr, qs) for rho in (0.0, -1.0) for (T, qt, ql, qi, qr, qs) in
         ((290.0, 1.5e-2, 1e-3, 0.0, 5e-3, 0.0), (260.0, 3e-3, 1e-4, 2e-4, 1e-4, 3e-3), (275.0, 4e-3, 0.0, 1e-4, 0.0, 1e-3), (240.0, 3e-4, 0.0, 0.0, 0.0, 0.0))]
arr1 = np.array(rows1).T
for ft, dt in (("f32", torch.float32), ("f64", torch.float64)):
    cols = [torch.tensor(a, dtype=dt, device=dev) for a in arr2]
    got = cmx.bulk_microphysics_tendencies(cmx.Microphysics2Moment(), P.Microphysics2MParams(ft), P.ThermodynamicsParameters(ft), *cols, vel=cmx.SB2006VelType)
    ref = ob.sb2006_warm_rain_tendencies(_abi.F64, P.WarmRainParams2M("f64").c, P.ThermodynamicsParameters("f64"), P.rain_vel_params("f64"),
                                         _abi.CMX_SB2006_LIMITED | _abi.CMX_VEL_SB2006, *arr2, float32_gates=(ft == "f32"), nthreads=1)
    for k, v in got._asdict().items():
        print(f"SB2006 {ft} {k:10s} device {cls(v.cpu().numpy())}  oracle {cls(ref[k])}")
    cols = [torch.tensor(a, dtype=dt, device=dev) for a in arr1]
    mp = P.Microphysics1MParams(ft)
    got = cmx.bulk_microphysics_tendencies_1m(cmx.Instantaneous(), cmx.Microphysics1Moment(), mp, P.ThermodynamicsParameters(ft), *cols)
    mp64 = P.Microphysics1MParams("f64")
    ref = ob.mp1m(_abi.F64, mp64.c, P.ThermodynamicsParameters("f64"), mp64.flags, *arr1, nthreads=1, float32_gates=(ft == "f32"))
    for k, v in got._asdict().items():
        print(f"1M     {ft} {k:10s} device {cls(v.cpu().numpy())}  oracle {cls(ref[k])}")
