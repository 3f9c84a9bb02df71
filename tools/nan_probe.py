"""Which outputs of the 2M warm-rain entry are NaN when exactly one input column holds a NaN (documentation probe)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "cloudmicrophysics.jl_amd"))
import cmx  # noqa: E402
from cmx import parameters as P, synthetic  # noqa: E402

dev = torch.device("cuda", 0)
names = ("rho", "T", "q_tot", "q_lcl", "n_lcl", "q_rai", "n_rai")
for sfx, dt in (("f32", torch.float32), ("f64", torch.float64)):
    mp, tps = P.Microphysics2MParams(sfx), P.ThermodynamicsParameters(sfx)
    st = [c.clone() for c in synthetic.sb2006_state(4096, dtype=dt, device=dev, seed=3)]
    for k, nm in enumerate(names):
        cols = [c.clone() for c in st]
        cols[k][:] = float("nan")
        out = cmx.bulk_microphysics_tendencies(cmx.Microphysics2Moment(), mp, tps, *cols)
        frac = [float(torch.isnan(o).float().mean()) for o in out[:4]]
        print(sfx, f"NaN in {nm:6s} -> NaN fraction of (dq_lcl, dn_lcl, dq_rai, dn_rai):", [round(f, 3) for f in frac])
