"""Which outputs of the 2M warm-rain entry are NaN when exactly one input column holds a NaN (documentation probe, HISTORY.md §5, "NaN inputs")."""
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


def probe(title, call, cols, names):
    for k, nm in enumerate(names):
        bad = [c.clone() for c in cols]
        bad[k][:] = float("nan")
        outs = [o for o in call(bad) if o is not None]
        frac = [round(float(torch.isnan(o).float().mean()), 3) for o in outs]
        print(f"{title}: NaN in {nm:10s} -> NaN fraction per output column {frac}")


for sfx, dt in (("f32", torch.float32),):
    tps = P.ThermodynamicsParameters(sfx)
    sta = [c.clone() for c in synthetic.arg_state(4096, dtype=dt, device=dev, seed=4)]
    ap, aip, ad = P.AerosolActivationParameters(sfx), P.AirProperties(sfx), synthetic.arg_config3_distribution()
    probe("ARG2000", lambda c: cmx.aerosol_activation(ap, ad, aip, tps, *c, want=("N_act", "S_max")).N_act[:2] + (cmx.aerosol_activation(ap, ad, aip, tps, *c, want=("S_max",)).S_max,),
          sta, ("T", "p", "w", "q_tot"))
    sti = [c.clone() for c in synthetic.ice_nucleation_state(4096, dtype=dt, device=dev, seed=5)]
    dust, koop = P.Kaolinite(sfx), P.Koop2000(sfx)
    probe("icenuc", lambda c: tuple(cmx.ice_nucleation_rates(tps, dust, koop, *c))[:5], sti, ("T", "a_w", "r"))
    p3 = [c.clone() for c in synthetic.p3_state(4096, dtype=dt, device=dev, seed=6)]
    probe("P3 shape", lambda c: tuple(cmx.p3_shape(P.ParametersP3(sfx), *c)), p3, ("rho_q_ice", "rho_n_ice", "rho_q_rim", "rho_b_rim"))
    st2 = [c.clone() for c in synthetic.sb2006_state(4096, dtype=dt, device=dev, seed=3)]
    mp2 = P.Microphysics2MParams(sfx)
    q_tot, q_lcl, n_lcl, q_rai, n_rai = st2[2], st2[3], st2[4], st2[5], st2[6]
    rho, T = st2[0], st2[1]
    probe("SB2006 process rates", lambda c: tuple(cmx.sb2006_process_rates(mp2, tps, *c))[:6],
          [q_tot, q_lcl, q_rai, n_lcl * rho, n_rai * rho, rho, T], ("q_tot", "q_lcl", "q_rai", "N_lcl", "N_rai", "rho", "T"))
