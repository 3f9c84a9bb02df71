"""CloudDiagnostics over columns (round 5 — the diagnostics a host model computes from the same state columns as the tendencies):
    CMD.radar_reflectivity_1M, radar_reflectivity_2M, effective_radius_2M, effective_radius_Liu_Hallet_97      src/CloudDiagnostics.jl:31-163
CPU: the oracle restatement (oracle/cmx_oracle_diag_impl.h) against the reference's own known answers (test/cloud_diagnostics.jl, committed as
tests/golden/cloud_diagnostics_kats.json) in Float64 and Float32 arithmetic; GPU: the device entry cmx_cloud_diagnostics_* against those KATs and,
on random states, against the oracle."""
import json
from pathlib import Path

import numpy as np
import pytest

torch = pytest.importorskip("torch")
from cmx import _abi  # noqa: E402
from cmx import parameters as P  # noqa: E402

G = json.loads((Path(__file__).parent / "golden" / "cloud_diagnostics_kats.json").read_text())
NPF = {"f32": np.float32, "f64": np.float64}
DT = {"f32": torch.float32, "f64": torch.float64}


def _sb(ft, limited):
    return P.SB2006(P.create_toml_dict(ft, P.SB2006_LIMITERS_OVERRIDE), limited)


@pytest.mark.parametrize("ft", ["f64", "f32"])
def test_oracle_reproduces_the_reference_kats(oracle, ft):
    fam = _abi.family(ft)
    f32 = ft == "f32"
    g = G["radar_reflectivity_1M"]
    rain = P.Microphysics1MParams(ft).c.rain      # CMP.Rain(FT) = the `rain` member of Microphysics1MParams
    Z = oracle.cloud_diagnostics(fam, np.full(2, g["rho"]), np.zeros(2), g["q_rai"], rain=rain, float32_gates=f32, want=("Z_1m",))["Z_1m"]
    assert np.all(np.abs(Z - g["expected_dBZ"]) <= g["atol"])
    g = G["sb2006_2M"]
    for limited in (True, False):
        sb = _sb(ft, limited)
        out = oracle.cloud_diagnostics(fam, np.full(4, g["rho"]), g["q_lcl"], g["q_rai"], g["N_lcl"], g["N_rai"], pdf_c=sb.pdf_c, pdf_r=sb.pdf_r,
                                       limited=limited, float32_gates=f32, want=("Z_2m", "reff_2m"))
        assert np.all(np.abs(out["Z_2m"] - g["radar_reflectivity_dBZ"]) <= (g["atol_Z"] if ft == "f64" else 2e-3)), (limited, out["Z_2m"])
        assert np.all(np.abs(out["reff_2m"] - g["effective_radius_m"]) <= g["atol_reff"]), (limited, out["reff_2m"])
        s = G["sb2006_2M_small_numbers"]
        out = oracle.cloud_diagnostics(fam, [s["rho"]], [s["q_lcl"]], [s["q_rai"]], [s["N_lcl"]], [s["N_rai"]], pdf_c=sb.pdf_c, pdf_r=sb.pdf_r, limited=limited,
                                       float32_gates=f32, want=("Z_2m", "reff_2m"))
        assert abs(out["Z_2m"][0] - s["radar_reflectivity_dBZ"]) <= s["atol_Z"] and abs(out["reff_2m"][0] - s["effective_radius_m"]) <= s["atol_reff"]
    g = G["liu_hallett_97"]
    r = oracle.cloud_diagnostics(fam, [g["rho"]], [g["q_lcl"]], [g["q_rai"]], [g["N_lcl"]], [g["N_rai"]], rho_w=g["rho_w"], float32_gates=f32, want=("reff_lh97",))["reff_lh97"]
    assert abs(r[0] - g["expected_m"]) <= g["atol"]
    # the three-argument method = N_lcl 100, no rain (test/cloud_diagnostics.jl:118-126)
    a = oracle.cloud_diagnostics(fam, [g["rho"]], [g["q_lcl"]], rho_w=g["rho_w"], float32_gates=f32, want=("reff_lh97",))["reff_lh97"]
    b = oracle.cloud_diagnostics(fam, [g["rho"]], [g["q_lcl"]], [g["default_q_rai"]], [g["default_N_lcl"]], [g["default_N_rai"]], rho_w=g["rho_w"], float32_gates=f32,
                                 want=("reff_lh97",))["reff_lh97"]
    assert a[0] == b[0]
    c = G["effective_radius_const"]
    mp = P.Microphysics1MParams(ft)
    assert mp.c.cloud_liquid.r_eff == NPF[ft](c["cloud_liquid_m"]) and mp.c.cloud_ice.r_eff == NPF[ft](c["cloud_ice_m"])
