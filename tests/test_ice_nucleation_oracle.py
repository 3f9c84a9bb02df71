"""CPU tests: the ice-nucleation part of the oracle against the reference's known-answer tests
(tests/golden/ice_nucleation_kats.json) and an independent numpy statement of the formulas."""
import json
import math
from pathlib import Path

import numpy as np
import pytest

from cmx import _abi
from cmx import parameters as P

F64 = _abi.F64
G = json.loads((Path(__file__).parent / "golden" / "ice_nucleation_kats.json").read_text())


def _solve_a_w(oracle, tps, T, delta):
    ice, _ = oracle.water_activity(F64, tps, np.array([T]))
    return ice[0] + delta


def test_water_activity_kats(oracle):
    tps = P.ThermodynamicsParameters("f64")
    for e in G["a_w_ice"]:
        ice, _ = oracle.water_activity(F64, tps, np.array([e["T"]]))
        assert math.isclose(ice[0], e["expected"], rel_tol=e["rtol"])
    for e in G["a_w_eT"]:
        _, eT = oracle.water_activity(F64, tps, np.array([e["T"]]), np.array([e["e"]]))
        assert math.isclose(eT[0], e["expected"], rel_tol=e["rtol"])


def test_abifm_and_koop_kats(oracle):
    tps, koop = P.ThermodynamicsParameters("f64"), P.Koop2000("f64")
    T = 220.0
    for e in G["ABIFM_J"]:
        dust = getattr(P, e["dust"])("f64")
        a_w = _solve_a_w(oracle, tps, T, e["delta_a_w"])
        r = oracle.ice_nucleation_rates(F64, tps, dust, koop, _abi.CMX_ICENUC_HOM_LINEAR, [T], [a_w], [1e-6])
        assert math.isclose(r["delta_a_w"][0], e["delta_a_w"], rel_tol=1e-13)
        assert math.isclose(r["J_het"][0], e["expected"], rel_tol=1e-10), (e, r["J_het"][0])
        assert math.isclose(r["rate_het"][0], r["J_het"][0] * 4 * math.pi * 1e-12, rel_tol=1e-14)
    h = G["homogeneous_J"]
    a_w = _solve_a_w(oracle, tps, T, h["delta_a_w"])
    dust = P.Kaolinite("f64")
    rc = oracle.ice_nucleation_rates(F64, tps, dust, koop, 0, [T], [a_w], [2e-6])
    rl = oracle.ice_nucleation_rates(F64, tps, dust, koop, _abi.CMX_ICENUC_HOM_LINEAR, [T], [a_w], [2e-6])
    assert math.isclose(rc["J_hom"][0], h["J_cubic"], rel_tol=1e-9), rc["J_hom"][0]
    assert math.isclose(rl["J_hom"][0], h["J_linear"], rel_tol=2e-7), rl["J_hom"][0]   # coefficients printed to 9 digits
    assert math.isclose(rc["rate_hom"][0], rc["J_hom"][0] * 4 / 3 * math.pi * 8e-18, rel_tol=1e-14)
    assert rc["n_domain_errors"] == 0


def test_cubic_domain_error_becomes_nan_and_count(oracle):
    tps, koop, dust = P.ThermodynamicsParameters("f64"), P.Koop2000("f64"), P.Illite("f64")
    d = G["homogeneous_J_cubic_domain"]
    T = np.full(4, 225.0)
    deltas = np.array([d["too_small"], d["too_large"], koop.delta_a_w_min, koop.delta_a_w_max])
    ice, _ = oracle.water_activity(F64, tps, T)
    # evaluate with Δa_w exactly at the probes: feed a_w = a_w_ice + Δ and accept the rounding of the sum by
    # nudging the two in-range end points inwards by one ulp of a_w
    a_w = ice + deltas
    a_w[2] = np.nextafter(a_w[2], 1.0)
    a_w[3] = np.nextafter(a_w[3], 0.0)
    r = oracle.ice_nucleation_rates(F64, tps, dust, koop, 0, T, a_w, np.full(4, 1e-6))
    assert np.isnan(r["J_hom"][:2]).all() and np.isnan(r["rate_hom"][:2]).all()
    assert np.isfinite(r["J_hom"][2:]).all() and r["n_domain_errors"] == 2
    assert np.isfinite(r["J_het"]).all()            # ABIFM has no domain restriction
    rl = oracle.ice_nucleation_rates(F64, tps, dust, koop, _abi.CMX_ICENUC_HOM_LINEAR, T, a_w, np.full(4, 1e-6))
    assert np.isfinite(rl["J_hom"]).all() and rl["n_domain_errors"] == 0


def test_against_numpy_restatement(oracle):
    rng = np.random.default_rng(3)
    n = 5000
    tps, koop, dust = P.ThermodynamicsParameters("f64"), P.Koop2000("f64"), P.Kaolinite("f64")
    T = rng.uniform(190, 240, n)
    dcl, dci = tps.cp_v - tps.cp_l, tps.cp_v - tps.cp_i
    ps = lambda LH, dcp: tps.press_triple * (T / tps.T_triple) ** (dcp / tps.R_v) * np.exp(  # noqa: E731
        (LH - dcp * tps.T_0) / tps.R_v * (1 / tps.T_triple - 1 / T))
    a_ice = ps(tps.LH_s0, dci) / ps(tps.LH_v0, dcl)
    delta = rng.uniform(0.2, 0.4, n)
    r = 10 ** rng.uniform(-8, -5, n)
    out = oracle.ice_nucleation_rates(F64, tps, dust, koop, 0, T, a_ice + delta, r)
    d = (a_ice + delta) - a_ice
    np.testing.assert_allclose(out["delta_a_w"], d, rtol=0, atol=1e-15)
    np.testing.assert_allclose(out["J_het"], 10 ** (dust.ABIFM_m * d + dust.ABIFM_c + 4), rtol=1e-12)
    ok = (d >= koop.delta_a_w_min) & (d <= koop.delta_a_w_max)
    J = 10 ** (koop.c1 + koop.c2 * d - koop.c3 * d ** 2 + koop.c4 * d ** 3 + 6)
    np.testing.assert_allclose(out["J_hom"][ok], J[ok], rtol=1e-11)
    assert np.isnan(out["J_hom"][~ok]).all() and out["n_domain_errors"] == int((~ok).sum())
    # monotonicity asserted by the reference (test/homogeneous_ice_nucleation_tests.jl:26-35): colder → larger J
    assert np.all(np.diff(J[ok][np.argsort(d[ok])]) > 0)


def test_homogeneous_J_linear_coefficients_are_the_references_literals():
    """Both Linear_J_hom coefficients are literals in the reference tree (papers/ice_nucleation_2024/calibration_setup.jl:149,
    calibration.jl:271-272); the parameter table must hold exactly those, and they must reproduce the reference's combined KAT."""
    g = G["homogeneous_J_linear_coefficients"]
    k = P.Koop2000("f64")
    assert k.linear_c2 == g["linear_c2_slope"] and k.linear_c1 == g["linear_c1_intercept"]
    J = 10.0 ** (g["linear_c2_slope"] * g["kat"]["delta_a_w"] + g["linear_c1_intercept"]) * 1e6      # [cm⁻³ s⁻¹] → [m⁻³ s⁻¹]
    assert math.isclose(J, g["kat"]["J_linear"], rel_tol=g["kat"]["rtol"])       # 9.3e-8: the residual stated in the fixture's note
