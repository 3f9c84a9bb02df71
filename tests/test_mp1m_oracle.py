"""CPU tests: the 1-moment part of the oracle against the reference's known-answer tests
(tests/golden/mp1m_kats.json), limits / gates asserted by the reference, and an independent numpy statement."""
import json
import math
from pathlib import Path

import numpy as np
import pytest

from cmx import _abi
from cmx import parameters as P

F64 = _abi.F64
G = json.loads((Path(__file__).parent / "golden" / "mp1m_kats.json").read_text())


def _run(oracle, mp, rho, T, q_tot, q_lcl, q_icl, q_rai, q_sno, **kw):
    cols = np.broadcast_arrays(*[np.atleast_1d(np.asarray(v, dtype=np.float64)) for v in (rho, T, q_tot, q_lcl, q_icl, q_rai, q_sno)])
    return oracle.mp1m(F64, mp.c, P.ThermodynamicsParameters("f64"), mp.flags, *[np.ascontiguousarray(c) for c in cols], **kw)


def test_accretion_kats(oracle):
    g = G["accretion"]
    mp = P.Microphysics1MParams("f64")
    q, rho = g["inputs"]["q"], g["inputs"]["rho"]
    cold = _run(oracle, mp, rho, 260.0, 5e-3, q, q, q, q)["sources"]
    warm = _run(oracle, mp, rho, 290.0, 5e-3, q, q, q, q)["sources"]
    e = g["expected"]
    close = lambda x, y: math.isclose(x, y, rel_tol=g["rtol"])  # noqa: E731
    assert close(cold["S_accr_lcl_rai"][0], e["liq_rai"])
    assert close(cold["S_accr_icl_sno"][0], e["ice_sno"])
    assert close(cold["S_accr_lcl_sno_cold"][0], e["liq_sno"]) and cold["S_accr_lcl_sno_warm"][0] == 0
    assert close(warm["S_accr_lcl_sno_warm"][0], e["liq_sno"]) and warm["S_accr_lcl_sno_cold"][0] == 0
    assert close(cold["S_accr_icl_rai"][0], e["ice_rai"])
    assert close(cold["S_accr_freeze_icl_rai"][0], e["rai_sink"])
    # the reference test calls accretion_snow_rain(snow, rain, …) "sno_rai" and (rain, snow, …) "rai_sno"
    # (test/gpu_tests.jl:170-197): cold arm = S_rai_sno = 2.466e-4, warm arm = S_sno_rai = 6.83e-5 (microphysics1M_tests.jl:501-502)
    assert close(cold["S_accr_rai_sno_cold"][0], e["sno_rai"]) and cold["S_accr_rai_sno_warm"][0] == 0
    assert close(warm["S_accr_rai_sno_warm"][0], e["rai_sno"]) and warm["S_accr_rai_sno_cold"][0] == 0
    assert cold["S_accr_melt_lcl_sno"][0] == 0 and cold["S_accr_melt_rai_sno"][0] == 0      # S_melt == 0 when cold
    zero = _run(oracle, mp, rho, 290.0, 0.0, 0.0, 0.0, 0.0, 0.0)["sources"]
    for k in _abi.MP1M_SOURCE_COLUMNS:
        if k.startswith("S_accr") or k.startswith("S_melt") or k.startswith("S_acnv"):
            assert zero[k][0] == 0.0, k


def test_snow_melt_and_cloud_ice_melt(oracle):
    g = G["snow_melt"]
    mp = P.Microphysics1MParams("f64")
    for dT, q_sno, exp in g["cases"]:
        r = _run(oracle, mp, g["rho"], g["T_freeze"] + dT, 0.0, 0.0, 0.0, 0.0, q_sno)["sources"]
        assert math.isclose(r["S_melt_sno_rai"][0], exp, rel_tol=g["rtol"], abs_tol=0.0), (dT, q_sno, r["S_melt_sno_rai"][0])
    # cloud ice melt: zero when cold or without ice (test/microphysics1M_tests.jl:700-747), positive when warm
    tps = P.ThermodynamicsParameters("f64")
    r = _run(oracle, mp, 1.2, tps.T_freeze + 2, 0.0, 0.0, 1e-4, 0.0, 0.0)["sources"]["S_melt_icl_lcl"][0]
    li = (1.2 * 1e-4 * 1e-5 ** 3 / (mp.c.cloud_ice.mass.m0 * 2e7 * math.gamma(4.0))) ** 0.25
    L_f = (tps.LH_s0 - tps.LH_v0) + (tps.cp_l - tps.cp_i) * (tps.T_freeze + 2 - tps.T_0)
    assert math.isclose(r, 4 * math.pi * 2e7 / 1.2 * 0.024 / L_f * 2.0 * li ** 2, rel_tol=1e-13)
    assert _run(oracle, mp, 1.2, tps.T_freeze - 2, 0.0, 0.0, 1e-4, 0.0, 0.0)["sources"]["S_melt_icl_lcl"][0] == 0
    off = P.Microphysics1MParams("f64", cloud_ice_melt=None, snow_melt=None)
    r = _run(oracle, off, 1.2, tps.T_freeze + 2, 0.0, 0.0, 1e-4, 0.0, 1e-4)["sources"]
    assert r["S_melt_icl_lcl"][0] == 0 and r["S_melt_sno_rai"][0] == 0


def test_terminal_velocities(oracle):
    mp = P.Microphysics1MParams("f64")
    chen = P.Chen2022VelTypeRain("f64")
    for key in ("chen2022_rain_velocity_1m", "chen2022_rain_velocity_gpu"):
        g = G[key]
        v = oracle.mp1m_terminal_velocity(F64, mp.c, chen, [g["rho"]], [g["q_rai"]], [0.0])
        assert math.isclose(v["vt_rai_chen"][0], g["expected"], rel_tol=g["rtol"]), (key, v["vt_rai_chen"][0])
    # Blk1M: closed form (CM1:223-249) with the documented defaults; zero for q = 0 (microphysics1M_tests.jl:112)
    rho, q = 1.2, 1e-3
    v = oracle.mp1m_terminal_velocity(F64, mp.c, chen, [rho, rho], [q, 0.0], [q, 0.0])
    v0 = math.sqrt(8 / 3 / 0.55 * (1000 / rho - 1) * 9.81 * 1e-3)
    lam = (rho * q * 1e-9 / (4 / 3 * math.pi * 1000 * 1e-9 * 16e6 * math.gamma(4.0))) ** 0.25
    assert math.isclose(v["vt_rai_blk1m"][0], v0 * (lam / 1e-3) ** 0.5 * math.gamma(4.5) / math.gamma(4.0), rel_tol=1e-13)
    n0s = 4.36e9 * (rho * q) ** 0.63
    lams = (rho * q * 1e-6 / (0.1 * 1e-6 * n0s * math.gamma(3.0))) ** (1 / 3)
    assert math.isclose(v["vt_sno_blk1m"][0], 2 ** 2.25 * 1e-3 ** 0.25 * (lams / 1e-3) ** 0.25 * math.gamma(3.25) / math.gamma(3.0),
                        rel_tol=1e-13)
    assert v["vt_rai_blk1m"][1] == 0 and v["vt_sno_blk1m"][1] == 0 and v["vt_rai_chen"][1] == 0
    # empirical check of the reference (test/microphysics1M_tests.jl:30-47): within 20 % of Smolarkiewicz & Grabowski
    for q_rai in np.linspace(1e-8, 5e-3, 10):
        rr = q_rai / (1 - 20e-3)
        emp = 14.34 * 1.22 ** 0.5 * 1.2 ** -0.3654 * rr ** 0.1346
        got = oracle.mp1m_terminal_velocity(F64, mp.c, chen, [1.2], [q_rai], [0.0])["vt_rai_blk1m"][0]
        assert abs(got - emp) <= 0.2 * emp


def test_autoconversion(oracle):
    mp = P.Microphysics1MParams("f64")
    for key, col, slot in (("kessler_bounds", "S_acnv_lcl_rai", 3), ("snow_acnv_bounds", "S_acnv_icl_sno", 4)):
        g = G[key]
        unit = g["q_threshold"] / g["tau"]
        for frac, target in ((0.5, 0.0), (1.5, 0.5)):
            args = [1.0, 280.0, 0.0, 0.0, 0.0, 0.0, 0.0]
            args[slot] = frac * g["q_threshold"]
            r = _run(oracle, mp, *args)["sources"][col][0]
            assert abs(r - target * unit) <= g["atol_factor"] * unit
    g = G["prescribed_nd"]
    nd = P.Microphysics1MParams("f64", rain_autoconversion=P.PrescribedNd())
    r = _run(oracle, nd, 1.0, 280.0, 0.0, [g["q_lcl"], 0.0, -1e-5], 0.0, 0.0, 0.0)["sources"]["S_acnv_lcl_rai"]
    assert math.isclose(r[0], g["expected"], rel_tol=g["rtol"]) and r[1] == 0 and r[2] == 0
    for x, x0, k, exp, atol in G["logistic_function_integral"]["cases"]:
        got = oracle.logistic_function_integral(F64, x, x0, k)
        assert abs(got - exp) <= atol, (x, x0, k, got)


def test_phase_change_terms_and_gates(oracle, golden):
    mp = P.Microphysics1MParams("f64")
    tps = P.ThermodynamicsParameters("f64")
    # cond/evap KAT shared with the 2M path (test/gpu_tests.jl:606)
    e = golden["condevap"][0]["inputs"]
    r = _run(oracle, mp, e["rho"], e["T"], e["q_tot"], 0.0, 0.0, 0.0, 0.0)["sources"]
    assert math.isclose(r["S_phase_change_vap_lcl"][0], golden["condevap"][0]["expected"], rel_tol=1e-14)
    # _conv_q_vap_to_q_icl_const KAT: test/microphysics_noneq_tests.jl:40-88: ρ = 0.8, T = 263, q_tot = 1.2 q_sat(liq)
    from scipy.optimize import brentq  # noqa: F401  (not needed: q_sat closed form)
    dcl = tps.cp_v - tps.cp_l
    T, rho = 263.0, 0.8
    p_sat = tps.press_triple * (T / tps.T_triple) ** (dcl / tps.R_v) * math.exp((tps.LH_v0 - dcl * tps.T_0) / tps.R_v * (1 / tps.T_triple - 1 / T))
    q_sat_liq = p_sat / (rho * tps.R_v * T)
    r = _run(oracle, mp, rho, T, 1.2 * q_sat_liq, 0.0, 0.0, 0.0, 0.0)["sources"]
    assert math.isclose(r["S_phase_change_vap_lcl"][0], 3.763045798130144e-5, rel_tol=1e-12)
    dci = tps.cp_v - tps.cp_i
    p_sat_i = tps.press_triple * (T / tps.T_triple) ** (dci / tps.R_v) * math.exp((tps.LH_s0 - dci * tps.T_0) / tps.R_v * (1 / tps.T_triple - 1 / T))
    ri = _run(oracle, mp, rho, T, 1.2 * p_sat_i / (rho * tps.R_v * T), 0.0, 0.0, 0.0, 0.0)["sources"]
    assert math.isclose(ri["S_phase_change_vap_icl"][0], 3.235984203087906e-5, rel_tol=1e-12)
    assert r["S_phase_change_vap_lcl"][0] < r["S_phase_change_vap_icl"][0]     # ice grows faster (noneq tests :90-92)
    # INP limiter: no deposition above freezing (NonEq:56-58)
    assert _run(oracle, mp, 1.0, 274.0, 2e-2, 0.0, 0.0, 0.0, 0.0)["sources"]["S_phase_change_vap_icl"][0] == 0
    # rain evaporation only when sub-saturated, ≤ 0 (test/microphysics1M_tests.jl:560-600)
    sub = _run(oracle, mp, 1.2, 288.0, 2e-3, 0.0, 0.0, 1e-4, 0.0)["sources"]["S_phase_change_vap_rai"][0]
    sup = _run(oracle, mp, 1.2, 288.0, 3e-2, 0.0, 0.0, 1e-4, 0.0)["sources"]["S_phase_change_vap_rai"][0]
    assert sub < 0 and sup == 0
    # snow: deposition allowed by default, clipped with SublimationOnly (CM1:979-999)
    args = (1.0, 250.0, 2e-3, 0.0, 0.0, 0.0, 1e-4)
    dep = _run(oracle, mp, *args)["sources"]["S_phase_change_vap_sno"][0]
    only = _run(oracle, P.Microphysics1MParams("f64", snow_deposition_sublimation=P.SublimationOnly()), *args)["sources"]
    assert dep > 0 and only["S_phase_change_vap_sno"][0] == 0


def test_tendencies_are_the_aggregate_of_sources(oracle):
    """_aggregate_tendencies (BMT:227-252) re-stated in numpy + conservation: Σ of the four tendencies equals the net
    vapour exchange (collisions / melting / autoconversion only move mass between hydrometeors)."""
    rng = np.random.default_rng(5)
    n = 20000
    rho = rng.uniform(0.3, 1.3, n)
    T = rng.uniform(230, 300, n)
    q = lambda: np.where(rng.random(n) < 0.3, 0.0, 10 ** rng.uniform(-8, -2.5, n))  # noqa: E731
    q_lcl, q_icl, q_rai, q_sno = q(), q(), q(), q()
    q_tot = q_lcl + q_icl + q_rai + q_sno + 10 ** rng.uniform(-5, -1.8, n)
    for opts in ({}, {"snow_autoconversion": P.WithSupersaturation(), "snow_deposition_sublimation": P.SublimationOnly(),
                      "rain_autoconversion": P.PrescribedNd()}, {"rain_snow_accretion": None, "cloud_ice_melt": None}):
        mp = P.Microphysics1MParams("f64", **opts)
        r = _run(oracle, mp, rho, T, q_tot, q_lcl, q_icl, q_rai, q_sno)
        s = r["sources"]
        dq_lcl = (s["S_phase_change_vap_lcl"] - s["S_acnv_lcl_rai"] - s["S_accr_lcl_rai"] - s["S_accr_lcl_sno_cold"]
                  - s["S_accr_lcl_sno_warm"] + s["S_melt_icl_lcl"])
        np.testing.assert_array_equal(r["dq_lcl_dt"], dq_lcl)
        total = r["dq_lcl_dt"] + r["dq_icl_dt"] + r["dq_rai_dt"] + r["dq_sno_dt"]
        vap = s["S_phase_change_vap_lcl"] + s["S_phase_change_vap_icl"] + s["S_phase_change_vap_rai"] + s["S_phase_change_vap_sno"]
        scale = sum(r["scale"].values())
        assert np.all(np.abs(total - vap) <= 1e-12 * scale + 1e-300)
        for k in ("dq_lcl_dt", "dq_icl_dt", "dq_rai_dt", "dq_sno_dt"):
            assert np.all(np.isfinite(r[k])), k
        if "rain_snow_accretion" in opts:
            assert np.all(s["S_accr_rai_sno_cold"] == 0) and np.all(s["S_melt_icl_lcl"] == 0)


def test_options_and_parameter_layout():
    assert P.Microphysics1MOptions().flags == _abi.CMX_1M_DEFAULT_OPTIONS
    o = P.Microphysics1MOptions(cloud_ice_melt=None, rain_autoconversion=P.PrescribedNd)
    assert not (o.flags & _abi.CMX_1M_CLOUD_ICE_MELT) and (o.flags & _abi.CMX_1M_RAIN_ACNV_PRESCRIBED_ND)
    with pytest.raises(TypeError):
        P.Microphysics1MOptions(snow_melt=P.Kessler1M())
    with pytest.raises(TypeError):
        P.Microphysics1MOptions(not_a_field=None)
    mp = P.Microphysics1MParams("f64")
    assert math.isclose(mp.c.vel_snow.v0, 2 ** 2.25 * 1e-3 ** 0.25) and math.isclose(mp.c.rain.mass.m0, 4 / 3 * math.pi * 1e-6)
    assert math.isclose(mp.c.vel_rain.gamma_accr_rain_sink, math.gamma(6.5)) and mp.c.snow.mass.gamma_coeff == 2.0


def test_float32_arithmetic_oracle_tracks_float64_1m(oracle):
    import parity
    from cmx import synthetic
    st = synthetic.mp1m_state(200000, seed=21)
    cols = [c.numpy() for c in st]
    r64 = oracle.mp1m(F64, P.Microphysics1MParams("f64").c, P.ThermodynamicsParameters("f64"), _abi.CMX_1M_DEFAULT_OPTIONS,
                      *[c.astype(np.float64) for c in cols], float32_gates=True, nthreads=4, want_sources=False)
    r32 = oracle.mp1m(_abi.F32, P.Microphysics1MParams("f32").c, P.ThermodynamicsParameters("f32"), _abi.CMX_1M_DEFAULT_OPTIONS,
                      *cols, nthreads=4, want_sources=False)
    near = np.abs(cols[1].astype(np.float64) - 273.15) < 1e-4
    r64["near_branch"] = near
    parity.assert_parity(r32, r64, 1e-3, names=["dq_lcl_dt", "dq_icl_dt", "dq_rai_dt", "dq_sno_dt"], what="1M f32 oracle")


def test_chen2022_sedimentation_velocity_kats(oracle):
    """test/gpu_tests.jl:608-630: the four bulk fall speeds (cloud liquid Stokes, cloud ice / rain / snow Chen-2022)."""
    g = G["chen2022_sedimentation_velocities"]
    mp = P.Microphysics1MParams("f64")
    r = oracle.sedimentation_velocities(F64, mp.c, P.StokesRegimeVelType("f64"), P.Chen2022VelTypeRain("f64"), P.Chen2022VelTypeIce("f64"),
                                        [g["rho"]], [g["q_lcl"]], [g["q_icl"]], [g["q_rai"]], [g["q_sno"]])
    for k in ("w_lcl", "w_icl", "w_rai", "w_sno"):
        assert math.isclose(r[k][0], g[k], rel_tol=g["rtol"]), k
    e = g["extra"]      # test/microphysics1M_tests.jl:56-78
    x = oracle.sedimentation_velocities(F64, mp.c, P.StokesRegimeVelType("f64"), P.Chen2022VelTypeRain("f64"), P.Chen2022VelTypeIce("f64"),
                                        [e["snow"]["rho"], e["rain"]["rho"]], [0.0, 0.0], [0.0, 0.0], [0.0, e["rain"]["q_rai"]],
                                        [e["snow"]["q_sno"], 0.0])
    assert math.isclose(x["w_sno"][0], e["snow"]["w_sno"], rel_tol=1e-14) and math.isclose(x["w_rai"][1], e["rain"]["w_rai"], rel_tol=1e-14)
    z = oracle.sedimentation_velocities(F64, mp.c, P.StokesRegimeVelType("f64"), P.Chen2022VelTypeRain("f64"), P.Chen2022VelTypeIce("f64"),
                                        [1.0, 1.0], [0.0, -1e-9], [0.0, -1e-9], [0.0, -1e-9], [0.0, -1e-9])
    assert all(np.all(v == 0) for v in z.values())
