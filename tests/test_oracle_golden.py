"""CPU tests: the oracle (oracle/) against the reference's known-answer tests (tests/golden) and
against an independent numpy re-statement of each SB2006 formula — the same two styles the
reference's own suite uses (absolute KATs: test/gpu_tests.jl; inline re-derivation at rtol 1e-6:
test/microphysics2M_tests.jl:194-565)."""
import math

import numpy as np
import pytest
from scipy import special as sp

from cmx import _abi
from cmx import parameters as P

F64 = _abi.F64


def _a(v):
    return np.array([v], dtype=np.float64)


def _flags(limited, vel=_abi.CMX_VEL_SB2006):
    return (_abi.CMX_SB2006_LIMITED if limited else 0) | vel


def _close(got, e):
    if "rtol" in e:
        return math.isclose(got, e["expected"], rel_tol=e["rtol"], abs_tol=0.0)
    return abs(got - e["expected"]) <= e["atol"]


@pytest.mark.parametrize("limited", [True, False])
def test_process_rates_kats(oracle, golden, limited):
    g = golden["process_rates_default_params"]
    i = g["inputs"]
    wr = P.WarmRainParams2M("f64", is_limited=limited)
    r = oracle.sb2006_process_rates(F64, wr.c, P.ThermodynamicsParameters("f64"), P.rain_vel_params("f64"),
                                    _flags(limited), _a(i["q_tot"]), _a(i["q_lcl"]), _a(i["q_rai"]), _a(i["N_lcl"]),
                                    _a(i["N_rai"]), _a(i["rho"]), _a(i["T"]))
    for e in g["common"] + g["limited" if limited else "notlimited"]:
        assert _close(r[e["col"]][0], e), (e, r[e["col"]][0])
    assert r["accr_dq_lcl_dt"][0] == -r["accr_dq_rai_dt"][0]


def test_condevap_and_thermo_kats(oracle, golden):
    tps = P.ThermodynamicsParameters("f64")
    wr = P.WarmRainParams2M("f64")
    for e in golden["condevap"]:
        i = e["inputs"]
        r = oracle.sb2006_process_rates(F64, wr.c, tps, None, _flags(True, 0), _a(i["q_tot"]), _a(i["q_lcl"]),
                                        _a(i["q_rai"]), _a(0.0), _a(0.0), _a(i["rho"]), _a(i["T"]))
        assert math.isclose(r["condevap"][0], e["expected"], rel_tol=e["rtol"])
    for e in golden["thermo"]:
        if e["quantity"] == "psat_ice_over_liquid":
            got = oracle.psat_ice(F64, tps, e["T"]) / oracle.psat_liquid(F64, tps, e["T"])
        else:
            got = e["e"] / oracle.psat_liquid(F64, tps, e["T"])
        assert math.isclose(got, e["expected"], rel_tol=e["rtol"]), (e, got)


@pytest.mark.parametrize("limited", [True, False])
def test_chen2022_rain_velocity_kat(oracle, golden, limited):
    g = golden["chen2022_rain_velocity_2m"]
    td = P.create_toml_dict("f64", P.SB2006_LIMITERS_OVERRIDE)
    wr = P.WarmRainParams2M(td, is_limited=limited)
    i = g["inputs"]
    r = oracle.sb2006_process_rates(F64, wr.c, P.ThermodynamicsParameters("f64"), P.rain_vel_params("f64"),
                                    _flags(limited, _abi.CMX_VEL_CHEN2022), _a(1e-3), _a(0.0), _a(i["q_rai"]), _a(0.0),
                                    _a(i["N_rai"]), _a(i["rho"]), _a(288.15))
    assert math.isclose(r["rain_vel_n"][0], g["expected"][0], rel_tol=g["rtol"])
    assert math.isclose(r["rain_vel_m"][0], g["expected"][1], rel_tol=g["rtol"])


def test_chen2022_coeffs_table_b1(oracle):
    """test/common_functions_tests.jl:127-151 pins (aiu, bi, ciu) at ρ = 1.2 through closed forms."""
    ch = P.Chen2022VelTypeRain("f64")
    aiu, bi, ciu = oracle.chen2022_rain_coeffs(F64, ch, 1.2)
    q = math.exp(0.115231 * 1.2)
    b = [2.2955 - 0.038465 * 1.2, 2.2955 - 0.038465 * 1.2, 1.1451 - 0.038465 * 1.2]
    a = [0.044612 * q, -0.263166 * q, 4.7178 * q * 1.2 ** (-0.47335)]
    for k in range(3):
        assert math.isclose(bi[k], b[k], rel_tol=1e-15)
        assert math.isclose(aiu[k], a[k] * 1000 ** b[k], rel_tol=1e-14)
    assert ciu == [0.0, 184.325, 184.325]


def test_gamma_incl_against_exact_constants(oracle, golden):
    """test/microphysics2M_tests.jl:512-550: the Γ_incl rational approximation vs exact incomplete gamma."""
    g = golden["gamma_incl_exact"]
    td = P.create_toml_dict("f64", P.SB2006_LIMITERS_OVERRIDE)
    tps = P.ThermodynamicsParameters("f64")
    i = g["inputs"]
    for limited in (True, False):
        wr = P.WarmRainParams2M(td, is_limited=limited)
        sb, aps = wr.c.seifert_beheng, wr.c.air_properties
        r = oracle.sb2006_process_rates(F64, wr.c, tps, None, _flags(limited, 0), _a(i["q_tot"]), _a(i["q_lcl"]),
                                        _a(i["q_rai"]), _a(0.0), _a(i["N_rai"]), _a(i["rho"]), _a(i["T"]))
        # the reference test's inline formula with the exact constants
        pdf = oracle.pdf_rain_parameters(F64, sb.pdf_r, limited, i["q_rai"], i["rho"], i["N_rai"])
        xr = pdf["xr_mean"]
        rho, T = i["rho"], i["T"]
        L = tps.LH_v0 + (tps.cp_v - tps.cp_l) * (T - tps.T_0)
        p_vs = oracle.psat_liquid(F64, tps, T)
        G = 1 / (L / aps.K_therm / T * (L / tps.R_v / T - 1) + tps.R_v * T / aps.D_vapor / p_vs)
        S = (i["q_tot"] - i["q_rai"]) * rho * tps.R_v * T / p_vs - 1
        Dr = (6 / math.pi / 1000.0) ** (1 / 3) * xr ** (1 / 3)
        ev = sb.evap
        N_Re = ev.alpha * xr ** ev.beta * math.sqrt(ev.rho_0 / rho) * Dr / aps.nu_air
        Sc3 = (aps.nu_air / aps.D_vapor) ** (1 / 3)
        Fv0 = ev.av * g["a_vent_0_over_av"] + ev.bv * g["b_vent_0_over_bv"] * Sc3 * math.sqrt(N_Re)
        Fv1 = ev.av * g["a_vent_1_over_av"] + ev.bv * g["b_vent_1_over_bv"] * Sc3 * math.sqrt(N_Re)
        dN = 2 * math.pi * G * S * i["N_rai"] * Dr * Fv0 / xr
        dq = 2 * math.pi * G * S * i["N_rai"] * Dr * Fv1 / rho
        assert math.isclose(r["evap_dN_rai_dt"][0], dN, rel_tol=g["rtol_number"])
        assert math.isclose(r["evap_dq_rai_dt"][0], dq, rel_tol=g["rtol_mass"])
    # derived evaporation constants (src/parameters/Microphysics2M.jl:599-606) vs exact Γ
    ev = P.EvaporationSB2006("f64")
    assert math.isclose(ev.b_vent_1, 0.308 * sp.gamma(2.5 + 1.5 * 0.266) / 6 ** (0.266 / 2 + 0.5), rel_tol=1e-15)
    assert math.isclose(ev.beta_vent_0, -0.101, rel_tol=1e-12)


def _numpy_restatement(sb, vel, limited, q_lcl, q_rai, rho, N_lcl, N_rai, eps=np.finfo(np.float64).eps):
    """Second, independent statement of the SB2006 collision rates (SB2006 Eqs. 4-13, 94-97)."""
    pi = np.pi
    Lc, Lr = rho * q_lcl, rho * q_rai
    ac, cr, se, br, pr = sb.acnv, sb.accr, sb.self, sb.brek, sb.pdf_r
    nu = sb.pdf_c.nu_c
    xc = np.minimum(ac.x_star, Lc / N_lcl)
    tau = 1 - Lc / (Lc + Lr)
    phi_au = ac.A * tau ** ac.a * (1 - tau ** ac.a) ** ac.b
    dqr_au = ac.kcc / 20 / ac.x_star * (nu + 2) * (nu + 4) / (nu + 1) ** 2 * Lc ** 2 * xc ** 2 * (
        1 + phi_au / (1 - tau) ** 2) * (ac.rho_0 / rho) / rho
    dNc_au = -2 / ac.x_star * rho * dqr_au
    dNc_sc = -ac.kcc * (nu + 2) / (nu + 1) * (ac.rho_0 / rho) * Lc ** 2 - dNc_au
    phi_ac = (tau / (tau + cr.tau_0)) ** cr.c
    dqr_ac = cr.kcr * Lc * Lr * phi_ac * np.sqrt(cr.rho_0 / rho) / rho
    dNc_ac = -dqr_ac * rho / (Lc / N_lcl)
    if limited:
        xt = np.clip(Lr / N_rai, pr.xr_min, pr.xr_max)
        N0 = np.clip(N_rai * (pi * pr.rho_w / xt) ** (1 / 3), pr.N0_min, pr.N0_max)
        lam = np.clip((pi * pr.rho_w * N0 / Lr) ** 0.25, pr.lambda_min, pr.lambda_max)
        xr = np.clip(Lr * lam / N0, pr.xr_min, pr.xr_max)
    else:
        xr = Lr / N_rai
        lam = (pi * pr.rho_w / xr) ** (1 / 3)
    Br = (6 / xr) ** (1 / 3)
    sc = -se.krr * N_rai * Lr * (1 + se.kappa_rr / Br) ** se.d * np.sqrt(pr.rho_0 / rho)
    Dr = (xr / pr.rho_w / pi * 6) ** (1 / 3)
    phi_br = np.where(Dr < br.Dr_th, -1.0, np.where(Dr <= br.Deq, br.kbr * (Dr - br.Deq),
                                                    np.exp(br.kappa_br * (Dr - br.Deq)) - 1))
    brk = -(phi_br + 1) * sc
    s = np.sqrt(vel.rho_0 / rho)
    if limited:
        vt0 = np.maximum(0, s * (vel.aR - vel.bR / (1 + vel.cR / lam)))
        vt1 = np.maximum(0, s * (vel.aR - vel.bR / (1 + vel.cR / lam) ** 4))
    else:
        rc = -1 / (2 * vel.cR) * np.log(vel.aR / vel.bR)
        G1 = lambda t: np.exp(-t)  # noqa: E731
        G4 = lambda t: (t ** 3 + 3 * t ** 2 + 6 * t + 6) * np.exp(-t)  # noqa: E731
        vt0 = np.maximum(0, s * (vel.aR * G1(2 * rc * lam) - vel.bR * G1(2 * rc * (lam + vel.cR)) / (1 + vel.cR / lam)))
        vt1 = np.maximum(0, s * (vel.aR * G4(2 * rc * lam) / 6 - vel.bR * G4(2 * rc * (lam + vel.cR)) / 6 / (1 + vel.cR / lam) ** 4))
    return dict(acnv_dq_rai_dt=dqr_au, acnv_dq_lcl_dt=-dqr_au, acnv_dN_lcl_dt=dNc_au, acnv_dN_rai_dt=-0.5 * dNc_au,
                lcl_self_collection=dNc_sc, accr_dq_rai_dt=dqr_ac, accr_dq_lcl_dt=-dqr_ac, accr_dN_lcl_dt=dNc_ac,
                rain_self_collection=sc, rain_breakup=brk, rain_vel_n=vt0, rain_vel_m=vt1)


@pytest.mark.parametrize("limited", [True, False])
@pytest.mark.parametrize("override", [False, True])
def test_collision_rates_vs_numpy_restatement(oracle, limited, override):
    """Random states well inside every gate; rtol 1e-6 like test/microphysics2M_tests.jl:194-452.
    Covers all three breakup regimes (incl. the exponential branch the reference suite never reaches)."""
    rng = np.random.default_rng(7)
    n = 4000
    rho = rng.uniform(0.3, 1.3, n)
    q_lcl = 10 ** rng.uniform(-6, -2.5, n)
    q_rai = 10 ** rng.uniform(-7, -2.3, n)
    N_lcl = 10 ** rng.uniform(6, 9, n)
    N_rai = 10 ** rng.uniform(0.5, 7, n)
    td = P.create_toml_dict("f64", P.SB2006_LIMITERS_OVERRIDE if override else None)
    wr = P.WarmRainParams2M(td, is_limited=limited)
    vel = P.rain_vel_params("f64")
    r = oracle.sb2006_process_rates(F64, wr.c, P.ThermodynamicsParameters("f64"), vel, _flags(limited),
                                    np.full(n, 5e-3), q_lcl, q_rai, N_lcl, N_rai, rho, np.full(n, 285.0))
    ref = _numpy_restatement(wr.c.seifert_beheng, vel.sb2006, limited, q_lcl, q_rai, rho, N_lcl, N_rai)
    regimes = set()
    for k, v in ref.items():
        np.testing.assert_allclose(r[k], v, rtol=1e-6, atol=0, err_msg=k)
    sb = wr.c.seifert_beheng
    ratio = np.where(r["rain_self_collection"] != 0, r["rain_breakup"] / r["rain_self_collection"], np.nan)
    regimes = {"none": np.sum(ratio == 0), "linear": np.sum((ratio < 0) & (ratio > -1)), "exp": np.sum(ratio < -1)}
    assert all(v > 0 for v in regimes.values()), (regimes, sb.brek.Deq)


def test_zero_and_threshold_behaviour(oracle):
    """Limits asserted by the reference (test/microphysics2M_tests.jl:143-192,254-281,316-323,366-378,
    404-415,438-443,552-563): every rate is exactly 0 when its species is absent."""
    wr = P.WarmRainParams2M("f64")
    tps, vel = P.ThermodynamicsParameters("f64"), P.rain_vel_params("f64")
    z = _a(0.0)
    for limited in (True, False):
        wr = P.WarmRainParams2M("f64", is_limited=limited)
        r = oracle.sb2006_process_rates(F64, wr.c, tps, vel, _flags(limited), _a(1e-3), z, z, _a(1e8), _a(1e4), _a(1.1), _a(288.15))
        for k in ("acnv_dq_lcl_dt", "acnv_dN_lcl_dt", "acnv_dq_rai_dt", "acnv_dN_rai_dt", "lcl_self_collection",
                  "accr_dq_lcl_dt", "accr_dN_lcl_dt", "accr_dq_rai_dt", "rain_self_collection", "rain_breakup",
                  "rain_vel_m", "evap_dq_rai_dt", "evap_dN_rai_dt"):
            assert r[k][0] == 0.0, k
        r = oracle.sb2006_process_rates(F64, wr.c, tps, vel, _flags(limited), _a(1e-3), z, _a(1e-6), _a(1e8), z, _a(1.1), _a(288.15))
        assert r["rain_vel_n"][0] == 0.0 and r["evap_dN_rai_dt"][0] == 0.0 and r["rain_self_collection"][0] == 0.0
        # pdf_rain_parameters limiting behaviour (test/microphysics2M_tests.jl:64-92)
        p = oracle.pdf_rain_parameters(F64, wr.c.seifert_beheng.pdf_r, limited, 0.0, 1.2, 0.0)
        assert p == dict(N0r=0.0, Dr_mean=0.0, xr_mean=0.0)


def test_limited_psd_stays_in_bounds(oracle):
    """test/microphysics2M_tests.jl:93-114."""
    td = P.create_toml_dict("f64", P.SB2006_LIMITERS_OVERRIDE)
    pdf = P.RainParticlePDF_SB2006(td, True)
    for Nr in (1e1, 1e1, 1e3, 1e5):
        for qr in (0.0, 1e-3, 1e-4, 1e-2):
            p = oracle.pdf_rain_parameters(F64, pdf, True, qr, 1.0, Nr)
            lam = 1 / p["Dr_mean"]
            assert pdf.lambda_min * (1 - 1e-15) <= lam <= pdf.lambda_max * (1 + 1e-15)
            assert pdf.xr_min <= p["xr_mean"] <= pdf.xr_max


def test_number_adjustment_horn2012(oracle):
    """test/microphysics2M_tests.jl:720-747 semantics: relax n towards q/clamp(q/n, x_min, x_max)."""
    wr = P.WarmRainParams2M("f64")
    sb = wr.c.seifert_beheng
    tps = P.ThermodynamicsParameters("f64")
    rho = 1.2
    q = np.array([1e-4, 1e-4, 1e-4, 0.0])
    x = np.array([sb.pdf_r.xr_min / 10, sb.pdf_r.xr_max * 10, 1e-8, 1.0])
    n = np.where(q > 0, q / x, 5.0)
    r = oracle.sb2006_process_rates(F64, wr.c, tps, None, _flags(True, 0), np.full(4, 5e-3), np.zeros(4), q,
                                    np.zeros(4), n * rho, np.full(4, rho), np.full(4, 290.0))
    exp = np.array([(q[0] / sb.pdf_r.xr_min - n[0]) / sb.numadj.tau, (q[1] / sb.pdf_r.xr_max - n[1]) / sb.numadj.tau,
                    0.0, -n[3] / sb.numadj.tau])
    np.testing.assert_allclose(r["numadj_rai"], exp, rtol=1e-12, atol=1e-12)


def test_fused_entry_equals_sum_of_processes(oracle):
    """The fused tendency has no absolute KAT in the reference; it is pinned transitively:
    per-process KATs + the accumulation order of BMT:738-779, re-stated here in numpy."""
    from cmx import synthetic
    s = synthetic.sb2006_state(20000, seed=3)
    cols = [c.numpy().astype(np.float64) for c in s]
    rho, T, q_tot, q_lcl, n_lcl, q_rai, n_rai = cols
    for limited in (True, False):
        wr = P.WarmRainParams2M("f64", is_limited=limited)
        tps, vel = P.ThermodynamicsParameters("f64"), P.rain_vel_params("f64")
        fused = oracle.sb2006_warm_rain_tendencies(F64, wr.c, tps, vel, _flags(limited), *cols)
        c = lambda a: np.maximum(a, 0.0)  # noqa: E731  BMT:828-837
        rho_c, qt, ql, qr, nl, nr = c(rho), c(q_tot), c(q_lcl), c(q_rai), c(n_lcl), c(n_rai)
        p = oracle.sb2006_process_rates(F64, wr.c, tps, vel, _flags(limited), qt, ql, qr, rho_c * nl, rho_c * nr, rho_c, T)
        np.testing.assert_array_equal(fused["dq_lcl_dt"], p["condevap"] + p["acnv_dq_lcl_dt"] + p["accr_dq_lcl_dt"])
        np.testing.assert_array_equal(fused["dq_rai_dt"], p["evap_dq_rai_dt"] + p["acnv_dq_rai_dt"] + p["accr_dq_rai_dt"])
        dn_l = p["acnv_dN_lcl_dt"] / rho_c + p["lcl_self_collection"] / rho_c + p["accr_dN_lcl_dt"] / rho_c
        # numadj in the fused entry uses n directly, the per-process probe N/ρ: allow the last-bit difference
        np.testing.assert_allclose(fused["dn_lcl_dt"], dn_l + p["numadj_lcl"], rtol=1e-12, atol=1e-9)
        dn_r = p["evap_dN_rai_dt"] / rho_c + p["acnv_dN_rai_dt"] / rho_c + p["rain_self_collection"] / rho_c + p["rain_breakup"] / rho_c
        np.testing.assert_allclose(fused["dn_rai_dt"], dn_r + p["numadj_rai"], rtol=1e-12, atol=1e-9)
        np.testing.assert_array_equal(fused["vt_rai_n"], p["rain_vel_n"])
        # conservation: collisions move mass between cloud and rain only (test/bulk_tendencies_tests.jl:1154-1250)
        np.testing.assert_allclose(p["acnv_dq_lcl_dt"] + p["acnv_dq_rai_dt"], 0, atol=0)
        assert np.all(np.isfinite(fused["dq_lcl_dt"])) and np.all(np.isfinite(fused["dn_rai_dt"]))


def test_float32_arithmetic_oracle_tracks_float64(oracle):
    """The reference's Float32 path (float arithmetic, float gates) vs its Float64 arithmetic with the same
    gates, in the scaled metric of tests/parity.py — documents what ≤1e-3 means on this path."""
    import parity
    from cmx import synthetic
    s = synthetic.sb2006_state(200000, seed=11)
    cols = [c.numpy() for c in s]
    for limited in (True, False):
        fl = _flags(limited)
        r64 = oracle.sb2006_warm_rain_tendencies(F64, P.WarmRainParams2M("f64", limited).c, P.ThermodynamicsParameters("f64"),
                                                 P.rain_vel_params("f64"), fl, *[c.astype(np.float64) for c in cols],
                                                 float32_gates=True, nthreads=4)
        r32 = oracle.sb2006_warm_rain_tendencies(_abi.F32, P.WarmRainParams2M("f32", limited).c, P.ThermodynamicsParameters("f32"),
                                                 P.rain_vel_params("f32"), fl, *cols, nthreads=4)
        parity.assert_parity(r32, r64, 1e-3, what=f"f32-oracle limited={limited}")


def test_bulk_2m_autoconversion_and_accretion_variants(oracle, golden):
    """KK2000 / B1994 / TC1980 / LD2004 (src/Microphysics2M.jl:920-1003) against test/gpu_tests.jl:782-818, plus the limits
    the reference's CPU tests assert (test/microphysics2M_tests.jl:20-85): zero cloud water → zero rate, thresholds."""
    from cmx import _abi
    from cmx import parameters as P
    g = golden["bulk_2m_variants"]
    sc = P.Bulk2MSchemes("f64")
    ids = {"KK2000": _abi.CMX_2M_KK2000, "B1994": _abi.CMX_2M_B1994, "TC1980": _abi.CMX_2M_TC1980, "LD2004": _abi.CMX_2M_LD2004}
    for name, exp in g["acnv"].items():
        a, _ = oracle.bulk_2m_cloud_to_rain(_abi.F64, sc, ids[name], [g["q_lcl"]], None, [g["rho"]], [g["N_d"]])
        assert math.isclose(a[0], exp, rel_tol=g["rtol_acnv"]), name
    for name, exp in g["accr"].items():
        _, b = oracle.bulk_2m_cloud_to_rain(_abi.F64, sc, ids[name], [g["q_lcl"]], [g["q_rai"]], [g["rho"]], [g["N_d"]])
        assert math.isclose(b[0], exp, rel_tol=g["rtol_accr"]), name
    for name, sid in ids.items():
        a, _ = oracle.bulk_2m_cloud_to_rain(_abi.F64, sc, sid, [0.0, -1e-6], None, [1.2, 1.2], [1e8, 1e8])
        assert np.all(a == 0), name
    # TC1980 threshold: q below 4/3 π ρw N_d r₀³ / ρ gives zero with the step, a small positive value with the logistic
    thr = 4 / 3 * math.pi * 1000.0 * 1e8 * 7e-6 ** 3 / 1.2
    a, _ = oracle.bulk_2m_cloud_to_rain(_abi.F64, sc, _abi.CMX_2M_TC1980, [0.9 * thr, 1.1 * thr], None, [1.2] * 2, [1e8] * 2)
    assert a[0] == 0 and a[1] > 0
    s, _ = oracle.bulk_2m_cloud_to_rain(_abi.F64, sc, _abi.CMX_2M_TC1980 | _abi.CMX_2M_SMOOTH_TRANSITION, [0.9 * thr, 1.1 * thr], None,
                                        [1.2] * 2, [1e8] * 2)
    assert 0 < s[0] < s[1] < a[1]
    # B1994: d switches at N_0 = 2e8; the smooth version interpolates
    a, _ = oracle.bulk_2m_cloud_to_rain(_abi.F64, sc, _abi.CMX_2M_B1994, [2e-3] * 2, None, [1.2] * 2, [1.9e8, 2.1e8])
    s, _ = oracle.bulk_2m_cloud_to_rain(_abi.F64, sc, _abi.CMX_2M_B1994 | _abi.CMX_2M_SMOOTH_TRANSITION, [2e-3] * 2, None, [1.2] * 2, [1.9e8, 2.1e8])
    assert a[0] < s[0] and s[1] < a[1] * 10 and np.all(np.isfinite(s))
