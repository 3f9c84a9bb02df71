"""CPU tests: the ARG2000 part of the oracle against the data the reference's tests hold (tests/golden/arg2000_kats.json:
digitised ARG2000 Fig. 1, κ-vs-B consistency), the structural properties the reference asserts, and an INDEPENDENT
mpmath re-statement of the published formulas at 30 digits (this path has no absolute KAT in the reference — see the
fixture's provenance note)."""
import json
import math
from pathlib import Path

import mpmath as mp
import numpy as np
import pytest

from cmx import _abi
from cmx import parameters as P
from cmx.aerosol import AerosolDistribution, Mode_B, Mode_kappa

F64 = _abi.F64
G = json.loads((Path(__file__).parent / "golden" / "arg2000_kats.json").read_text())


def _conditions():
    tps = P.ThermodynamicsParameters("f64")
    T, p, w = G["conditions"]["T"], G["conditions"]["p"], G["conditions"]["w"]
    dcl = tps.cp_v - tps.cp_l
    p_vs = tps.press_triple * (T / tps.T_triple) ** (dcl / tps.R_v) * math.exp((tps.LH_v0 - dcl * tps.T_0) / tps.R_v * (1 / tps.T_triple - 1 / T))
    q_vs = 1 / (1 - (tps.R_v / tps.R_d) * (p_vs - p) / p_vs)
    return tps, T, p, w, q_vs


def _sulfate_B(FT, r, s, N):
    x = P.Sulfate(FT)
    return Mode_B(r, s, N, (1.0,), (x.eps,), (x.phi,), (x.M,), (x.nu,), (x.rho,))


def _sulfate_k(FT, r, s, N):
    x = P.Sulfate(FT)
    return Mode_kappa(r, s, N, (1.0,), (1.0,), (x.M,), (x.kappa,))


def _run(oracle, ap, ad, T, p, w, q_tot, **kw):
    a = lambda v: np.atleast_1d(np.asarray(v, dtype=np.float64))  # noqa: E731
    return oracle.arg2000_activation(F64, ap, ad.c_struct(ap, F64), P.AirProperties("f64"), P.ThermodynamicsParameters("f64"),
                                     a(T), a(p), a(w), a(q_tot), **kw)


def test_digitised_fig1_of_arg2000(oracle):
    tps, T, p, w, q_vs = _conditions()
    ap = P.AerosolActivationParameters("f64")
    f = G["fig1"]
    m1 = f["mode1"]
    for mk, rtol in ((_sulfate_B, f["rtol_B"]), (_sulfate_k, f["rtol_kappa"])):
        frac = []
        for N2 in f["N_2_per_cm3"]:
            ad = AerosolDistribution([mk("f64", m1["r_dry"], m1["stdev"], m1["N"]), mk("f64", m1["r_dry"], m1["stdev"], N2 * 1e6)])
            frac.append(_run(oracle, ap, ad, T, p, w, q_vs)["N_act"][0][0] / m1["N"])
        # Julia's `isapprox(::Vector, ::Vector; rtol)` — what the reference test evaluates — is NORM-based:
        # ‖x − y‖₂ ≤ rtol · max(‖x‖₂, ‖y‖₂)
        obs = np.array(f["N_act_fraction_mode1"])
        err = np.linalg.norm(np.array(frac) - obs) / max(np.linalg.norm(frac), np.linalg.norm(obs))
        assert err <= rtol, (err, frac)


def test_kappa_vs_B_consistency_and_structure(oracle):
    tps, T, p, w, q_vs = _conditions()
    for override in (None, P.ARG2000_CALIBRATED_OVERRIDE):
        ap = P.AerosolActivationParameters(P.create_toml_dict("f64", override))
        g = G["gpu_consistency"]
        for m in g["modes"]:
            B = Mode_B(m["r_dry"], m["stdev"], m["N"], (1.0,), (m["eps"],), (m["phi"],), (m["M"],), (m["nu"],), (m["rho"],))
            K = Mode_kappa(m["r_dry"], m["stdev"], m["N"], (1.0,), (1.0,), (m["M"],), (m["kappa"],))
            assert math.isclose(B.hygroscopicity(ap), K.hygroscopicity(ap), rel_tol=g["rtol_hygro"])
            rB = _run(oracle, ap, AerosolDistribution([B]), T, p, w, q_vs)
            rK = _run(oracle, ap, AerosolDistribution([K]), T, p, w, q_vs)
            assert math.isclose(rB["N_act"][0][0], rK["N_act"][0][0], rel_tol=g["rtol_act"])
            assert math.isclose(rB["M_act"][0][0], rK["M_act"][0][0], rel_tol=g["rtol_act"])
            assert rB["N_act"][0][0] > 0 and rB["M_act"][0][0] > 0 and rB["S_max"][0] >= 0
        # order of modes does not matter; same aerosol → same hygroscopicity (aerosol_activation_tests.jl:192-234)
        a = _sulfate_B("f64", 0.05e-6, 2.0, 1e8)
        b = Mode_B(0.243e-6, 1.4, 1e8, (1.0,), (1.0,), (0.9,), (0.058443,), (2.0,), (2170.0,))
        r1 = _run(oracle, ap, AerosolDistribution([a, b]), T, p, w, q_vs)
        r2 = _run(oracle, ap, AerosolDistribution([b, a]), T, p, w, q_vs)
        assert math.isclose(sum(x[0] for x in r1["N_act"]), sum(x[0] for x in r2["N_act"]), rel_tol=1e-14)
        assert math.isclose(r1["S_max"][0], r2["S_max"][0], rel_tol=1e-14)
        # liquid / ice sinks reduce the maximum supersaturation (AA:187-197)
        s0 = _run(oracle, ap, AerosolDistribution([a]), T, p, w, q_vs)["S_max"][0]
        s1 = _run(oracle, ap, AerosolDistribution([a]), T, p, w, q_vs, q_liq=np.array([1e-4]), q_ice=np.array([0.0]),
                  N_liq=np.array([1e8]), N_ice=np.array([0.0]))["S_max"][0]
        assert 0 <= s1 < s0


def _mp_restatement(ap, ad_c, aip, tps, T, p, w, q_tot):
    """Abdul-Razzak & Ghan (2000) / Korolev & Mazin (2003, A11) from the published formulas, 30 digits."""
    mp.mp.dps = 30
    f = mp.mpf
    T, p, w, q = f(T), f(p), f(w), f(q_tot)
    Rv, Rd = f(tps.R_v), f(tps.R_d)
    R_m = Rd * (1 - q) + Rv * q
    cp_m = f(tps.cp_d) + (f(tps.cp_v) - f(tps.cp_d)) * q
    dcl = f(tps.cp_v) - f(tps.cp_l)
    L = f(tps.LH_v0) + dcl * (T - f(tps.T_0))
    p_vs = f(tps.press_triple) * (T / f(tps.T_triple)) ** (dcl / Rv) * mp.e ** ((f(tps.LH_v0) - dcl * f(tps.T_0)) / Rv * (1 / f(tps.T_triple) - 1 / T))
    rho = p / (R_m * T)
    p_v = q * rho * Rv * T
    G = 1 / (L / f(aip.K_therm) / T * (L / Rv / T - 1) + Rv * T / f(aip.D_vapor) / p_vs) / f(ap.rho_w)
    alpha = p_v / p_vs * (L * f(ap.g) / Rv / cp_m / T ** 2 - f(ap.g) / R_m / T)
    gamma = Rv * T / p_vs + p_v / p_vs * R_m * L ** 2 / Rv / cp_m / T / p
    A = 2 * f(ap.sigma) * f(ap.M_w) / f(ap.rho_w) / f(ap.R) / T
    zeta = 2 * A / 3 * mp.sqrt(alpha * w / G)
    tot, Sm = f(0), []
    for k in range(ad_c.n_modes):
        m = ad_c.modes[k]
        sm = 2 / mp.sqrt(f(m.hygroscopicity)) * (A / 3 / f(m.r_dry)) ** f(1.5)
        ls = mp.log(f(m.stdev))
        fi, gi = f(ap.f1) * mp.e ** (f(ap.f2) * ls ** 2), f(ap.g1) + f(ap.g2) * ls
        eta = (alpha * w / G) ** f(1.5) / (2 * mp.pi * f(ap.rho_w) * gamma * f(m.N))
        tot += (fi * (zeta / eta) ** f(ap.p1) + gi * (sm ** 2 / (eta + 3 * zeta)) ** f(ap.p2)) / sm ** 2
        Sm.append(sm)
    smax = 1 / mp.sqrt(tot)
    n_act, m_act = [], []
    for k in range(ad_c.n_modes):
        m = ad_c.modes[k]
        ls = mp.log(f(m.stdev))
        u = 2 * mp.log(Sm[k] / smax) / (3 * mp.sqrt(2) * ls)
        n_act.append(float(f(m.N) / 2 * mp.erfc(u)))
        m_act.append(float(f(m.molar_mass_mix) / 2 * mp.erfc(u - 3 * ls * mp.sqrt(2) / 2)))
    return float(smax), n_act, m_act


def test_against_independent_mpmath_restatement(oracle):
    from cmx import synthetic
    ap, aip, tps = P.AerosolActivationParameters("f64"), P.AirProperties("f64"), P.ThermodynamicsParameters("f64")
    ad = synthetic.arg_config3_distribution()
    adc = ad.c_struct(ap, F64)
    st = synthetic.arg_state(40, dtype=__import__("torch").float64, seed=5)
    cols = [c.numpy() for c in st]
    r = oracle.arg2000_activation(F64, ap, adc, aip, tps, *cols)
    for i in range(40):
        smax, n_act, m_act = _mp_restatement(ap, adc, aip, tps, *[c[i] for c in cols])
        assert math.isclose(r["S_max"][i], smax, rel_tol=1e-12)
        for k in range(5):
            assert math.isclose(r["N_act"][k][i], n_act[k], rel_tol=1e-9, abs_tol=1e-12 * adc.modes[k].N)
            assert math.isclose(r["M_act"][k][i], m_act[k], rel_tol=1e-9, abs_tol=1e-12 * adc.modes[k].molar_mass_mix)


def test_distribution_validation():
    a = _sulfate_B("f64", 1e-7, 2.0, 1e8)
    k = _sulfate_k("f64", 1e-7, 2.0, 1e8)
    with pytest.raises(TypeError):
        AerosolDistribution([a, k])
    with pytest.raises(ValueError):
        AerosolDistribution([a] * 9)
    ap = P.AerosolActivationParameters("f64")
    assert math.isclose(k.hygroscopicity(ap), 0.53)
    assert math.isclose(a.hygroscopicity(ap), 3 * 1.0 * 1.0 / 0.132 * 1770.0 * 0.01801528 / 1000.0, rel_tol=1e-14)
