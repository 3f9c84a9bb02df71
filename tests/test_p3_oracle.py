"""CPU tests: the P3 part of the oracle against the reference's known-answer tests (tests/golden/p3_kats.json), its
robustness / round-trip sweeps, and scipy's exact incomplete gamma function."""
import itertools
import json
import math
from pathlib import Path

import numpy as np
import pytest
from scipy import special as sp

from cmx import _abi
from cmx import parameters as P

F64 = _abi.F64
G = json.loads((Path(__file__).parent / "golden" / "p3_kats.json").read_text())
STATE = _abi.CMX_P3_INPUT_IS_STATE


def test_rho_d_and_thresholds(oracle):
    p = P.ParametersP3("f64")
    g = G["rho_d"]
    assert math.isclose(oracle.p3_rho_d(F64, p.c, g["F_rim"], g["rho_rim"]), g["expected"], rel_tol=g["rtol"])
    t = G["fig1a_thresholds_mm"]
    r = oracle.p3_shape(F64, p.c, STATE, [0.22, 0.22], [1e6, 1e6], t["F_rim"], [t["rho_rim"]] * 2)
    np.testing.assert_allclose(1000 * r["D_cr"], t["D_cr"], rtol=t["rtol"])
    np.testing.assert_allclose(1000 * r["D_gr"], t["D_gr"], rtol=t["rtol"])
    # Eq. 17 of Morrison & Milbrandt 2015 re-derived (test/p3_tests.jl:57-77) and D_th < D_gr < D_cr
    D_th = (6 * p.c.alpha_va / (math.pi * p.c.rho_i)) ** (1 / (3 - p.c.beta_va))
    for F, rr in itertools.product((0.5, 0.8, 0.95), (200.0, 400.0, 800.0)):
        s = oracle.p3_shape(F64, p.c, STATE, [0.22], [1e6], [F], [rr])
        D_gr, D_cr = s["D_gr"][0], s["D_cr"][0]
        assert D_th < D_gr < D_cr
        bm2 = p.c.beta_va - 2
        rho_d_paper = 6 * p.c.alpha_va * (D_cr ** bm2 - D_gr ** bm2) / (math.pi * bm2 * (D_cr - D_gr))
        assert math.isclose(rho_d_paper, oracle.p3_rho_d(F64, p.c, F, rr), rel_tol=1e-9)


def test_D_m_kats_with_the_reference_iteration_budget_and_converged(oracle):
    p = P.ParametersP3("f64")
    g = G["D_m"]
    for iters in (0, 60):     # 0 → the reference's 10 fixed Brent iterations; 60 → converged
        r = oracle.p3_shape(F64, p.c, STATE, [g["L_ice"]] * 2, [g["N_ice"]] * 2, g["F_rim"], [g["rho_rim"]] * 2, maxiters=iters)
        np.testing.assert_allclose(r["D_m"], g["expected"], rtol=g["rtol"])
    assert np.all(r["D_m"] > 0)


def test_robustness_sweep_and_absent_ice(oracle):
    p = P.ParametersP3("f64")
    g = G["robustness_sweep"]
    grid = np.array(list(itertools.product(g["L_ice"], g["N_ice"], g["F_rim"], g["rho_rim"]))).T
    r = oracle.p3_shape(F64, p.c, STATE, *grid)
    assert np.all(np.isfinite(r["log_lambda"])) and np.all((r["log_lambda"] >= 2) & (r["log_lambda"] <= 17))
    e = G["regression_state"]
    r = oracle.p3_shape(F64, p.c, STATE, [e["L_ice"]], [e["N_ice"]], [e["F_rim"]], [e["rho_rim"]])
    assert 2 < r["log_lambda"][0] < 17
    r = oracle.p3_shape(F64, p.c, STATE, [0.0, 1e-4, 1e-17], [1e5, 0.0, 1e5], [0.0] * 3, [400.0] * 3)   # p3_tests.jl:181-186
    assert np.all(r["log_lambda"] == -np.inf)


def test_round_trip_of_the_shape_solver(oracle):
    """L_calc = N exp(logLdivN(log λ_ex)) → the solver recovers a root of the SAME residual: |logLdivN(root) − target| tiny,
    and root == log λ_ex whenever the residual is single-signed on either side (test/p3_tests.jl:189-227)."""
    p = P.ParametersP3("f64")
    g = G["round_trip"]
    n_checked = 0
    for N, lam, rr, F in itertools.product(g["N_ice"], g["lambda"], g["rho_rim"], g["F_rim"]):
        ll_ex = math.log(lam)
        L = N * math.exp(oracle.p3_logLdivN(F64, p.c, 0, F, rr, ll_ex))
        if not L < 1.0:
            continue
        r = oracle.p3_shape(F64, p.c, STATE, [L], [N], [F], [rr], maxiters=80)
        ll = r["log_lambda"][0]
        assert abs(oracle.p3_logLdivN(F64, p.c, 0, F, rr, ll) - (math.log(L) - math.log(N))) < 1e-9
        assert abs(ll - ll_ex) <= 1.0 * abs(ll_ex)          # the reference's own (loose) assertion
        n_checked += 1
    assert n_checked > 100


def test_gamma_inc_against_scipy(oracle):
    g = G["gamma_inc_grid"]
    for a, x in itertools.product(g["a"], g["x"]):
        P64, Q64 = oracle.gamma_inc(F64, a, x)
        assert abs(P64 - sp.gammainc(a, x)) <= g["atol_f64"] and abs(Q64 - sp.gammaincc(a, x)) <= g["atol_f64"]
        P32, Q32 = oracle.gamma_inc(_abi.F32, a, x)
        assert abs(P32 - sp.gammainc(a, x)) <= g["atol_f32"] and abs(Q32 - sp.gammaincc(a, x)) <= g["atol_f32"]
    assert oracle.gamma_inc(F64, 2.0, 0.0) == (0.0, 1.0) and oracle.gamma_inc(F64, 2.0, float("inf")) == (1.0, 0.0)


def test_state_from_prognostic_regularisation(oracle):
    """src/P3_particle_properties.jl:101-106 + Utilities.jl:445-509: F_rim = min(q_rim, q_ice)/q_ice clamped below 1,
    ρ_rim = q_rim/b_rim capped at 0.8 ρ_l, both → 0 smoothly for vanishing denominators."""
    p = P.ParametersP3("f64")
    L = np.array([1e-4, 1e-4, 1e-4, 1e-4, 0.0])
    q_rim = np.array([5e-5, 2e-4, 5e-5, 0.0, 0.0])
    b_rim = np.array([1e-7, 4e-7, 1e-9, 0.0, 0.0])
    r = oracle.p3_shape(F64, p.c, 0, L, np.full(5, 1e4), q_rim, b_rim)
    np.testing.assert_allclose(r["F_rim"][:4], [0.5, 1 - np.finfo(float).eps, 0.5, 0.0], rtol=1e-15)
    np.testing.assert_allclose(r["rho_rim"][:4], [500.0, 500.0, 800.0, 0.0], rtol=1e-15)
    assert r["F_rim"][4] == 0 and r["log_lambda"][4] == -np.inf
    assert P.ParametersP3("f64", "constant").flags == _abi.CMX_P3_SLOPE_CONSTANT
    with pytest.raises(ValueError):
        P.ParametersP3("f64", "quadratic")


def test_particle_and_bulk_fall_speed_kats(oracle):
    """test/p3_tests.jl:283-307 (Chen-2022 ice particle velocity per regime, with / without the oblate aspect factor)
    and :336-400 (number- and mass-weighted fall speeds, GaussLegendre(12))."""
    p, vel = P.ParametersP3("f64").c, P.Chen2022VelTypeIce("f64")
    NOAR = _abi.CMX_P3_NO_ASPECT_RATIO
    g = G["particle_velocity"]
    for flags, key in ((NOAR, "expected_no_aspect_ratio"), (0, "expected_oblate")):
        v = [oracle.p3_particle_velocity(F64, p, vel, flags, g["F_rim"], g["rho_rim"], g["rho_a"], D) for D in g["D"]]
        np.testing.assert_allclose(v, g[key], rtol=g["rtol"])
    g = G["bulk_velocity"]
    quad = P.GaussLegendre("f64", g["quad"]["n"])
    n = len(g["F_rim"])
    cols = ([g["L_ice"]] * n, [g["N_ice"]] * n, g["F_rim"], [g["rho_rim"]] * n)
    ll = oracle.p3_shape(F64, p, STATE, *cols)["log_lambda"]
    for flags, sfx, tight in ((NOAR, "no_aspect_ratio", False), (0, "oblate", True)):
        v_n, v_m = oracle.p3_terminal_velocities(F64, p, vel, quad, STATE | flags, *cols, [g["rho_a"]] * n, ll)
        np.testing.assert_allclose(v_n, g[f"v_n_{sfx}"], rtol=1e-13 if tight else g["rtol_v_n"])
        np.testing.assert_allclose(v_m, g[f"v_m_{sfx}"], rtol=1e-13 if tight else g["rtol_v_m"])
    # absent ice → exactly zero (p3_tests.jl:352-364)
    v_n, v_m = oracle.p3_terminal_velocities(F64, p, vel, quad, STATE, [0.0, 0.22], [1e6, 0.0], [0.5] * 2, [800.0] * 2, [1.2] * 2, [10.0] * 2)
    assert np.all(v_n == 0) and np.all(v_m == 0)
    # the default rule ChebyshevGauss(100) agrees with GaussLegendre(40) on the smooth integrals
    a = oracle.p3_terminal_velocities(F64, p, vel, P.ChebyshevGauss("f64", 100), STATE, *cols, [g["rho_a"]] * n, ll)
    b = oracle.p3_terminal_velocities(F64, p, vel, P.GaussLegendre("f64", 40), STATE, *cols, [g["rho_a"]] * n, ll)
    np.testing.assert_allclose(a[0], b[0], rtol=2e-3)
    np.testing.assert_allclose(a[1], b[1], rtol=2e-3)


def test_gamma_inc_inv_against_scipy(oracle):
    for a, pp in itertools.product((0.5, 1.0, 2.0, 3.5, 7.0), (1e-6, 1e-3, 0.3, 0.5, 0.7, 1 - 1e-6)):
        x = oracle.gamma_inc_inv(F64, a, pp, 1 - pp)
        assert math.isclose(x, sp.gammaincinv(a, pp), rel_tol=1e-9), (a, pp)
        x32 = oracle.gamma_inc_inv(_abi.F32, a, pp, 1 - np.float32(pp))
        assert math.isclose(x32, sp.gammaincinv(a, float(np.float32(pp))), rel_tol=5e-3), (a, pp)
    assert oracle.gamma_inc_inv(F64, 2.0, 0.0, 1.0) == 0 and oracle.gamma_inc_inv(F64, 2.0, 1.0, 0.0) == float("inf")


def test_warm_start_guess_keeps_the_root(oracle):
    """get_distribution_logλ(state, logλ_guess): a guess inside the bracket narrows it on its side of the root
    (_narrow_bracket, P3_size_distribution.jl:336-353); invalid guesses (NaN, outside [2, 17]) are ignored."""
    p = P.ParametersP3("f64", "constant")          # monotonic residual: a single root whatever the bracket
    rng = np.random.default_rng(3)
    n = 2000
    L, N = np.exp(rng.uniform(np.log(1e-6), np.log(1e-3), n)), np.exp(rng.uniform(np.log(1e2), np.log(1e6), n))
    F, rr = rng.uniform(0, 0.9, n), rng.uniform(200, 800, n)
    base = oracle.p3_shape(F64, p.c, STATE | p.flags, L, N, F, rr, maxiters=60)["log_lambda"]
    for guess in (base + 0.3, base - 0.5, np.full(n, np.nan), np.full(n, 1.0), np.full(n, 25.0)):
        ll = oracle.p3_shape(F64, p.c, STATE | p.flags, L, N, F, rr, guess=guess, maxiters=60)["log_lambda"]
        np.testing.assert_allclose(ll, base, rtol=0, atol=1e-8)
    # with the reference budget a good guess does not hurt accuracy
    ll10 = oracle.p3_shape(F64, p.c, STATE | p.flags, L, N, F, rr, guess=base + 0.05)["log_lambda"]
    assert np.abs(ll10 - base).max() < 1e-6


def test_ice_melt_kats(oracle):
    """test/p3_tests.jl:617-668: zero at and below freezing, the two reference rates above (GaussLegendre(12))."""
    g = G["ice_melt"]
    p, vel = P.ParametersP3("f64").c, P.Chen2022VelTypeIce("f64")
    aps, tps, vent = P.AirProperties("f64"), P.ThermodynamicsParameters("f64"), P.VentilationFactorP3("f64")
    quad = P.GaussLegendre("f64", 12)
    n = len(g["T"])
    cols = ([g["L_ice"]] * n, [g["N_ice"]] * n, [g["F_rim"]] * n, [g["rho_rim"]] * n)
    ll = oracle.p3_shape(F64, p, STATE, *cols)["log_lambda"]
    dN, dL = oracle.p3_ice_melt(F64, p, vel, aps, tps, vent, quad, STATE, *cols, [g["rho_a"]] * n, g["T"], ll)
    assert dN[0] == 0 and dL[0] == 0
    np.testing.assert_allclose(dN[1:], g["dNdt"][1:], rtol=g["rtol"])
    np.testing.assert_allclose(dL[1:], g["dLdt"][1:], rtol=g["rtol"])
    # dN/dt = N/L · dL/dt exactly
    np.testing.assert_allclose(dN[1:] / dL[1:], g["N_ice"] / g["L_ice"], rtol=1e-14)


def test_ice_self_collection_properties(oracle):
    """test/p3_tests.jl:885-917 (positive loss rate, zero without ice) + quadrature convergence + N² scaling of the
    double integral (doubling N at fixed λ doubles n(D) everywhere → 4× the rate)."""
    p, vel = P.ParametersP3("f64").c, P.Chen2022VelTypeIce("f64")
    cols = ([1.2e-4], [2.4e5], [0.8], [800.0])
    ll = oracle.p3_shape(F64, p, STATE, *cols)["log_lambda"]
    r12 = oracle.p3_ice_self_collection(F64, p, vel, P.GaussLegendre("f64", 12), STATE, *cols, [1.2], ll)[0]
    r40 = oracle.p3_ice_self_collection(F64, p, vel, P.GaussLegendre("f64", 40), STATE, *cols, [1.2], ll)[0]
    rc = oracle.p3_ice_self_collection(F64, p, vel, P.ChebyshevGauss("f64", 100), STATE, *cols, [1.2], ll)[0]
    assert r12 > 0 and abs(r12 / r40 - 1) < 2e-3 and abs(rc / r40 - 1) < 1e-4
    assert oracle.p3_ice_self_collection(F64, p, vel, P.GaussLegendre("f64", 12), STATE, [0.0], [0.0], [0.8], [800.0], [1.2], [-np.inf])[0] == 0
    # same λ (same L/N), twice the number: n(D) doubles → rate × 4
    r2 = oracle.p3_ice_self_collection(F64, p, vel, P.GaussLegendre("f64", 40), STATE, [2.4e-4], [4.8e5], [0.8], [800.0], [1.2], ll)[0]
    assert abs(r2 / r40 - 4) < 1e-9
