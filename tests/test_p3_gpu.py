"""GPU parity tests of the P3 shape-solver kernel through the C ABI: the reference's KATs (D_m, robustness sweep,
absent ice), random-state parity of (F_rim, ρ_rim, logλ, D_m, log N₀) against the oracle for both input conventions and
both float types, and BASELINE config 5's size (1e7 Float64 columns) through size-independent properties.

How logλ is compared.  The shape residual logLdivN(logλ) − log(L/N) is NOT monotonic where μ(λ) ramps from 0 to 6
(logλ ∈ [8.7, 10.4]): some states have three roots, and the reference's fixed Brent budget (8/10 iterations) leaves
≈5-10 % of random states short of convergence.  The device runs the SAME restated algorithm with the SAME budget as
the oracle, so the two are compared iterate-for-iterate: every point must agree to the north-star tolerance except a
counted handful (< 0.1 %) where a rounding-level difference flipped a Brent branch decision in a multi-root state.
With a converged budget (brent_iters = 40) the same holds against the converged oracle, and every disagreeing point is
certified to be a genuine root of the residual (|residual| ≤ 1e-6), i.e. another solution of the same equation."""
import itertools
import json
import math
from pathlib import Path

import numpy as np
import pytest
import torch

from cmx import _abi
from cmx import parameters as P

pytestmark = pytest.mark.gpu
DT = {"f32": torch.float32, "f64": torch.float64}
G = json.loads((Path(__file__).parent / "golden" / "p3_kats.json").read_text())
ALL = ("F_rim", "rho_rim", "log_lambda", "D_m", "log_N0")
RTOL = {"f64": 1e-6, "f32": 1e-3}            # north_star: ≤1e-6 (Float64) / ≤1e-3 (Float32)
EPS = {"f64": np.finfo(np.float64).eps, "f32": float(np.finfo(np.float32).eps)}
MAX_FLIPPED = 1e-3                            # fraction of points allowed to sit on another Brent path / root
STATE = _abi.CMX_P3_INPUT_IS_STATE


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _np64(t):
    return t.cpu().numpy().astype(np.float64)


def _solver_parity(oracle, got, rho_q, rho_n, ft, slope_flags, budget, what):
    """Compare the device's (logλ, D_m, log N₀) with the oracle run in Float64 arithmetic with ft's gates from the
    device's own regularised state (F_rim, ρ_rim), at Brent budget `budget` (0 → the reference's)."""
    p64 = P.ParametersP3("f64", "constant" if slope_flags else "powerlaw").c
    F, rr = _np64(got.F_rim), _np64(got.rho_rim)
    ref = oracle.p3_shape(_abi.F64, p64, STATE | slope_flags, rho_q, rho_n, F, rr, float32_gates=(ft == "f32"),
                          maxiters=budget if budget > 0 else (8 if ft == "f32" else 10), nthreads=8)
    ll, rl = _np64(got.log_lambda), ref["log_lambda"]
    assert np.array_equal(np.isneginf(ll), np.isneginf(rl)), what
    fin = np.isfinite(rl)
    assert np.all(np.isfinite(ll[fin])) and np.all((ll[fin] >= 2) & (ll[fin] <= 17)), what
    err = np.zeros_like(rl)
    err[fin] = np.abs(ll[fin] - rl[fin]) / np.abs(rl[fin])
    ok = fin & (err <= RTOL[ft])
    flipped = fin & ~ok
    rep = {"n": int(fin.sum()), "flipped": int(flipped.sum()), "log_lambda": float(err[ok].max(initial=0.0))}
    assert flipped.sum() <= MAX_FLIPPED * fin.sum(), (what, rep)
    for k in ("D_m", "log_N0"):
        x, r = _np64(getattr(got, k))[ok], ref[k][ok]
        # both inherit the root's error: |d ln D_m/d logλ| ≈ 1 and |d log N₀/d logλ| = μ+1 ≤ 7, with |logλ| ≤ 17
        e = np.abs(x - r) / np.maximum(np.abs(r), 1.0 if k == "log_N0" else 1e-300)
        rep[k] = float(e.max(initial=0.0))
        assert rep[k] <= RTOL[ft] * 20, (what, k, rep)
    import parity
    parity.record("P3 shape " + what, ft, {k: _np64(getattr(got, k)) for k in ("log_lambda", "D_m", "log_N0")}, ref, family="P3 shape solve (a5)",
                  pinned_by="oracle restatement of src/P3_size_distribution.jl:284-320 (KATs: D_m, thresholds, robustness sweep); iterate-for-iterate at the same Brent budget",
                  keep=ok, scale={"log_N0": np.ones_like(rl)},
                  note=f"{int(flipped.sum())} of {int(fin.sum())} states on another Brent path / root are set aside (multi-root band of the residual)")
    return rep, flipped, ref


def _state_parity(oracle, got, cols64, ft, from_state):
    """(F_rim, ρ_rim) of state_from_prognostic against Float64 arithmetic with ft's gates.  Inside the blending band
    of the regularised ratio, eps/4 ≤ denominator ≤ 42 eps (Utilities.jl:445-488), the weight (1+tanh(2 atanh(1 −
    2(1−a)^(…))))/2 is evaluated by the reference on 1−a with a ≈ eps: its value there is rounding noise of FT, so only
    bounds are checked (0 ≤ ρ_rim ≤ q_rim/b_rim); outside the band the ratio is exact."""
    p64 = P.ParametersP3("f64").c
    ref = oracle.p3_shape(_abi.F64, p64, STATE if from_state else 0, *cols64, float32_gates=(ft == "f32"), maxiters=1, nthreads=8)
    F, rr = _np64(got.F_rim), _np64(got.rho_rim)
    if from_state:
        assert np.array_equal(F, cols64[2]) and np.array_equal(rr, cols64[3])
        return {"band": 0}
    L, _, q_rim, b_rim = cols64
    eps = EPS[ft]
    band_F = (L >= eps / 4) & (L <= 42 * eps)
    band_r = (b_rim >= eps / 4) & (b_rim <= 42 * eps)
    eF = np.abs(F - ref["F_rim"])[~band_F]
    er = (np.abs(rr - ref["rho_rim"]) / np.maximum(ref["rho_rim"], 1.0))[~band_r]
    assert eF.max(initial=0.0) <= RTOL[ft] and er.max(initial=0.0) <= RTOL[ft], (eF.max(initial=0.0), er.max(initial=0.0))
    cap = np.minimum(q_rim[band_r] / b_rim[band_r], 800.0)
    assert np.all(rr[band_r] >= 0) and np.all(rr[band_r] <= cap * (1 + 1e-3))
    return {"band": float(band_r.mean()), "F_rim": float(eF.max(initial=0.0)), "rho_rim": float(er.max(initial=0.0))}


@pytest.mark.parametrize("ft", ["f64", "f32"])
def test_kats_and_robustness_through_the_abi(dev, ft):
    import cmx
    p = P.ParametersP3(ft)
    g = G["D_m"]
    col = lambda v: torch.tensor(v, dtype=DT[ft], device=dev)  # noqa: E731
    r = cmx.p3_shape(p, col([g["L_ice"]] * 2), col([g["N_ice"]] * 2), col(g["F_rim"]), col([g["rho_rim"]] * 2), from_state=True)
    np.testing.assert_allclose(r.D_m.cpu().numpy(), g["expected"], rtol=g["rtol"] if ft == "f64" else 2e-3)
    s = G["robustness_sweep"]
    grid = np.array(list(itertools.product(s["L_ice"], s["N_ice"], s["F_rim"], s["rho_rim"]))).T
    r = cmx.p3_shape(p, *[col(c) for c in grid], from_state=True, want=("log_lambda",))
    ll = r.log_lambda.cpu().numpy()
    assert np.all(np.isfinite(ll)) and np.all((ll >= 2) & (ll <= 17))
    e = G["regression_state"]
    r = cmx.p3_shape(p, col([e["L_ice"]]), col([e["N_ice"]]), col([e["F_rim"]]), col([e["rho_rim"]]), from_state=True)
    assert 2 < r.log_lambda[0].item() < 17
    r = cmx.p3_shape(p, col([0.0, 1e-4]), col([1e5, 0.0]), col([0.0, 0.0]), col([400.0, 400.0]), from_state=True)
    assert bool(torch.isneginf(r.log_lambda).all())


def _columns(n, ft, from_state, seed=1234):
    from cmx import synthetic
    st = synthetic.p3_state(n, dtype=torch.float64, seed=seed)
    if from_state:   # (F_rim, ρ_rim) columns as in P3State(params, L, N, F_rim, ρ_rim)
        F = torch.where(st.rho_q_ice > 0, st.rho_q_rim / st.rho_q_ice.clamp(min=1e-300), torch.zeros_like(st.rho_q_ice))
        rr = torch.where(st.rho_b_rim > 0, st.rho_q_rim / st.rho_b_rim.clamp(min=1e-300), torch.full_like(F, 400.0))
        cols = (st.rho_q_ice, st.rho_n_ice, F, rr)
    else:
        cols = tuple(st)
    return [c.to(DT[ft]) for c in cols]


@pytest.mark.parametrize("ft", ["f64", "f32"])
@pytest.mark.parametrize("from_state", [False, True])
def test_random_state_parity_at_the_reference_budget(dev, oracle, ft, from_state):
    import cmx
    n = 200_000
    cols = _columns(n, ft, from_state)
    r = cmx.p3_shape(P.ParametersP3(ft), *[c.to(dev) for c in cols], from_state=from_state, want=ALL)
    torch.cuda.synchronize()
    cols64 = [c.numpy().astype(np.float64) for c in cols]
    srep = _state_parity(oracle, r, cols64, ft, from_state)
    rep, _, ref = _solver_parity(oracle, r, cols64[0], cols64[1], ft, 0, 0, f"{ft} from_state={from_state}")
    print(f"\n[P3 parity @reference budget] {ft} from_state={from_state}: state {srep} solver {rep}")
    assert np.isneginf(ref["log_lambda"]).mean() > 0.005          # the absent-ice path is exercised


@pytest.mark.parametrize("ft", ["f64", "f32"])
def test_random_state_parity_converged(dev, oracle, ft):
    import cmx
    n = 200_000
    cols = _columns(n, ft, True, seed=99)
    r = cmx.p3_shape(P.ParametersP3(ft), *[c.to(dev) for c in cols], from_state=True, want=ALL, brent_iters=40)
    torch.cuda.synchronize()
    cols64 = [c.numpy().astype(np.float64) for c in cols]
    rep, flipped, _ = _solver_parity(oracle, r, cols64[0], cols64[1], ft, 0, 80, f"{ft} converged")
    # every disagreeing point is another genuine root of the same residual
    p64 = P.ParametersP3("f64").c
    ll = _np64(r.log_lambda)
    worst = 0.0
    for i in np.flatnonzero(flipped):
        res = oracle.p3_logLdivN(_abi.F64, p64, 0, cols64[2][i], cols64[3][i], ll[i]) - (np.log(cols64[0][i]) - np.log(cols64[1][i]))
        worst = max(worst, abs(res))
    assert worst <= (1e-6 if ft == "f64" else 2e-3), worst
    print(f"\n[P3 parity converged] {ft}: {rep}; worst residual on the other-root points {worst:.2e}")


def test_constant_slope_and_errors(dev, oracle):
    import cmx
    from cmx import synthetic
    st = synthetic.p3_state(20_000, seed=5)
    p = P.ParametersP3("f64", "constant")
    r = cmx.p3_shape(p, *[c.to(dev) for c in st], want=ALL)
    cols64 = [c.numpy() for c in st]
    rep, _, _ = _solver_parity(oracle, r, cols64[0], cols64[1], "f64", _abi.CMX_P3_SLOPE_CONSTANT, 0, "constant slope")
    assert rep["flipped"] == 0          # μ constant: the residual is monotonic, a single root
    with pytest.raises(TypeError):
        cmx.p3_shape(P.ParametersP3("f32"), *[c.to(dev) for c in st])
    with pytest.raises(ValueError):
        cmx.p3_shape(p, *[c.to(dev) for c in st], want=("nope",))
    z = [c[:0].to(dev) for c in st]
    assert cmx.p3_shape(p, *z).D_m.shape == (0,)


def test_full_size_1e7_f64_properties(dev, oracle):
    """BASELINE config 5: 1e7 Float64 columns."""
    import cmx
    from cmx import sharding, synthetic
    n = 10_000_000
    st = synthetic.p3_state(n, dtype=torch.float64, device=dev, seed=1234)
    p = P.ParametersP3("f64")
    full = cmx.p3_shape(p, *st, want=ALL)
    torch.cuda.synchronize()
    ll = full.log_lambda
    none = st.rho_q_ice == 0
    assert bool(torch.isneginf(ll[none]).all()) and bool(torch.isfinite(ll[~none]).all())
    assert bool(((ll[~none] >= 2) & (ll[~none] <= 17)).all())
    assert bool((full.D_m[~none] > 0).all()) and bool(torch.isfinite(full.D_m[~none]).all())
    # chunk invariance over the 8-rank shard layout (pointwise solver: every shard evaluated alone is bit-identical)
    for rk in (0, 5, 7):
        lo, hi = sharding.shard_bounds(n, rk, 8)
        part = cmx.p3_shape(p, *[c[lo:hi] for c in st])
        assert torch.equal(part.log_lambda, ll[lo:hi]) and torch.equal(torch.nan_to_num(part.D_m), torch.nan_to_num(full.D_m[lo:hi]))
    # value parity on a strided sample of the full-size run
    stride = 97
    samp = cmx.P3Shape(*[c[::stride].contiguous() for c in full])
    rho_q, rho_n = [_np64(c[::stride]) for c in (st.rho_q_ice, st.rho_n_ice)]
    rep, _, _ = _solver_parity(oracle, samp, rho_q, rho_n, "f64", 0, 0, "1e7 sample")
    print(f"\n[P3 parity 1e7 f64, {rho_q.size} sampled points] {rep}")


# ----------------------------------------------------------------------------------------------------------------
# number- and mass-weighted fall speeds (cmx_p3_terminal_velocities_*)
# ----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("ft", ["f64", "f32"])
def test_bulk_fall_speed_kats_through_the_abi(dev, ft):
    import cmx
    g = G["bulk_velocity"]
    p, vel, quad = P.ParametersP3(ft), P.Chen2022VelTypeIce(ft), P.GaussLegendre(ft, g["quad"]["n"])
    n = len(g["F_rim"])
    col = lambda v: torch.tensor(v, dtype=DT[ft], device=dev)  # noqa: E731
    cols = (col([g["L_ice"]] * n), col([g["N_ice"]] * n), col(g["F_rim"]), col([g["rho_rim"]] * n))
    ll = cmx.p3_shape(p, *cols, from_state=True, want=("log_lambda",)).log_lambda
    rho_a = col([g["rho_a"]] * n)
    loose = 1.0 if ft == "f64" else 30.0
    for ar, sfx, tight in ((True, "oblate", True), (False, "no_aspect_ratio", False)):
        v = cmx.p3_terminal_velocities(p, vel, rho_a, *cols, ll, from_state=True, aspect_ratio=ar, quad=quad)
        np.testing.assert_allclose(v.v_n.cpu().numpy(), g[f"v_n_{sfx}"], rtol=(1e-9 if tight else g["rtol_v_n"]) * loose if ft == "f64" else 3e-3)
        np.testing.assert_allclose(v.v_m.cpu().numpy(), g[f"v_m_{sfx}"], rtol=(1e-9 if tight else g["rtol_v_m"]) * loose if ft == "f64" else 3e-3)
    # absent ice → exactly zero (test/p3_tests.jl:352-364)
    v = cmx.p3_terminal_velocities(p, vel, col([1.2, 1.2]), col([0.0, 0.22]), col([1e6, 0.0]), col([0.5, 0.5]), col([800.0, 800.0]),
                                   col([10.0, 10.0]), from_state=True, quad=quad)
    assert bool((v.v_n == 0).all()) and bool((v.v_m == 0).all())
    with pytest.raises(TypeError):
        cmx.p3_terminal_velocities(p, vel, rho_a, *cols, ll, quad=P.GaussLegendre("f32" if ft == "f64" else "f64", 12))


@pytest.mark.parametrize("ft", ["f64", "f32"])
@pytest.mark.parametrize("rule", ["ChebyshevGauss100", "GaussLegendre40"])
@pytest.mark.parametrize("aspect", [True, False])
def test_random_state_fall_speed_parity(dev, oracle, ft, rule, aspect):
    import cmx
    from cmx import synthetic
    n = 20_000
    cols = _columns(n, ft, True, seed=77)
    rho_a = synthetic.p3_air_density(n, dtype=DT[ft])
    p, vel = P.ParametersP3(ft), P.Chen2022VelTypeIce(ft)
    mk = (lambda f: P.ChebyshevGauss(f, 100)) if rule.startswith("Cheb") else (lambda f: P.GaussLegendre(f, 40))
    dcols = [c.to(dev) for c in cols]
    ll = cmx.p3_shape(p, *dcols, from_state=True, want=("log_lambda",), brent_iters=40).log_lambda
    v = cmx.p3_terminal_velocities(p, vel, rho_a.to(dev), *dcols, ll, from_state=True, aspect_ratio=aspect, quad=mk(ft))
    torch.cuda.synchronize()
    c64 = [c.numpy().astype(np.float64) for c in cols]
    flags = STATE | (0 if aspect else _abi.CMX_P3_NO_ASPECT_RATIO)
    # reference: Float64 arithmetic with ft's gates at the SAME logλ the kernel was given
    r_n, r_m = oracle.p3_terminal_velocities(_abi.F64, P.ParametersP3("f64").c, P.Chen2022VelTypeIce("f64"), mk("f64"), flags, *c64,
                                             rho_a.numpy().astype(np.float64), _np64(ll), float32_gates=(ft == "f32"), nthreads=8)
    rep = {}
    for name, x, r in (("v_n", _np64(v.v_n), r_n), ("v_m", _np64(v.v_m), r_m)):
        assert np.array_equal(x == 0, r == 0), name
        nz = r != 0
        e = np.abs(x[nz] - r[nz]) / np.abs(r[nz])
        rep[name] = float(e.max())
        assert rep[name] <= RTOL[ft], (name, rep)
    import parity
    parity.record(f"P3 fall speeds {ft} {rule} aspect={aspect}", ft, {"v_n": _np64(v.v_n), "v_m": _np64(v.v_m)}, {"v_n": r_n, "v_m": r_m},
                  family="P3 fall speeds (a5)", pinned_by="oracle restatement of src/P3_terminal_velocity.jl:72-178 + the reference's KATs to 1e-14", assert_wellcond=True)
    assert (r_n == 0).mean() > 0.005 and np.all(r_n >= 0) and np.all(r_m[r_n > 0] >= r_n[r_n > 0] * 0.5)
    print(f"\n[P3 fall speeds] {ft} {rule} aspect={aspect}: {rep}")


def test_fall_speeds_from_prognostic_and_full_size_properties(dev, oracle):
    """BASELINE config 5 at full size (1e7 Float64 columns): the five-in / four-out P3 pass (logλ, D_m, v_n, v_m)."""
    import cmx
    from cmx import sharding, synthetic
    n = 10_000_000
    st = synthetic.p3_state(n, dtype=torch.float64, device=dev, seed=1234)
    rho_a = synthetic.p3_air_density(n, dtype=torch.float64, device=dev)
    p, vel, quad = P.ParametersP3("f64"), P.Chen2022VelTypeIce("f64"), P.GaussLegendre("f64", 40)
    shp = cmx.p3_shape(p, *st)
    v = cmx.p3_terminal_velocities(p, vel, rho_a, *st, shp.log_lambda, quad=quad)
    torch.cuda.synchronize()
    none = st.rho_q_ice == 0
    assert bool((v.v_n[none] == 0).all()) and bool((v.v_m[none] == 0).all())
    assert bool(torch.isfinite(v.v_n).all()) and bool(torch.isfinite(v.v_m).all())
    assert bool((v.v_n[~none] > 0).all()) and bool((v.v_n[~none] < 20).all()) and bool((v.v_m[~none] < 40).all())
    assert float((v.v_m[~none] >= v.v_n[~none]).double().mean()) > 0.99      # mass weighting favours the large, fast particles
    for rk in (0, 3, 7):     # chunk invariance over the 8-rank shard layout
        lo, hi = sharding.shard_bounds(n, rk, 8)
        part = cmx.p3_terminal_velocities(p, vel, rho_a[lo:hi], *[c[lo:hi] for c in st], shp.log_lambda[lo:hi], quad=quad)
        assert torch.equal(part.v_n, v.v_n[lo:hi]) and torch.equal(part.v_m, v.v_m[lo:hi])
    stride = 997
    samp = [_np64(c[::stride]) for c in (*st, rho_a, shp.log_lambda)]
    r_n, r_m = oracle.p3_terminal_velocities(_abi.F64, p.c, vel, quad, 0, *samp, nthreads=8)
    e_n = np.abs(_np64(v.v_n[::stride]) - r_n) / np.maximum(r_n, 1e-300)
    e_m = np.abs(_np64(v.v_m[::stride]) - r_m) / np.maximum(r_m, 1e-300)
    ok = r_n > 0
    print(f"\n[P3 fall speeds 1e7 f64, {ok.sum()} sampled points] v_n {e_n[ok].max():.2e} v_m {e_m[ok].max():.2e}")
    assert e_n[ok].max() <= 1e-6 and e_m[ok].max() <= 1e-6


def test_warm_start_guess_on_device(dev, oracle):
    import cmx
    cols = _columns(50_000, "f64", True, seed=5)
    p = P.ParametersP3("f64", "constant")            # monotonic residual → the guess cannot change which root is found
    dcols = [c.to(dev) for c in cols]
    base = cmx.p3_shape(p, *dcols, from_state=True, want=("log_lambda",), brent_iters=40).log_lambda
    fin = torch.isfinite(base)
    for guess in (base + 0.3, base - 0.5, torch.full_like(base, float("nan")), torch.full_like(base, 30.0)):
        ll = cmx.p3_shape(p, *dcols, from_state=True, want=("log_lambda",), brent_iters=40, log_lambda_guess=guess).log_lambda
        assert float((ll[fin] - base[fin]).abs().max()) < 1e-8 and bool(torch.isneginf(ll[~fin]).all())
    # iterate-for-iterate against the oracle at the reference budget, with a guess
    g = (base + 0.2).cpu().numpy()
    ll = cmx.p3_shape(p, *dcols, from_state=True, want=("log_lambda",), log_lambda_guess=base + 0.2).log_lambda
    c64 = [c.numpy() for c in cols]
    ref = oracle.p3_shape(_abi.F64, p.c, STATE | p.flags, *c64, guess=g)["log_lambda"]
    f = np.isfinite(ref)
    assert np.abs(_np64(ll)[f] - ref[f]).max() < 1e-9


@pytest.mark.parametrize("ft", ["f64", "f32"])
def test_ice_melt(dev, oracle, ft):
    """cmx_p3_ice_melt_*: the reference's melting KATs (test/p3_tests.jl:617-668) through the C ABI + random-state parity."""
    import cmx
    from cmx import synthetic
    g = G["ice_melt"]
    p, vel = P.ParametersP3(ft), P.Chen2022VelTypeIce(ft)
    aps, tps, vent = P.AirProperties(ft), P.ThermodynamicsParameters(ft), P.VentilationFactorP3(ft)
    col = lambda v: torch.tensor(v, dtype=DT[ft], device=dev)  # noqa: E731
    n = len(g["T"])
    cols = (col([g["L_ice"]] * n), col([g["N_ice"]] * n), col([g["F_rim"]] * n), col([g["rho_rim"]] * n))
    ll = cmx.p3_shape(p, *cols, from_state=True, want=("log_lambda",)).log_lambda
    r = cmx.p3_ice_melt(p, vel, aps, tps, vent, col(g["T"]), col([g["rho_a"]] * n), *cols, ll, from_state=True,
                        quad=P.GaussLegendre(ft, 12))
    assert float(r.dNdt[0]) == 0 and float(r.dLdt[0]) == 0
    if ft == "f64":
        np.testing.assert_allclose(r.dNdt.cpu().numpy()[1:], g["dNdt"][1:], rtol=1e-9)
        np.testing.assert_allclose(r.dLdt.cpu().numpy()[1:], g["dLdt"][1:], rtol=1e-9)
    else:   # the reference's own Float32 values differ from its Float64 ones by 1e-3 (T − T_freeze = 0.01 K in Float32)
        np.testing.assert_allclose(r.dLdt.cpu().numpy()[1:], g["dLdt"][1:], rtol=3e-3)
    m = 20_000
    st = _columns(m, ft, True, seed=31)
    rho_a = synthetic.p3_air_density(m, dtype=DT[ft])
    gen = torch.Generator().manual_seed(2)
    T = (268.0 + 12.0 * torch.rand(m, generator=gen, dtype=torch.float64)).to(DT[ft])
    dcols = [c.to(dev) for c in st]
    ll = cmx.p3_shape(p, *dcols, from_state=True, want=("log_lambda",), brent_iters=40).log_lambda
    got = cmx.p3_ice_melt(p, vel, aps, tps, vent, T.to(dev), rho_a.to(dev), *dcols, ll, from_state=True, quad=P.GaussLegendre(ft, 40))
    c64 = [c.numpy().astype(np.float64) for c in st]
    dN, dL = oracle.p3_ice_melt(_abi.F64, P.ParametersP3("f64").c, P.Chen2022VelTypeIce("f64"), P.AirProperties("f64"),
                                P.ThermodynamicsParameters("f64"), P.VentilationFactorP3("f64"), P.GaussLegendre("f64", 40), STATE, *c64,
                                rho_a.numpy().astype(np.float64), T.numpy().astype(np.float64), _np64(ll), float32_gates=(ft == "f32"),
                                nthreads=8)
    # T − T_freeze carries the Float32 rounding of T (≈3e-5 K): conditioning scale |T| eps / |T − T_freeze|
    dT = np.abs(T.numpy().astype(np.float64) - 273.15)
    amp = 1.0 + (280.0 * 1.2e-7 / np.maximum(dT, 1e-30) / RTOL[ft] if ft == "f32" else 0.0)
    for x, r_ in ((got.dNdt, dN), (got.dLdt, dL)):
        x = _np64(x)
        assert np.array_equal(x == 0, r_ == 0) or ft == "f32"
        nz = r_ != 0
        e = np.abs(x[nz] - r_[nz]) / np.abs(r_[nz]) / (amp[nz] if ft == "f32" else 1.0)
        assert e.max() <= RTOL[ft], float(e.max())
    import parity
    # operand scale of the melt rate: the rate at |T − T_freeze| = |T| (the Float32 rounding of T is the cancelling operand)
    sc = {k: np.abs(r_) * 273.15 / np.maximum(dT, 1e-30) for k, r_ in (("dNdt", dN), ("dLdt", dL))}
    parity.record(f"P3 ice melt {ft}", ft, {"dNdt": _np64(got.dNdt), "dLdt": _np64(got.dLdt)}, {"dNdt": dN, "dLdt": dL}, family="P3 processes (f2)",
                  pinned_by="oracle restatement of src/P3_processes.jl:64-94 + the reference's melting KATs to 1e-9", scale=sc, assert_wellcond=True)
    assert (dL > 0).mean() > 0.3 and (dL == 0).mean() > 0.3


@pytest.mark.parametrize("ft", ["f64", "f32"])
def test_ice_self_collection(dev, oracle, ft):
    """cmx_p3_ice_self_collection_*: the reference's tests (test/p3_tests.jl:885-917; KA kernel test/gpu_tests.jl:1285-1300:
    positive loss rate, zero without ice) + random-state parity with the oracle (GaussLegendre(12) — 1152 nodes per point)."""
    import cmx
    from cmx import synthetic
    p, vel = P.ParametersP3(ft), P.Chen2022VelTypeIce(ft)
    col = lambda v: torch.tensor(v, dtype=DT[ft], device=dev)  # noqa: E731
    cols = (col([1.2e-4, 0.0]), col([2.4e5, 0.0]), col([0.8, 0.8]), col([800.0, 800.0]))
    ll = cmx.p3_shape(p, *cols, from_state=True, want=("log_lambda",)).log_lambda
    r = cmx.p3_ice_self_collection(p, vel, col([1.2, 1.2]), *cols, ll, from_state=True, quad=P.GaussLegendre(ft, 12))
    assert float(r[0]) > 0 and float(r[1]) == 0
    m = 4_000
    st = _columns(m, ft, True, seed=41)
    rho_a = synthetic.p3_air_density(m, dtype=DT[ft])
    dcols = [c.to(dev) for c in st]
    ll = cmx.p3_shape(p, *dcols, from_state=True, want=("log_lambda",), brent_iters=40).log_lambda
    for aspect in (True, False):
        got = cmx.p3_ice_self_collection(p, vel, rho_a.to(dev), *dcols, ll, from_state=True, aspect_ratio=aspect,
                                         quad=P.GaussLegendre(ft, 12))
        c64 = [c.numpy().astype(np.float64) for c in st]
        ref = oracle.p3_ice_self_collection(_abi.F64, P.ParametersP3("f64").c, P.Chen2022VelTypeIce("f64"), P.GaussLegendre("f64", 12),
                                            STATE | (0 if aspect else _abi.CMX_P3_NO_ASPECT_RATIO), *c64,
                                            rho_a.numpy().astype(np.float64), _np64(ll), float32_gates=(ft == "f32"), nthreads=8)
        x = _np64(got)
        assert np.array_equal(x == 0, ref == 0)
        nz = ref != 0
        e = np.abs(x[nz] - ref[nz]) / ref[nz]
        print(f"\n[P3 self-collection] {ft} aspect={aspect}: max rel err {e.max():.2e}")
        assert e.max() <= RTOL[ft] and np.all(x >= 0)
        import parity
        parity.record(f"P3 self-collection {ft} aspect={aspect}", ft, {"dNdt": x}, {"dNdt": ref}, family="P3 processes (f2)",
                      pinned_by="oracle restatement of src/P3_processes.jl:676-712 (the reference tests sign and zero only)", assert_wellcond=True)


@pytest.mark.parametrize("ft", ["f64", "f32"])
@pytest.mark.parametrize("f_rim", [0.5, 0.9, 0.99])
def test_ice_self_collection_of_heavily_rimed_ice(dev, oracle, ft, f_rim):
    """States in which the partially rimed regime (D > D_cr, mixed area law) carries the integral: the device forms that regime's area^(−½) from the reciprocal
    root of its collision radius (round 5, cmx_p3_kernels.hip / cmx_p3_collisions.hip), the oracle takes exp(−½ ln area) as the reference writes it
    (src/P3_particle_properties.jl ϕᵢ with the mixed area).  Both aspect settings; the same bound as the random-state test."""
    import cmx
    from cmx import synthetic
    m = 1_500
    rng = np.random.default_rng(int(f_rim * 100))
    L = np.exp(rng.uniform(np.log(1e-6), np.log(1e-3), m)); N = np.exp(rng.uniform(np.log(1e2), np.log(1e6), m))
    st = [torch.tensor(v, dtype=DT[ft]) for v in (L, N, np.full(m, f_rim), rng.uniform(200.0, 800.0, m))]
    p, vel = P.ParametersP3(ft), P.Chen2022VelTypeIce(ft)
    rho_a = synthetic.p3_air_density(m, dtype=DT[ft])
    dcols = [c.to(dev) for c in st]
    ll = cmx.p3_shape(p, *dcols, from_state=True, want=("log_lambda",), brent_iters=40).log_lambda
    c64 = [c.numpy().astype(np.float64) for c in st]
    for aspect in (True, False):
        got = _np64(cmx.p3_ice_self_collection(p, vel, rho_a.to(dev), *dcols, ll, from_state=True, aspect_ratio=aspect, quad=P.GaussLegendre(ft, 12)))
        ref = oracle.p3_ice_self_collection(_abi.F64, P.ParametersP3("f64").c, P.Chen2022VelTypeIce("f64"), P.GaussLegendre("f64", 12),
                                            STATE | (0 if aspect else _abi.CMX_P3_NO_ASPECT_RATIO), *c64, rho_a.numpy().astype(np.float64), _np64(ll),
                                            float32_gates=(ft == "f32"), nthreads=8)
        nz = ref != 0
        assert nz.mean() > 0.9 and np.array_equal(got == 0, ref == 0)
        e = np.abs(got[nz] - ref[nz]) / ref[nz]
        print(f"\n[P3 self-collection, F_rim = {f_rim}] {ft} aspect={aspect}: max rel err {e.max():.2e}")
        assert e.max() <= RTOL[ft]


@pytest.mark.parametrize("ft", ["f32", "f64"])
def test_fused_shape_and_velocities_is_the_two_calls(dev, ft):
    """cmx_p3_shape_terminal_velocities_* (BASELINE config 5 as one launch) against cmx_p3_shape_* followed by
    cmx_p3_terminal_velocities_*: the same device functions, so log λ, D_m, v_n, v_m must agree bit for bit; with and without a warm-start
    guess, absent ice, NaN inputs."""
    import cmx
    from cmx import synthetic
    n = 20_011
    st = synthetic.p3_state(n, dtype=DT[ft], device=dev, seed=321)
    rho_a = synthetic.p3_air_density(n, dtype=DT[ft], device=dev, seed=123)
    st = [c.clone() for c in st]
    st[0][5] = 0.0                      # absent ice
    st[1][7] = float("nan")             # NaN input
    p, vel = P.ParametersP3(ft), P.Chen2022VelTypeIce(ft)
    quad = P.GaussLegendre(ft, 20)
    for guess in (None, torch.full((n,), 9.0, dtype=DT[ft], device=dev)):
        shp = cmx.p3_shape(p, *st, log_lambda_guess=guess)
        v = cmx.p3_terminal_velocities(p, vel, rho_a, *st, shp.log_lambda, quad=quad)
        f = cmx.p3_shape_and_terminal_velocities(p, vel, rho_a, *st, log_lambda_guess=guess, quad=quad)
        torch.cuda.synchronize()
        eq = lambda a, b: bool(((a == b) | (torch.isnan(a) & torch.isnan(b))).all())  # noqa: E731
        assert eq(f.log_lambda, shp.log_lambda) and eq(f.D_m, shp.D_m)
        assert eq(f.v_n, v.v_n) and eq(f.v_m, v.v_m)
        assert f.log_lambda[5].item() == float("-inf") and f.v_n[5].item() == 0.0
        assert math.isnan(f.log_lambda[7].item())
