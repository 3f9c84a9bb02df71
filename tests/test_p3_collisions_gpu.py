"""GPU parity tests of the P3 liquid–ice collision kernel (cmx_p3_liquid_ice_collisions_*) through the C ABI: the reference's KATs
(test/p3_tests.jl:823-870), its edge cases (no liquid, no rain, above freezing, very cold, absent ice) and random-state parity of
the ten collision integrals and the seven bulk sources against the oracle, Float64 and Float32.

Tolerance.  The integrals are sums of positive terms, so they are compared relatively (north-star 1e-6 / 1e-3).  Two of them are
sensitive to an O(1) decision per outer node — the freeze / shed split min(M_col, M_max) — so QCSHD, QRSHD and ∫𝟙_wet M_col are
compared against the scale of the total collected mass ∫M_col instead of their own (possibly tiny) value."""
import json
from pathlib import Path

import numpy as np
import pytest

import parity
import torch

from cmx import _abi
from cmx import parameters as P

pytestmark = pytest.mark.gpu
PIN = "oracle restatement of src/P3_processes.jl:527-655 + the ten reference collision KATs (test/p3_tests.jl:823-880)"
DT = {"f32": torch.float32, "f64": torch.float64}
G = json.loads((Path(__file__).parent / "golden" / "p3_kats.json").read_text())
RTOL = {"f64": 1e-6, "f32": 1e-3}
STATE = _abi.CMX_P3_INPUT_IS_STATE


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _np64(t):
    return t.cpu().numpy().astype(np.float64)


def test_collision_kats(dev):
    import cmx
    g = G["liquid_ice_collisions"]
    for ft in ("f64", "f32"):
        ip = P.P3IceParams(ft, quad=P.GaussLegendre(ft, 12))
        aps, tps = P.AirProperties(ft), P.ThermodynamicsParameters(ft)
        col = lambda v: torch.tensor([v], dtype=DT[ft], device=dev)  # noqa: E731
        st = (col(g["L_ice"]), col(g["N_ice"]), col(g["F_rim"]), col(g["rho_rim"]))
        ll = cmx.p3_shape(ip_scheme(ft), *st, from_state=True, want=("log_lambda",)).log_lambda
        T = col(ip.c.scheme.T_freeze + g["T_minus_T_freeze"])
        src, rates = cmx.p3_liquid_ice_collisions(ip, aps, tps, col(g["rho_a"]), T, *st, ll, col(g["L_c"]), col(g["N_c"]), col(g["L_r"]),
                                                  col(g["N_r"]), from_state=True, want_rates=True)
        r = {k: float(v[0]) for k, v in rates.items()}
        for k, e in zip(g["names"], g["expected"]):
            assert abs(r[k] - e) <= g["rtol"] * abs(e) + (3e-4 * abs(e) if ft == "f32" else 0), (ft, k, r[k], e)
        if ft == "f64":
            for k in g["reproduced_to_1e-13"]:
                e = dict(zip(g["names"], g["expected"]))[k]
                assert abs(r[k] - e) <= 1e-9 * abs(e), (k, r[k], e)
        assert abs(r["QCFRZ"] + r["QCSHD"] + r["QRFRZ"] + r["QRSHD"] - r["int_M_col"]) <= (1e-12 if ft == "f64" else 1e-5) * r["int_M_col"]
        assert float(src["dL_ice"][0]) > 0 and float(src["dq_c"][0]) < 0 and float(src["dN_c"][0]) < 0


def ip_scheme(ft):
    return P.ParametersP3(ft)


def _random_states(n, ft, seed=77):
    from cmx import synthetic
    st = synthetic.p3_state(n, dtype=torch.float64, seed=seed)
    rng = np.random.default_rng(seed)
    rho = synthetic.p3_air_density(n, dtype=torch.float64).numpy()
    T = rng.uniform(205.0, 285.0, n)
    T[rng.random(n) < 0.2] = rng.uniform(268.0, 276.0, int((rng.random(n) < 0.2).sum()) or 1)[0]
    L_c = np.where(rng.random(n) < 0.8, 10 ** rng.uniform(-6, -2.5, n), 0.0)
    N_c = 10 ** rng.uniform(6.5, 9, n)
    L_r = np.where(rng.random(n) < 0.7, 10 ** rng.uniform(-7, -3, n), 0.0)
    N_r = 10 ** rng.uniform(1, 6, n)
    # (F_rim, ρ_rim) columns as in P3State(params, L, N, F_rim, ρ_rim): the prognostic → state regularisation has its own
    # parity test (test_p3_gpu.py) and, in Float32, a blending band in which it is only bounded
    F = torch.where(st.rho_q_ice > 0, st.rho_q_rim / st.rho_q_ice.clamp(min=1e-300), torch.zeros_like(st.rho_q_ice))
    rr = torch.where(st.rho_b_rim > 0, st.rho_q_rim / st.rho_b_rim.clamp(min=1e-300), torch.full_like(F, 400.0))
    cols = [st.rho_q_ice.numpy(), st.rho_n_ice.numpy(), F.numpy(), rr.numpy(), L_c, N_c, L_r, N_r, rho, T]
    return [torch.from_numpy(np.ascontiguousarray(c)).to(DT[ft]) for c in cols]


@pytest.mark.parametrize("ft", ["f64", "f32"])
@pytest.mark.parametrize("order", [12, 16])
def test_random_state_parity(dev, oracle, ft, order):
    import cmx
    n = 3000
    cols = _random_states(n, ft)
    ip = P.P3IceParams(ft, quadrature_order=16, quad=(P.GaussLegendre(ft, 12) if order == 12 else None))
    aps, tps = P.AirProperties(ft), P.ThermodynamicsParameters(ft)
    d = [c.to(dev) for c in cols]
    ll = cmx.p3_shape(P.ParametersP3(ft), *d[:4], from_state=True, want=("log_lambda",), brent_iters=40).log_lambda
    ll = torch.where(torch.isfinite(ll), ll, torch.zeros_like(ll))
    src, rates = cmx.p3_liquid_ice_collisions(ip, aps, tps, d[8], d[9], *d[:4], ll, *d[4:8], from_state=True, want_rates=True)
    torch.cuda.synchronize()
    ip64 = P.P3IceParams("f64", quadrature_order=16, quad=(P.GaussLegendre("f64", 12) if order == 12 else None))
    c64 = [c.numpy().astype(np.float64) for c in cols]
    osrc, orates = oracle.p3_liquid_ice_collisions(_abi.F64, ip64.c, P.AirProperties("f64"), P.ThermodynamicsParameters("f64"), ip64.c.quad,
                                                   ip64.flags | STATE, *c64[:4], *c64[4:8], c64[8], c64[9], _np64(ll), float32_gates=(ft == "f32"),
                                                   nthreads=8)
    names = list(rates.keys())
    tot = orates[6]
    worst = {}
    for q, k in enumerate(names):
        x, r = _np64(rates[k]), orates[q]
        assert np.all(np.isfinite(x)), k
        scale = np.abs(r) if k not in ("QCSHD", "QRSHD", "int_wet_M_col") else np.maximum(np.abs(r), tot)
        err = np.abs(x - r) / np.maximum(scale, 1e-300)
        err[(r == 0) & (x == 0)] = 0
        worst[k] = err.max()
        assert err.max() <= RTOL[ft], (k, int(err.argmax()), x[err.argmax()], r[err.argmax()])
        parity.record(f"P3 liquid-ice collision integrals {ft} GL{order}", ft, {k: x}, {k: r}, family="P3 collisions (f2)", pinned_by=PIN,
                      scale=None if k not in ("QCSHD", "QRSHD", "int_wet_M_col") else {k: tot}, wellcond=1e-3 if ft == "f64" else 0.1, assert_wellcond=True,
                      note="shedding / wet-growth integrals are differences against the total collected mass (operand scale)")
    print(f"\n[P3 collisions] {ft} GL{order}: worst rel err " + " ".join(f"{k}={v:.1e}" for k, v in worst.items()))
    assert (tot > 0).mean() > 0.5 and (orates[9] > 0).any() and (orates[9] == 0).any()
    # bulk sources: compared against the scale of the integrals they are assembled from
    sc = [tot / c64[8], tot / c64[8], np.abs(orates[2]), np.abs(orates[5]) + orates[4] * 1.9e6, tot + np.abs(osrc[4]), tot, np.abs(osrc[6]) + orates[7] + orates[8]]
    for q, k in enumerate(src.keys()):
        x, r = _np64(src[k]), osrc[q]
        err = np.abs(x - r) / np.maximum(sc[q], 1e-300)
        err[(r == 0) & (x == 0)] = 0
        assert err.max() <= RTOL[ft], (k, int(err.argmax()), x[err.argmax()], r[err.argmax()])
        parity.record(f"P3 bulk collision sources {ft} GL{order}", ft, {k: x}, {k: r}, family="P3 collisions (f2)", pinned_by=PIN, scale={k: sc[q]},
                      note="sums of integrals of both signs: operand scale = the integrals they are assembled from")


def test_edge_cases_and_validation(dev):
    import cmx
    ft = "f64"
    g = G["liquid_ice_collisions"]
    ip = P.P3IceParams(ft, quad=P.GaussLegendre(ft, 12))
    aps, tps = P.AirProperties(ft), P.ThermodynamicsParameters(ft)
    Tf = ip.c.scheme.T_freeze
    col = lambda v: torch.tensor(v, dtype=DT[ft], device=dev)  # noqa: E731
    five = lambda v: col([v] * 5)  # noqa: E731
    st = (col([g["L_ice"]] * 4 + [0.0]), col([g["N_ice"]] * 4 + [0.0]), five(g["F_rim"]), five(g["rho_rim"]))
    ll = cmx.p3_shape(P.ParametersP3(ft), *st, from_state=True, want=("log_lambda",)).log_lambda
    ll = torch.where(torch.isfinite(ll), ll, torch.zeros_like(ll))
    L_c, N_c = col([0.0, 1e-3, 1e-3, 1e-3, 1e-3]), col([0.0, 1e8, 1e8, 1e8, 1e8])
    L_r, N_r = col([0.0, 0.0, 1e-4, 1e-4, 1e-4]), col([0.0, 0.0, 1e6, 1e6, 1e6])
    T = col([Tf - 5, Tf - 5, Tf + 2, 205.0, Tf - 5])
    src, r = cmx.p3_liquid_ice_collisions(ip, aps, tps, five(g["rho_a"]), T, *st, ll, L_c, N_c, L_r, N_r, from_state=True, want_rates=True)
    R = np.stack([_np64(v) for v in r.values()])
    S = np.stack([_np64(v) for v in src.values()])
    assert np.all(R[:, 0] == 0) and np.all(S[:, 0] == 0)                                 # no liquid
    assert np.all(R[3:6, 1] == 0) and R[8, 1] == 0 and R[0, 1] > 0                       # no rain
    assert R[0, 2] == 0 and R[3, 2] == 0 and R[1, 2] > 0 and R[4, 2] > 0 and abs(R[9, 2] - R[6, 2]) <= 1e-14 * R[6, 2]   # above freezing: all shed, all wet (two accumulations of the same terms)
    assert R[1, 3] == 0 and R[4, 3] == 0 and R[9, 3] == 0 and R[0, 3] > 0                # very cold: f_frz = 1
    assert np.all(R[:, 4] == 0) and np.all(S[:, 4] == 0)                                 # absent ice
    with pytest.raises(TypeError):
        cmx.p3_liquid_ice_collisions(P.P3IceParams("f32"), aps, tps, five(1.2), T, *st, ll, L_c, N_c, L_r, N_r)
    # a ragged size: 37 points (2 workgroups + a partial group) equals the same points computed one by one
    n = 37
    cols = [c.to(dev) for c in _random_states(n, ft, seed=3)]
    llr = cmx.p3_shape(P.ParametersP3(ft), *cols[:4], from_state=True, want=("log_lambda",), brent_iters=40).log_lambda
    llr = torch.where(torch.isfinite(llr), llr, torch.zeros_like(llr))
    full = cmx.p3_liquid_ice_collisions(ip, aps, tps, cols[8], cols[9], *cols[:4], llr, *cols[4:8], from_state=True)
    for i in (0, 15, 16, 36):
        one = cmx.p3_liquid_ice_collisions(ip, aps, tps, cols[8][i:i + 1].clone(), cols[9][i:i + 1].clone(), *[c[i:i + 1].clone() for c in cols[:4]],
                                           llr[i:i + 1].clone(), *[c[i:i + 1].clone() for c in cols[4:8]], from_state=True)
        for k in full:
            assert float(full[k][i]) == float(one[k][0]), (k, i)


@pytest.mark.parametrize("quad", ["cheb100", "gl128", "gl40"])
def test_large_quadrature_orders(dev, oracle, quad):
    """The reference's default rule ChebyshevGauss(100), the production order 40 and the largest supported order 128: the per-state LDS
    caches grow with the order and the launch falls back to smaller workgroups (Float64, 128 nodes: 107 KB for 16 states)."""
    import cmx
    ft = "f64"
    mk = {"cheb100": lambda f: P.ChebyshevGauss(f, 100), "gl128": lambda f: P.GaussLegendre(f, 128), "gl40": lambda f: P.GaussLegendre(f, 40)}[quad]
    n = 70
    cols = _random_states(n, ft, seed=19)
    ip = P.P3IceParams(ft, quad=mk(ft))
    aps, tps = P.AirProperties(ft), P.ThermodynamicsParameters(ft)
    d = [c.to(dev) for c in cols]
    ll = cmx.p3_shape(P.ParametersP3(ft), *d[:4], from_state=True, want=("log_lambda",), brent_iters=40).log_lambda
    ll = torch.where(torch.isfinite(ll), ll, torch.zeros_like(ll))
    src, rates = cmx.p3_liquid_ice_collisions(ip, aps, tps, d[8], d[9], *d[:4], ll, *d[4:8], from_state=True, want_rates=True)
    torch.cuda.synchronize()
    c64 = [c.numpy().astype(np.float64) for c in cols]
    _, orates = oracle.p3_liquid_ice_collisions(_abi.F64, ip.c, aps, tps, ip.c.quad, ip.flags | STATE, *c64[:4], *c64[4:8], c64[8], c64[9], _np64(ll),
                                                nthreads=8)
    for q, k in enumerate(rates.keys()):
        x, r = _np64(rates[k]), orates[q]
        scale = np.maximum(np.abs(r), orates[6] if k in ("QCSHD", "QRSHD", "int_wet_M_col") else 0)
        err = np.abs(x - r) / np.maximum(scale, 1e-300)
        err[(r == 0) & (x == 0)] = 0
        assert err.max() <= RTOL[ft], (quad, k, float(err.max()))


@pytest.mark.parametrize("ft", ["f64", "f32"])
def test_p3_het_ice_nucleation(dev, oracle, ft):
    """cmx_p3_het_ice_nucleation_*: the reference's six known answers (test/p3_tests.jl:590-612) and random-state parity."""
    import cmx
    from test_p3_collisions_oracle import _het_freezing_inputs
    g, tps64, ql, Nl, RH, T, rho = _het_freezing_inputs()
    tps, dust = P.ThermodynamicsParameters(ft), P.Illite(ft)
    col = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DT[ft]).to(dev)  # noqa: E731
    r = cmx.p3_het_ice_nucleation(dust, tps, col(ql), col(Nl), col(RH), col(T), col(rho))
    tol = 1e-9 if ft == "f64" else g["rtol_reference"]       # exp2(m·Δa_w·log2 10) with m ≈ 54: Float32 carries ≈1e-4 of J
    np.testing.assert_allclose(_np64(r.dNdt), g["dNdt"], rtol=tol)
    np.testing.assert_allclose(_np64(r.dLdt), g["dLdt"], rtol=tol)
    rng = np.random.default_rng(4)
    n = 100_000
    T = rng.uniform(200, 272, n); RHr = rng.uniform(0.5, 1.15, n); rho = rng.uniform(0.3, 1.3, n)
    ql = 10 ** rng.uniform(-7, -3, n); Nl = 10 ** rng.uniform(6, 9, n)
    cols32 = [col(a) for a in (ql, Nl, RHr, T, rho)]
    got = cmx.p3_het_ice_nucleation(dust, tps, *cols32)
    c64 = [_np64(c) for c in cols32]
    dN, dL = oracle.p3_het_ice_nucleation(_abi.F64, P.Illite("f64"), tps64, *c64)
    ok = np.isfinite(dN) & (dN < (1e30 if ft == "f32" else 1e300)) & (dL < (1e30 if ft == "f32" else 1e300))
    rt = 1e-9 if ft == "f64" else 2e-3
    for x, r_ in ((_np64(got.dNdt), dN), (_np64(got.dLdt), dL)):
        live = ok & (r_ > (1e-30 if ft == "f32" else 1e-290))
        assert np.all(np.abs(x[live] - r_[live]) <= rt * r_[live]) and np.all(x >= 0)
    parity.record(f"P3 het_ice_nucleation {ft}", ft, {"dNdt": _np64(got.dNdt), "dLdt": _np64(got.dLdt)}, {"dNdt": dN, "dLdt": dL}, family="P3 processes (f2)",
                  pinned_by="oracle restatement of src/P3_processes.jl:20-46 + the reference's KATs (test/p3_tests.jl:572-613)", keep=ok,
                  note="J = 10^(m Δa_w + c): the Float32 rounding of RH − a_w_ice is amplified by m ln 10 ≈ 125")


@pytest.mark.parametrize("ft", ["f64", "f32"])
@pytest.mark.parametrize("psd", ["cloud", "rain_limited", "rain_notlimited"])
def test_liquid_freezing_rate(dev, oracle, ft, psd):
    """cmx_liquid_freezing_rate_*: Bigg freezing over the cloud / rain PSD against the oracle (test/gpu_tests.jl:1072-1091 and
    test/heterogeneous_ice_nucleation_tests.jl:430-480: colder ⇒ larger, −4 °C gate, zero N or q ⇒ zero)."""
    import cmx
    limited = psd != "rain_notlimited"
    cloud = psd == "cloud"
    ip, tps = P.P3IceParams(ft, is_limited=limited), P.ThermodynamicsParameters(ft)
    rng = np.random.default_rng(8)
    n = 50_000
    T = rng.uniform(235, 275, n); rho = rng.uniform(0.3, 1.3, n)
    q = np.where(rng.random(n) < 0.9, 10 ** rng.uniform(-7, -3, n), 0.0)
    N = np.where(rng.random(n) < 0.95, 10 ** (rng.uniform(6, 9, n) if cloud else rng.uniform(1, 6, n)), 0.0)
    cols = [torch.from_numpy(a).to(DT[ft]).to(dev) for a in (q, rho, N, T)]
    got = cmx.liquid_freezing_rate(ip, tps, *cols, cloud=cloud)
    c64 = [_np64(c) for c in cols]
    ip64 = P.P3IceParams("f64", is_limited=limited)
    dn, dq = oracle.liquid_freezing_rate(_abi.F64, ip64.c.rain_freezing, ip64.c.cloud_pdf if cloud else ip64.c.rain_pdf,
                                         P.ThermodynamicsParameters("f64"), *c64, cloud=cloud, limited=limited, float32_gates=(ft == "f32"))
    for x, r in ((_np64(got.dn_frz), dn), (_np64(got.dq_frz), dq)):
        assert np.array_equal(x == 0, r == 0)
        live = (r > (1e-30 if ft == "f32" else 1e-290)) & (r < (1e30 if ft == "f32" else 1e290))
        assert np.all(np.abs(x[live] - r[live]) <= RTOL[ft] * r[live]) and live.mean() > 0.3
    parity.record(f"liquid_freezing_rate {ft} {psd}", ft, {"dn_frz": _np64(got.dn_frz), "dq_frz": _np64(got.dq_frz)}, {"dn_frz": dn, "dq_frz": dq},
                  family="P3 processes (f2)", pinned_by="oracle restatement of src/IceNucleation.jl:274-389 + test/gpu_tests.jl:1072-1091", assert_wellcond=True)
    warm = c64[3] >= tps.T_freeze - 4
    assert warm.any() and np.all(_np64(got.dn_frz)[warm] == 0)


@pytest.mark.parametrize("ft", ["f64", "f32"])
def test_million_states_identities(dev, ft):
    """Size-independent properties at 1e6 random mixed-phase states (no oracle at this size): the integrand identities of
    ∫liquid_ice_collisions (src/P3_processes.jl:466-486) — freeze + shed = collected mass for cloud and rain together, wet ≤ total,
    all ten integrals ≥ 0 and finite — and the bulk-source bookkeeping (:640-650)."""
    import cmx
    n = 1_000_000
    cols = [c.to(dev) for c in _random_states(n, ft, seed=101)]
    ip = P.P3IceParams(ft)
    aps, tps = P.AirProperties(ft), P.ThermodynamicsParameters(ft)
    ll = cmx.p3_shape(P.ParametersP3(ft), *cols[:4], from_state=True, want=("log_lambda",)).log_lambda
    ll = torch.where(torch.isfinite(ll), ll, torch.zeros_like(ll))
    src, r = cmx.p3_liquid_ice_collisions(ip, aps, tps, cols[8], cols[9], *cols[:4], ll, *cols[4:8], from_state=True, want_rates=True)
    torch.cuda.synchronize()
    for k, v in r.items():
        assert bool(torch.isfinite(v).all()) and float(v.min()) >= 0.0, k
    tot = r["int_M_col"]
    parts = r["QCFRZ"] + r["QCSHD"] + r["QRFRZ"] + r["QRSHD"]
    tol = 1e-12 if ft == "f64" else 2e-5
    tiny = 1e-300 if ft == "f64" else 1e-35
    assert float(((parts - tot).abs() / tot.clamp(min=tiny)).max()) <= tol
    assert bool((r["int_wet_M_col"] <= tot * (1 + tol)).all())
    rho = cols[8]
    assert float(((src["dq_c"] + (r["QCFRZ"] + r["QCSHD"]) / rho).abs() / ((r["QCFRZ"] + r["QCSHD"]) / rho).clamp(min=tiny)).max()) <= tol
    assert float(((src["dL_ice"] - (r["QCFRZ"] + r["QRFRZ"])).abs() / (r["QCFRZ"] + r["QRFRZ"]).clamp(min=tiny)).max()) <= tol
    assert bool((src["dL_rim"] >= src["dL_ice"] * (1 - tol)).all()) and float((tot > 0).float().mean()) > 0.5
