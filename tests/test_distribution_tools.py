"""DistributionTools and the SB2006 size-distribution accessors (VERDICT r03 "missing" 5): cmx_generalized_gamma_*, cmx_exponential_distribution_*,
cmx_sb2006_size_distribution_*.

CPU: the oracle restatement (oracle/cmx_oracle_dist_impl.h) against the reference's own tests — the quantile ↔ cdf correspondences of
test/DistributionTools_tests.jl:11-47 at its rtol 1e-10, its edge cases, the limiting behaviour of test/microphysics2M_tests.jl:142-163 and the
probability levels of :598-607 — and against scipy's closed forms.
GPU (-m gpu): the device entries against the oracle, both float types."""
import numpy as np
import pytest
import scipy.special as sp
import torch

import parity
from cmx import _abi
from cmx import parameters as P

F64, F32 = _abi.F64, _abi.F32
DT = {"f32": torch.float32, "f64": torch.float64}


# ---- oracle vs the reference's tests ---------------------------------------------------------------------------------------------------------
def test_generalized_gamma_quantile_cdf_correspondence(oracle):
    nu, mu, B = 2.0, 3.0, 2.0                                          # test/DistributionTools_tests.jl:6-9
    Y = np.array([0.1, 0.25, 0.5, 0.75, 0.9])
    x, _ = oracle.generalized_gamma(F64, nu, mu, np.full(5, B), Y=Y)
    _, p = oracle.generalized_gamma(F64, nu, mu, np.full(5, B), x=x)
    # the reference asserts rtol 1e-10 on its Halley inverse of its 30-term series: the restatement must meet the same
    np.testing.assert_allclose(p, Y, rtol=1e-10)
    np.testing.assert_allclose(x, (sp.gammaincinv((nu + 1) / mu, Y) / B) ** (1 / mu), rtol=1e-9)
    _, z = oracle.generalized_gamma(F64, nu, mu, np.full(2, B), x=np.array([0.0, -1.0]))
    assert z.tolist() == [0.0, 0.0]                                    # :19-20
    _, bad = oracle.generalized_gamma(F64, nu, -1.0, np.array([B]), x=np.array([1.0]))      # DomainError in the reference (:23-24) → NaN
    _, bad2 = oracle.generalized_gamma(F64, nu, mu, np.array([-1.0]), x=np.array([1.0]))
    assert np.isnan(bad[0]) and np.isnan(bad2[0])


def test_exponential_quantile_cdf_correspondence(oracle):
    D_mean = 2.0                                                       # test/DistributionTools_tests.jl:29
    Y = np.array([0.1, 0.25, 0.5, 0.75, 0.9])
    D, _ = oracle.exponential_distribution(F64, np.full(5, D_mean), Y=Y)
    _, p = oracle.exponential_distribution(F64, np.full(5, D_mean), D=D)
    np.testing.assert_allclose(p, Y, rtol=1e-10)
    np.testing.assert_allclose(D, -D_mean * np.log1p(-Y), rtol=1e-14)
    _, e = oracle.exponential_distribution(F64, np.full(3, D_mean), D=np.array([0.0, -1.0, np.inf]))
    assert e[0] == 0 and e[1] == 0 and e[2] == pytest.approx(1.0, rel=1e-10)        # :39-41
    q, c = oracle.exponential_distribution(F64, np.array([-1.0, D_mean, D_mean, -1.0]), Y=np.array([0.5, -0.1, 1.1, 0.5]), D=np.array([1.0, 1.0, 1.0, 1.0]))
    assert np.isnan(c[0]) and np.all(np.isnan(q[[1, 2, 3]]))                            # the four DomainErrors of :44-47


@pytest.mark.parametrize("limited", [True, False])
def test_sb2006_psd_limits_and_probability_levels(oracle, limited):
    sb = P.SB2006("f64", limited)
    zero = np.zeros(1)
    r = oracle.sb2006_size_distribution(F64, None, sb.pdf_r, zero, np.array([1.2]), zero, D=np.array([0.1]), limited=limited)
    assert r["n_D"][0] == 0 and r["D_min"][0] == 0 and r["D_max"][0] == 0          # test/microphysics2M_tests.jl:147-160
    c = oracle.sb2006_size_distribution(F64, sb.pdf_c, None, zero, np.array([1.2]), zero, D=np.array([1e-5]), cloud=True)
    assert c["n_D"][0] == 0                                                        # logN₀c = −Inf
    # probability levels (:598-607): the bounds ARE the p and 1 − p quantiles of the exponential PSD
    q, rho, N, p = np.array([1e-4, 1e-3]), np.array([1.1, 0.9]), np.array([1e4, 5e4]), 1e-6
    b = oracle.sb2006_size_distribution(F64, None, sb.pdf_r, q, rho, N, limited=limited, p=p)
    pr = [oracle.pdf_rain_parameters(F64, sb.pdf_r, limited, float(a), float(bb), float(cc)) for a, bb, cc in zip(q, rho, N)]
    Dm = np.array([x["Dr_mean"] for x in pr])
    _, lo = oracle.exponential_distribution(F64, Dm, D=b["D_min"])
    _, hi = oracle.exponential_distribution(F64, Dm, D=b["D_max"])
    np.testing.assert_allclose(lo, p, rtol=1e-9)
    np.testing.assert_allclose(hi, 1 - p, rtol=1e-9)
    # the cloud PSD integrates to N between its bounds (a generalized gamma in D): cdf(D_max) − cdf(D_min) = 1 − 2p
    cb = oracle.sb2006_size_distribution(F64, sb.pdf_c, None, np.array([5e-4]), np.array([1.0]), np.array([1e8]), cloud=True, p=1e-6)
    assert 0 < cb["D_min"][0] < 2e-5 < cb["D_max"][0] < 2e-4


def _chebyshev_gauss(f, a, b, n):
    """P3.integrate(f, a, b, ChebyshevGauss(n)) — src/Quadrature.jl:168-175"""
    y = np.cos(np.pi * (2 * np.arange(1, n + 1) - 1) / (2 * n))
    return float(np.sum(np.sqrt(1 - y * y) * np.pi / n * f(0.5 * (b - a) * y + 0.5 * (b + a))) * 0.5 * (b - a))


@pytest.mark.parametrize("limited", [True, False])
def test_sb2006_rain_psd_integrates_to_number_and_mass(oracle, limited):
    """test/microphysics2M_tests.jl:569-625: the rain PSD between its p = 1e-6 bounds integrates to N_r (ChebyshevGauss(1000), rtol 1e-6) and its third
    moment to q_r (ChebyshevGauss(100), rtol 6e-4); the moment identities of :627-651 through the exponential-PSD closed forms."""
    sb = P.SB2006(P.create_toml_dict("f64", P.SB2006_LIMITERS_OVERRIDE), limited)      # the override file of the reference's CPU tests (:26-31)
    rho, N, q, p = 1.2, 0.5e6, 0.5e-3, 1e-6
    one = lambda v: np.array([v])  # noqa: E731
    b = oracle.sb2006_size_distribution(F64, None, sb.pdf_r, one(q), one(rho), one(N), limited=limited, p=p)
    D_min, D_max = float(b["D_min"][0]), float(b["D_max"][0])

    def psd(D):
        D = np.atleast_1d(D)
        return oracle.sb2006_size_distribution(F64, None, sb.pdf_r, np.full_like(D, q), np.full_like(D, rho), np.full_like(D, N), D=D, limited=limited, bounds=False)["n_D"]
    pr = oracle.pdf_rain_parameters(F64, sb.pdf_r, limited, q, rho, N)
    f_D = lambda D: pr["N0r"] * np.exp(-D / pr["Dr_mean"])  # noqa: E731 — eq. (3) of the 2M docs, written by hand like the reference's test
    ND, ND_psd = _chebyshev_gauss(f_D, D_min, D_max, 1000), _chebyshev_gauss(psd, D_min, D_max, 1000)
    assert ND == pytest.approx(N, rel=1e-6) and ND_psd == pytest.approx(ND, rel=1e-13)
    k_m = np.pi * float(sb.pdf_r.rho_w) / 6
    qD = _chebyshev_gauss(lambda D: D ** 3 * f_D(D), D_min, D_max, 100) * k_m / rho
    qD_psd = _chebyshev_gauss(lambda D: D ** 3 * psd(D), D_min, D_max, 100) * k_m / rho
    assert qD == pytest.approx(q, rel=6e-4) and qD_psd == pytest.approx(qD, rel=1e-13)
    # exponential moments: M⁰ = N, k_m M³ = L (DT.exponential_Mⁿ = N D̄ⁿ Γ(n + 1))
    assert pr["N0r"] * pr["Dr_mean"] == pytest.approx(N, rel=1e-12)
    assert k_m * N * pr["Dr_mean"] ** 3 * 6 == pytest.approx(q * rho, rel=1e-12)


def test_sb2006_cloud_psd_integrates_to_number_and_mass(oracle):
    """test/microphysics2M_tests.jl:657-719: the cloud PSD (a generalized gamma in D) between its p = 1e-6 bounds integrates to N_l and its third moment to
    q_l with ChebyshevGauss(100), at the reference's tolerances 1e-5 and 2e-5."""
    sb = P.SB2006(P.create_toml_dict("f64", P.SB2006_LIMITERS_OVERRIDE), True)
    rho, N, q, p = 1.2, 1e9, 1e-3, 1e-6
    one = lambda v: np.array([v])  # noqa: E731
    b = oracle.sb2006_size_distribution(F64, sb.pdf_c, None, one(q), one(rho), one(N), cloud=True, p=p)
    D_min, D_max = float(b["D_min"][0]), float(b["D_max"][0])

    def psd(D):
        D = np.atleast_1d(D)
        return oracle.sb2006_size_distribution(F64, sb.pdf_c, None, np.full_like(D, q), np.full_like(D, rho), np.full_like(D, N), D=D, cloud=True, bounds=False)["n_D"]
    k_m = np.pi * float(sb.pdf_c.rho_w) / 6
    assert _chebyshev_gauss(psd, D_min, D_max, 100) == pytest.approx(N, rel=1e-5)
    assert _chebyshev_gauss(lambda D: D ** 3 * psd(D), D_min, D_max, 100) * k_m / rho == pytest.approx(q, rel=2e-5)


# ---- device vs oracle --------------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.gpu
@pytest.mark.parametrize("ft", ["f64", "f32"])
def test_device_distribution_tools(dev, oracle, ft):
    import cmx
    fam = _abi.family(ft)
    npf = np.float32 if ft == "f32" else np.float64
    rng = np.random.default_rng(3)
    n = 20_000
    rd = lambda a: np.asarray(a).astype(npf).astype(np.float64)  # noqa: E731
    to = lambda a: torch.from_numpy(a).to(DT[ft]).to(dev)  # noqa: E731
    back = lambda t: t.cpu().numpy().astype(np.float64)  # noqa: E731
    tol = parity.RTOL[ft]
    # the reference's own case first (ν, μ, B) = (2, 3, 2), then the cloud-PSD-like parameters (ν_D = 5, μ_D = 3, large B)
    for nu, mu, Bs in ((2.0, 3.0, rd(np.full(n, 2.0))), (5.0, 3.0, rd(10 ** rng.uniform(12, 16, n))), (-2.0 / 3.0, 1.0 / 3.0, rd(10 ** rng.uniform(2, 4, n)))):
        Y = rd(np.concatenate([rng.uniform(0.01, 0.99, n - 4), [0.1, 0.5, 0.9, 0.999]]))
        got = cmx.generalized_gamma(nu, mu, to(Bs), Y=to(Y))
        ref_q, _ = oracle.generalized_gamma(F64, nu, mu, Bs, Y=Y)
        np.testing.assert_allclose(back(got.quantile), ref_q, rtol=tol * (5 if ft == "f32" else 20))       # Halley's last step at eps·x, then ^(1/μ)
        x = ref_q * rd(rng.uniform(0.5, 1.5, n))
        gc = cmx.generalized_gamma(nu, mu, to(Bs), x=to(x))
        _, ref_c = oracle.generalized_gamma(F64, nu, mu, Bs, x=rd(x))
        assert np.max(np.abs(back(gc.cdf) - ref_c)) <= (2e-5 if ft == "f32" else 1e-9)                       # P is compared absolutely, like UT.gamma_inc itself
        parity.record(f"DT.generalized_gamma ν={nu:.2f} μ={mu:.2f} {ft}", ft, {"quantile": back(got.quantile), "cdf": back(gc.cdf)}, {"quantile": ref_q, "cdf": ref_c},
                      family="row g: DistributionTools / PSD accessors", pinned_by="oracle restatement of src/DistributionTools.jl:44-82 + test/DistributionTools_tests.jl",
                      scale={"cdf": np.ones(n)})
    Dm = rd(10 ** rng.uniform(-4, -2, n))
    Y = rd(np.concatenate([10 ** rng.uniform(-7, -0.01, n // 2), 1 - 10 ** rng.uniform(-6 if ft == "f64" else -4, -0.01, n - n // 2)]))
    ge = cmx.exponential_distribution(to(Dm), Y=to(Y))
    rq, _ = oracle.exponential_distribution(F64, Dm, Y=Y)
    np.testing.assert_allclose(back(ge.quantile), rq, rtol=tol)
    D = rq * rd(rng.uniform(0.2, 3.0, n))
    gc = cmx.exponential_distribution(to(Dm), D=to(D))
    _, rc = oracle.exponential_distribution(F64, Dm, D=rd(D))
    np.testing.assert_allclose(back(gc.cdf), rc, rtol=tol)
    parity.record(f"DT.exponential {ft}", ft, {"quantile": back(ge.quantile), "cdf": back(gc.cdf)}, {"quantile": rq, "cdf": rc},
                  family="row g: DistributionTools / PSD accessors", pinned_by="oracle restatement of src/DistributionTools.jl:124-151 + test/DistributionTools_tests.jl", assert_wellcond=True)
    # DomainErrors → NaN, edge values
    e = cmx.exponential_distribution(to(np.array([2.0, 2.0, -1.0])), Y=to(np.array([-0.1, 1.1, 0.5])), D=to(np.array([0.0, -1.0, 1.0])))
    assert torch.isnan(e.quantile).all() and e.cdf[:2].tolist() == [0.0, 0.0] and bool(torch.isnan(e.cdf[2]))
    g = cmx.generalized_gamma(2.0, -1.0, to(np.array([2.0])), x=to(np.array([1.0])))
    assert bool(torch.isnan(g.cdf[0]))


@pytest.mark.gpu
@pytest.mark.parametrize("ft", ["f64", "f32"])
@pytest.mark.parametrize("which", ["rain_limited", "rain_notlimited", "cloud"])
def test_device_sb2006_size_distribution(dev, oracle, ft, which):
    import cmx
    npf = np.float32 if ft == "f32" else np.float64
    rng = np.random.default_rng(9)
    n = 20_000
    rd = lambda a: np.asarray(a).astype(npf).astype(np.float64)  # noqa: E731
    to = lambda a: torch.from_numpy(a).to(DT[ft]).to(dev)  # noqa: E731
    back = lambda t: t.cpu().numpy().astype(np.float64)  # noqa: E731
    cloud, limited = which == "cloud", which != "rain_notlimited"
    sb, sb64 = P.SB2006(ft, limited), P.SB2006("f64", limited)
    q = rd(np.where(rng.random(n) < 0.9, 10 ** rng.uniform(-7, -3, n), 0.0))
    N = rd(np.where(rng.random(n) < 0.9, 10 ** (rng.uniform(6, 9, n) if cloud else rng.uniform(1, 6, n)), 0.0))
    rho = rd(rng.uniform(0.3, 1.3, n))
    D = rd(10 ** (rng.uniform(-6, -4, n) if cloud else rng.uniform(-4.5, -2, n)))
    p = 1e-6
    pdf, pdf64 = (sb.pdf_c, sb64.pdf_c) if cloud else (sb.pdf_r, sb64.pdf_r)
    # the rain PSD variant is read from the struct (the reference dispatches on its type): no is_limited argument
    got = cmx.size_distribution(pdf, to(q), to(rho), to(N), to(D), p=p)
    if not cloud:
        again = cmx.size_distribution(pdf, to(q), to(rho), to(N), to(D), p=p, is_limited=limited)
        assert all(torch.equal(a, b) for a, b in zip(got, again))
        if not limited:
            with pytest.raises(ValueError):
                cmx.size_distribution(pdf, to(q), to(rho), to(N), to(D), p=p, is_limited=True)
    ref = oracle.sb2006_size_distribution(_abi.F64, pdf64 if cloud else None, None if cloud else pdf64, q, rho, N, D=D, cloud=cloud, limited=limited, p=p,
                                          float32_gates=(ft == "f32"))
    tol = parity.RTOL[ft]
    nD, rn = back(got.n_D), ref["n_D"]
    tiny = 1e-36 if ft == "f32" else 0.0                      # below the Float32 range the Float64 oracle's value is the device's zero
    assert np.array_equal(nD[(rn == 0) | (rn > tiny)] == 0, rn[(rn == 0) | (rn > tiny)] == 0)
    assert np.all(nD[(rn > 0) & (rn <= tiny)] <= 2 * tiny)
    live = (rn > (1e-30 if ft == "f32" else 1e-290)) & (rn < (1e30 if ft == "f32" else 1e300))
    # n(D) = exp(log N₀ + ν log D − λ D^μ): the exponent (size up to 100) carries the Float32 rounding of its terms
    lam_term = np.abs(np.log(np.maximum(rn, 1e-300))) + 50
    assert np.all(np.abs(nD[live] - rn[live]) <= tol * rn[live] * (1 + (lam_term[live] * 2e-5 / tol if ft == "f32" else 0)))
    for k in ("D_min", "D_max"):
        x, r = back(getattr(got, k)), ref[k]
        assert np.array_equal(x == 0, r == 0) or cloud
        ok = np.isfinite(r) & (r > 0)
        assert np.all(np.abs(x[ok] - r[ok]) <= tol * (5 if cloud else 1) * r[ok]), k
    parity.record(f"CM2.size_distribution {which} {ft}", ft, {"D_min": back(got.D_min), "D_max": back(got.D_max)}, {"D_min": ref["D_min"], "D_max": ref["D_max"]},
                  family="row g: DistributionTools / PSD accessors", pinned_by="oracle restatement of src/Microphysics2M.jl:270-354 + test/microphysics2M_tests.jl:142-163,598-607",
                  keep=np.isfinite(ref["D_min"]) & np.isfinite(ref["D_max"]))
