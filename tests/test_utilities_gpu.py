"""UT.gamma_inc / UT.gamma_inc_inv on the device (cmx_gamma_inc_*, cmx_gamma_inc_inv_*) — the reference's device unit test
`test_gamma_inc_kernel!` (test/gpu_tests.jl:456-461,1314-1337) and its CPU grid test (test/gamma_inc_tests.jl:29-49), at the reference's
own tolerances, plus the comparison with the oracle's fixed 20 / 30-term restatement where that truncation has NOT converged
(x ∈ [a, a + 1), a = 20 … 60: the device may stop early only once converged, so it must reproduce the truncated value there)."""
import itertools
import json
from pathlib import Path

import numpy as np
import pytest
import scipy.special as sp
import torch

import cmx
import parity
from cmx import _abi

pytestmark = pytest.mark.gpu
DT = {"f32": torch.float32, "f64": torch.float64}
NP = {"f32": np.float32, "f64": np.float64}
G = json.loads((Path(__file__).parent / "golden" / "p3_kats.json").read_text())["gamma_inc_reference_grid"]


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _dev(x, ft, dev):
    return torch.tensor(np.asarray(x, dtype=NP[ft]), device=dev)


@pytest.mark.parametrize("ft", ["f64", "f32"])
def test_reference_grid_at_the_reference_tolerances(dev, ft):
    """test/gamma_inc_tests.jl:39-49: every (a, x) and (a, p) pair of the 7 × 7 grids against SpecialFunctions (here scipy)."""
    a, x = map(np.array, zip(*itertools.product(G["a"], G["x"])))
    r = cmx.gamma_inc(_dev(a, ft, dev), _dev(x, ft, dev))
    P, Q = r.P.cpu().numpy().astype(np.float64), r.Q.cpu().numpy().astype(np.float64)
    atol = G[f"atol_PQ_{ft}"]
    assert np.max(np.abs(P - sp.gammainc(a, x))) <= atol and np.max(np.abs(Q - sp.gammaincc(a, x))) <= atol
    np.testing.assert_allclose(P + Q, 1.0, atol=4 * np.finfo(NP[ft]).eps)
    a, p = map(np.array, zip(*itertools.product(G["a"], G["p"])))
    xi = cmx.gamma_inc_inv(_dev(a, ft, dev), _dev(p, ft, dev), _dev(1 - p, ft, dev)).cpu().numpy().astype(np.float64)
    want = sp.gammaincinv(a, p)
    rtol = G[f"rtol_inv_{ft}"]
    assert np.all(np.abs(xi - want) <= rtol + rtol * np.abs(want))          # isapprox(…; rtol, atol = rtol)
    parity.record(f"UT.gamma_inc reference grid {ft}", ft, {"P": P, "Q": Q, "x_inv": xi},
                  {"P": sp.gammainc(*map(np.array, zip(*itertools.product(G['a'], G['x'])))),
                   "Q": sp.gammaincc(*map(np.array, zip(*itertools.product(G['a'], G['x'])))), "x_inv": want}, family="utilities", pinned_by="scipy (SpecialFunctions)")


@pytest.mark.parametrize("ft", ["f64", "f32"])
def test_device_pairs_like_the_reference_kernel(dev, ft):
    """test_gamma_inc_kernel!: element i gets (a[i], x[i], p[i], 1 − p[i])."""
    a, x, p = (np.array(G[k], dtype=np.float64) for k in ("a", "x", "p"))
    r = cmx.gamma_inc(_dev(a, ft, dev), _dev(x, ft, dev))
    xi = cmx.gamma_inc_inv(_dev(a, ft, dev), _dev(p, ft, dev), _dev(1 - p, ft, dev))
    np.testing.assert_allclose(r.P.cpu().numpy(), sp.gammainc(a, x), atol=G[f"atol_PQ_{ft}"], rtol=0)
    np.testing.assert_allclose(r.Q.cpu().numpy(), sp.gammaincc(a, x), atol=G[f"atol_PQ_{ft}"], rtol=0)
    np.testing.assert_allclose(xi.cpu().numpy(), sp.gammaincinv(a, p), atol=G[f"rtol_inv_{ft}"], rtol=G[f"rtol_inv_{ft}"])


@pytest.mark.parametrize("ft", ["f64", "f32"])
def test_against_the_oracle_truncation(dev, oracle, ft):
    """The oracle restates the reference's fixed 20 (Float32) / 30 (Float64) terms (src/Utilities.jl:93-144).  Random (a, x) over the range
    the P3 kernels use, and the band x ∈ [a, a + 1) ∪ (a + 1, a + 2) at a = 20 … 60 where 20 / 30 terms are not enough — there the
    reference's value is its truncation, and so must the device's be."""
    fam = _abi.family(ft)
    rng = np.random.default_rng(11)
    a = np.concatenate([rng.uniform(0.5, 12.0, 4000), rng.uniform(20.0, 60.0, 4000), rng.uniform(20.0, 60.0, 2000)])
    x = np.concatenate([a[:4000] * rng.uniform(0.01, 4.0, 4000), a[4000:8000] + rng.uniform(0.0, 1.0, 4000), a[8000:] + rng.uniform(1.0, 2.0, 2000)])
    a, x = a.astype(NP[ft]).astype(np.float64), x.astype(NP[ft]).astype(np.float64)
    r = cmx.gamma_inc(_dev(a, ft, dev), _dev(x, ft, dev))
    P, Q = r.P.cpu().numpy().astype(np.float64), r.Q.cpu().numpy().astype(np.float64)
    # the oracle in the kernel's own arithmetic (the truncation error depends on the term count: 20 vs 30)
    ref = np.array([oracle.gamma_inc(fam, float(ai), float(xi)) for ai, xi in zip(a, x)], dtype=np.float64)
    # the series / continued fraction are sums of positive terms (resp. a convergent): the device's reassociated arithmetic agrees with the
    # oracle's to rounding of the dominant one of (P, Q); the small one of the two is 1 − the other in BOTH codes, so it is compared
    # absolutely at the same number of ulps of 1
    tol = 2e-4 if ft == "f32" else 1e-12
    assert np.max(np.abs(P - ref[:, 0])) <= tol and np.max(np.abs(Q - ref[:, 1])) <= tol
    # … and the truncation IS visible in that band: the oracle (and the device with it) is off the true value by far more than `tol`
    band = slice(4000, 8000)
    trunc = np.abs(ref[band, 0] - sp.gammainc(a[band], x[band]))
    assert np.max(trunc) > 30 * tol            # Float32, 20 terms: up to 8e-3; Float64, 30 terms: up to 1e-4
    parity.record(f"UT.gamma_inc vs the 20/30-term truncation {ft}", ft, {"P": P, "Q": Q}, {"P": ref[:, 0], "Q": ref[:, 1]}, family="utilities",
                  pinned_by="oracle restatement of src/Utilities.jl:93-144", scale={"P": np.ones_like(P), "Q": np.ones_like(Q)})


@pytest.mark.parametrize("ft", ["f64", "f32"])
def test_inverse_round_trip_and_edge_values(dev, oracle, ft):
    fam = _abi.family(ft)
    rng = np.random.default_rng(5)
    a = rng.uniform(0.5, 12.0, 5000).astype(NP[ft]).astype(np.float64)
    p = np.concatenate([10.0 ** rng.uniform(-7, -0.31, 2500), 1 - 10.0 ** rng.uniform(-6 if ft == "f64" else -4, -0.31, 2500)]).astype(NP[ft]).astype(np.float64)
    q = (NP[ft](1) - p.astype(NP[ft])).astype(np.float64)
    xi = cmx.gamma_inc_inv(_dev(a, ft, dev), _dev(p, ft, dev), _dev(q, ft, dev)).cpu().numpy().astype(np.float64)
    ref = np.array([oracle.gamma_inc_inv(fam, float(ai), float(pi), float(qi)) for ai, pi, qi in zip(a, p, q)])
    rel = np.abs(xi - ref) / np.abs(ref)
    # Halley's last step is accepted at |step| < eps·x: two evaluations in different arithmetic agree to a few of those steps
    assert np.quantile(rel, 0.999) <= (2e-4 if ft == "f32" else 1e-9), np.max(rel)
    back = cmx.gamma_inc(_dev(a, ft, dev), _dev(xi, ft, dev)).P.cpu().numpy().astype(np.float64)
    small = p < 0.5
    assert np.max(np.abs(back[small] - p[small]) / p[small]) <= (5e-3 if ft == "f32" else 1e-8)
    # edge values of the reference: x ≤ 0 → (0, 1); x = Inf → (1, 0); p ≤ 0 → 0; q ≤ 0 → Inf
    e = cmx.gamma_inc(_dev([2.0, 2.0, 2.0], ft, dev), _dev([0.0, -1.0, np.inf], ft, dev))
    assert e.P.tolist() == [0.0, 0.0, 1.0] and e.Q.tolist() == [1.0, 1.0, 0.0]
    ei = cmx.gamma_inc_inv(_dev([2.0, 2.0], ft, dev), _dev([0.0, 1.0], ft, dev), _dev([1.0, 0.0], ft, dev))
    assert ei.tolist() == [0.0, np.inf]
    parity.record(f"UT.gamma_inc_inv random {ft}", ft, {"x_inv": xi}, {"x_inv": ref}, family="utilities", pinned_by="oracle restatement of src/Utilities.jl:205-252")
