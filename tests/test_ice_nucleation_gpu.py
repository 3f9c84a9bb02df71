"""GPU parity tests of the fused ice-nucleation kernel (ABIFM + Koop 2000 + water activities) through the
C ABI: reference KATs, random-state parity against the oracle, the DomainError → NaN + count path, ragged /
unaligned inputs, and BASELINE config 4's full size (1e8 f32 points) through size-independent properties."""
import json
import math
from pathlib import Path

import numpy as np
import pytest
import torch

import parity
from cmx import _abi
from cmx import parameters as P

pytestmark = pytest.mark.gpu
DT = {"f32": torch.float32, "f64": torch.float64}
G = json.loads((Path(__file__).parent / "golden" / "ice_nucleation_kats.json").read_text())
ALL = ("delta_a_w", "J_het", "J_hom", "rate_het", "rate_hom")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _params(ft, dust="Kaolinite"):
    return P.ThermodynamicsParameters(ft), getattr(P, dust)(ft), P.Koop2000(ft)


def _edge_mask(delta_ref, koop, ft):
    tol = 1e-6 if ft == "f32" else 1e-13
    return (np.abs(delta_ref - koop.delta_a_w_min) < tol) | (np.abs(delta_ref - koop.delta_a_w_max) < tol)


def _compare(got, ref, ft, koop64, what):
    edge = _edge_mask(ref["delta_a_w"], koop64, ft)
    rtol = parity.RTOL[ft]
    rep = {}
    for k in ALL:
        if got.get(k) is None:
            continue
        g, r = got[k].astype(np.float64), ref[k]
        scale = None
        if k == "delta_a_w":
            scale = np.full_like(r, 1.0)          # Δa_w = a_w − a_w_ice: two O(1) terms
        e = parity.scaled_err(g, r, scale, parity.FLOOR[ft], parity.CEIL[ft], parity.CTOL[ft] / parity.RTOL[ft])
        e = np.nan_to_num(e, nan=np.inf)
        if k in ("J_hom", "rate_hom"):
            e = e[~edge]
        rep[k] = float(e.max()) if e.size else 0.0
        assert rep[k] <= rtol, (what, k, rep[k])
    have = {k: got[k] for k in ALL if got.get(k) is not None}
    pin = "oracle restatement (src/IceNucleation.jl:124-134,557-584, src/Common.jl:250-271) + the reference's KATs"
    parity.record("ice nucleation " + what, ft, have, ref, family="ice nucleation (a4)", pinned_by=pin, names=[k for k in have if k not in ("J_hom", "rate_hom")],
                  scale={"delta_a_w": np.ones_like(ref["delta_a_w"])})
    parity.record("ice nucleation " + what, ft, have, ref, family="ice nucleation (a4)", pinned_by=pin, names=[k for k in have if k in ("J_hom", "rate_hom")],
                  keep=~edge, note="points within rounding of the Koop Δa_w window edges are set aside (domain-error branch)")
    return rep, int(edge.sum())


@pytest.mark.parametrize("ft", ["f64", "f32"])
def test_kats_through_the_abi(dev, ft):
    import cmx
    tps, _, koop = _params(ft)
    tol = 1e-9 if ft == "f64" else 2e-5
    col = lambda v: torch.full((10,), v, dtype=DT[ft], device=dev)  # noqa: E731
    for e in G["a_w_ice"]:
        assert math.isclose(cmx.a_w_ice(tps, col(e["T"]))[0].item(), e["expected"], rel_tol=max(tol, e["rtol"]))
    for e in G["a_w_eT"]:
        assert math.isclose(cmx.a_w_eT(tps, col(e["e"]), col(e["T"]))[0].item(), e["expected"], rel_tol=max(tol, e["rtol"]))
    T = 220.0
    ice = cmx.a_w_ice(P.ThermodynamicsParameters("f64"), torch.full((1,), T, dtype=torch.float64, device=dev))[0].item()
    for e in G["ABIFM_J"]:
        dust = getattr(P, e["dust"])(ft)
        r = cmx.ice_nucleation_rates(tps, dust, koop, col(T), col(ice + e["delta_a_w"]), want=("J_het", "delta_a_w"))
        # Float32: a_w itself is rounded to 6e-8, and d log J / dΔ = m ln10 ≈ 126
        assert math.isclose(r.J_het[0].item(), e["expected"], rel_tol=1e-9 if ft == "f64" else 5e-5)
    h = G["homogeneous_J"]
    a_w = col(ice + h["delta_a_w"])
    rc = cmx.ice_nucleation_rates(tps, P.Kaolinite(ft), koop, col(T), a_w, want=("J_hom",))
    rl = cmx.ice_nucleation_rates(tps, P.Kaolinite(ft), koop, col(T), a_w, want=("J_hom",), linear=True)
    assert math.isclose(rc.J_hom[0].item(), h["J_cubic"], rel_tol=1e-9 if ft == "f64" else 2e-4)
    assert math.isclose(rl.J_hom[0].item(), h["J_linear"], rel_tol=2e-7 if ft == "f64" else 2e-4)
    assert cmx.domain_error_count(rc) == 0 and rl.n_domain_errors is None
    d = G["homogeneous_J_cubic_domain"]
    bad = cmx.ice_nucleation_rates(tps, P.Kaolinite(ft), koop, col(T), torch.tensor(
        [ice + d["too_small"], ice + d["too_large"]] * 5, dtype=DT[ft], device=dev), want=("J_hom", "J_het"))
    assert torch.isnan(bad.J_hom).all() and torch.isfinite(bad.J_het).all() and cmx.domain_error_count(bad) == 10


@pytest.mark.parametrize("ft", ["f32", "f64"])
@pytest.mark.parametrize("linear", [False, True])
@pytest.mark.parametrize("dust", ["Kaolinite", "Illite"])
def test_random_state_parity(dev, oracle, ft, linear, dust):
    import cmx
    from cmx import synthetic
    n = 1_000_003
    st = synthetic.ice_nucleation_state(n, dtype=DT[ft], seed=42)
    tps, du, koop = _params(ft, dust)
    r = cmx.ice_nucleation_rates(tps, du, koop, *[c.to(dev) for c in st], want=ALL, linear=linear)
    torch.cuda.synchronize()
    got = {k: getattr(r, k).cpu().numpy() for k in ALL}
    t64, d64, k64 = _params("f64", dust)
    ref = oracle.ice_nucleation_rates(_abi.F64, t64, d64, k64, _abi.CMX_ICENUC_HOM_LINEAR if linear else 0,
                                      *[c.numpy().astype(np.float64) for c in st])
    rep, nedge = _compare(got, ref, ft, k64, f"{ft} {dust} linear={linear}")
    print(f"\n[icenuc parity] {ft} {dust} linear={linear}: {rep}, edge points {nedge}")
    if not linear:
        assert abs(cmx.domain_error_count(r) - ref["n_domain_errors"]) <= nedge
        assert ref["n_domain_errors"] > 0.03 * n          # the synthetic state does exercise the error path
        nan_ref = np.isnan(ref["J_hom"])
        nan_got = np.isnan(got["J_hom"])
        assert (nan_ref != nan_got).sum() <= nedge


@pytest.mark.parametrize("ft", ["f32", "f64"])
@pytest.mark.parametrize("n", [0, 1, 3, 5, 257, 1023])
def test_ragged_sizes(dev, oracle, ft, n):
    import cmx
    from cmx import synthetic
    st = [c[:n].contiguous() for c in synthetic.ice_nucleation_state(max(n, 1), dtype=DT[ft], seed=n)]
    tps, du, koop = _params(ft)
    r = cmx.ice_nucleation_rates(tps, du, koop, *[c.to(dev) for c in st], want=ALL)
    assert r.rate_het.shape == (n,)
    if n:
        ref = oracle.ice_nucleation_rates(_abi.F64, *_params("f64"), 0, *[c.numpy().astype(np.float64) for c in st])
        _compare({k: getattr(r, k).cpu().numpy() for k in ALL}, ref, ft, P.Koop2000("f64"), f"n={n}")


def test_unaligned_and_optional_columns(dev):
    import cmx
    from cmx import synthetic
    st = [c.to(dev) for c in synthetic.ice_nucleation_state(10_002, seed=9)]
    tps, du, koop = _params("f32")
    a = cmx.ice_nucleation_rates(tps, du, koop, *[c[1:] for c in st], want=ALL)
    b = cmx.ice_nucleation_rates(tps, du, koop, *[c[1:].clone() for c in st], want=ALL)
    for k in ALL:
        x, y = getattr(a, k), getattr(b, k)
        assert torch.equal(torch.nan_to_num(x, nan=-1.0), torch.nan_to_num(y, nan=-1.0)), k
    assert cmx.domain_error_count(a) == cmx.domain_error_count(b)
    only = cmx.ice_nucleation_rates(tps, du, koop, st[0], st[1], want=("J_het",))      # no radius column needed
    assert only.rate_het is None and torch.equal(only.J_het[1:], a.J_het)
    with pytest.raises(ValueError):
        cmx.ice_nucleation_rates(tps, du, koop, st[0], st[1])                          # rates need r
    with pytest.raises(TypeError):
        cmx.ice_nucleation_rates(P.ThermodynamicsParameters("f64"), du, koop, *st)


def test_full_size_1e8_f32_properties(dev, oracle):
    """BASELINE config 4: 1e8 (T, a_w, r) Float32 points."""
    import cmx
    from cmx import sharding, synthetic
    n = 100_000_000
    st = synthetic.ice_nucleation_state(n, dtype=torch.float32, device=dev, seed=1234)
    tps, du, koop = _params("f32")
    full = cmx.ice_nucleation_rates(tps, du, koop, *st)
    torch.cuda.synchronize()
    nan = torch.isnan(full.rate_hom)
    assert int(nan.sum()) == cmx.domain_error_count(full)                       # count == number of NaN points
    assert 0.04 * n < cmx.domain_error_count(full) < 0.06 * n
    assert bool(torch.isfinite(full.rate_het).all()) and bool(torch.isfinite(full.rate_hom[~nan]).all())
    assert bool((full.rate_het > 0).all())
    # chunk invariance over the 8-rank shard layout + domain-error counts add up (checksum of checksums)
    total = 0
    for rk in range(8):
        lo, hi = sharding.shard_bounds(n, rk, 8)
        if rk in (0, 3, 7):
            part = cmx.ice_nucleation_rates(tps, du, koop, *[c[lo:hi] for c in st])
            assert torch.equal(part.rate_het, full.rate_het[lo:hi])
            assert torch.equal(torch.nan_to_num(part.rate_hom, nan=-1.0), torch.nan_to_num(full.rate_hom[lo:hi], nan=-1.0))
            total += cmx.domain_error_count(part)
        else:
            total += int(nan[lo:hi].sum())
    assert total == cmx.domain_error_count(full)
    # oracle on a strided 1e6-point sample of the same inputs
    stride = 101
    samp = [c[::stride].contiguous().cpu().numpy().astype(np.float64) for c in st]
    ref = oracle.ice_nucleation_rates(_abi.F64, *_params("f64"), 0, *samp)
    got = {"rate_het": full.rate_het[::stride].cpu().numpy(), "rate_hom": full.rate_hom[::stride].cpu().numpy()}
    rep, nedge = _compare(got, ref, "f32", P.Koop2000("f64"), "1e8 sample")
    print(f"\n[icenuc parity 1e8 f32, {samp[0].size} sampled points] {rep}, edge {nedge}")
