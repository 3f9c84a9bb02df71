"""GPU parity tests of the fused 1-moment kernels through the C ABI: the reference's KATs, all 18 source terms and
the 4 tendencies against the oracle for several option sets, BASELINE config 1 (1e6 Float64 (ρ, q) points:
autoconversion + terminal velocities), ragged / unaligned inputs, and 1e8-point size-independent properties."""
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
G = json.loads((Path(__file__).parent / "golden" / "mp1m_kats.json").read_text())
TN = ["dq_lcl_dt", "dq_icl_dt", "dq_rai_dt", "dq_sno_dt"]
OPTION_SETS = {
    "default": {},
    "alt": dict(snow_autoconversion=P.WithSupersaturation(), snow_deposition_sublimation=P.SublimationOnly(),
                rain_autoconversion=P.PrescribedNd()),
    "sparse": dict(rain_snow_accretion=None, cloud_ice_melt=None, cloud_liquid_snow_accretion=None, snow_melt=None),
    # TemperatureDependent cloud-ice formation (Frostenberg 2023 INP timescale, NonEq:32-50,194-224): round 3
    "tdep": dict(cloud_ice_formation=P.TemperatureDependent()),
}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _oracle(oracle, ft, opts, cols_np, **kw):
    mp = P.Microphysics1MParams("f64", **opts)
    r = oracle.mp1m(_abi.F64, mp.c, P.ThermodynamicsParameters("f64"), mp.flags, *[c.astype(np.float64) for c in cols_np],
                    float32_gates=(ft == "f32"), nthreads=8, **kw)
    tf = P.DEFAULT_PARAMETERS["temperature_water_freeze"]
    r["near_branch"] = np.abs(cols_np[1].astype(np.float64) - tf) < (1e-4 if ft == "f32" else 1e-11)   # is_warm routing
    return r


@pytest.mark.parametrize("ft", ["f64", "f32"])
def test_kats_through_the_abi(dev, ft):
    import cmx
    mp, tps = P.Microphysics1MParams(ft), P.ThermodynamicsParameters(ft)
    g = G["accretion"]
    q, rho = g["inputs"]["q"], g["inputs"]["rho"]
    col = lambda v: torch.full((2,), v, dtype=DT[ft], device=dev)  # noqa: E731
    tol = g["rtol"] if ft == "f64" else 2e-5
    cold = cmx.microphysics_source_terms_1m(mp, tps, col(rho), col(260.0), col(5e-3), col(q), col(q), col(q), col(q))
    warm = cmx.microphysics_source_terms_1m(mp, tps, col(rho), col(290.0), col(5e-3), col(q), col(q), col(q), col(q))
    e = g["expected"]
    for got, exp in ((cold.S_accr_lcl_rai, e["liq_rai"]), (cold.S_accr_icl_sno, e["ice_sno"]), (cold.S_accr_lcl_sno_cold, e["liq_sno"]),
                     (warm.S_accr_lcl_sno_warm, e["liq_sno"]), (cold.S_accr_icl_rai, e["ice_rai"]),
                     (cold.S_accr_freeze_icl_rai, e["rai_sink"]), (cold.S_accr_rai_sno_cold, e["sno_rai"]),
                     (warm.S_accr_rai_sno_warm, e["rai_sno"])):
        assert math.isclose(got[0].item(), exp, rel_tol=tol), (got[0].item(), exp)
    assert cold.S_accr_melt_lcl_sno[0].item() == 0 and warm.S_accr_rai_sno_cold[0].item() == 0
    g = G["snow_melt"]
    for dT, q_sno, exp in g["cases"]:
        r = cmx.microphysics_source_terms_1m(mp, tps, col(g["rho"]), col(g["T_freeze"] + dT), col(0.0), col(0.0), col(0.0),
                                             col(0.0), col(q_sno))
        # Float32: T − T_freeze = 2 K carries the 3e-5 K rounding of both operands
        assert math.isclose(r.S_melt_sno_rai[0].item(), exp, rel_tol=tol if ft == "f64" else 5e-5, abs_tol=0.0)
    for key in ("chen2022_rain_velocity_1m", "chen2022_rain_velocity_gpu"):
        g = G[key]
        v = cmx.terminal_velocity_1m(mp, col(g["rho"]), col(g["q_rai"]), chen=True)
        assert math.isclose(v.vt_rai_chen[0].item(), g["expected"], rel_tol=max(g["rtol"], 1e-7 if ft == "f64" else 5e-5))
    g = G["prescribed_nd"]
    nd = P.Microphysics1MParams(ft, rain_autoconversion=P.PrescribedNd())
    r = cmx.microphysics_source_terms_1m(nd, tps, col(1.0), col(280.0), col(0.0), torch.tensor([g["q_lcl"], 0.0], dtype=DT[ft], device=dev),
                                         col(0.0), col(0.0), col(0.0))
    assert math.isclose(r.S_acnv_lcl_rai[0].item(), g["expected"], rel_tol=g["rtol"]) and r.S_acnv_lcl_rai[1].item() == 0
    with pytest.raises(cmx.CmxStatusError):   # two variants of one process → CMX_ERR_BAD_ARG
        both = P.Microphysics1MParams(ft)
        fn = getattr(cmx._lib.lib(), f"cmx_mp1m_source_terms_{ft}")
        import ctypes as C
        arr = (C.c_void_p * _abi.CMX_MP1M_NSRC)()
        z = col(1.0)
        cmx._lib.check("cmx_mp1m_source_terms", fn(C.byref(both.c), C.byref(tps), both.flags | _abi.CMX_1M_CLOUD_ICE_FORMATION_TDEP, 2,
                                                   *[C.c_void_p(z.data_ptr())] * 7, arr, None))


@pytest.mark.parametrize("ft", ["f32", "f64"])
@pytest.mark.parametrize("optset", list(OPTION_SETS))
def test_tendencies_and_sources_match_oracle(dev, oracle, ft, optset):
    import cmx
    from cmx import synthetic
    opts = OPTION_SETS[optset]
    n = 1_000_003 if optset == "default" else 200_001
    st = synthetic.mp1m_state(n, dtype=DT[ft], seed=1234)
    mp, tps = P.Microphysics1MParams(ft, **opts), P.ThermodynamicsParameters(ft)
    dcols = [c.to(dev) for c in st]
    tend = cmx.bulk_microphysics_tendencies_1m(cmx.Instantaneous(), cmx.Microphysics1Moment(), mp, tps, *dcols)
    src = cmx.microphysics_source_terms_1m(mp, tps, *dcols)
    torch.cuda.synchronize()
    ref = _oracle(oracle, ft, opts, [c.numpy() for c in st])
    got = {k: getattr(tend, k).cpu().numpy() for k in TN}
    rep = parity.assert_parity(got, ref, parity.RTOL[ft], names=TN, what=f"1M {ft} {optset}")
    print(f"\n[1M parity] {ft} {optset} n={n}: {rep}")
    # every source term: products of positive factors except the vapour / melt terms, which get the operand scales
    # through the tendency they enter
    worst = {}
    keep = ~ref["near_branch"]
    cancel = {"S_phase_change_vap_lcl": "dq_lcl_dt", "S_phase_change_vap_icl": "dq_icl_dt", "S_phase_change_vap_rai": "dq_rai_dt",
              "S_phase_change_vap_sno": "dq_sno_dt", "S_melt_icl_lcl": "dq_icl_dt", "S_melt_sno_rai": "dq_sno_dt",
              "S_accr_melt_lcl_sno": "dq_sno_dt", "S_accr_melt_rai_sno": "dq_sno_dt",
              # WithSupersaturation ∝ S_i; Kessler-type logistic integrals cancel below the threshold
              "S_acnv_icl_sno": "dq_icl_dt", "S_acnv_lcl_rai": "dq_lcl_dt"}
    for k in _abi.MP1M_SOURCE_COLUMNS:
        sc = ref["scale"][cancel[k]] if k in cancel else None
        e = parity.scaled_err(getattr(src, k).cpu().numpy(), ref["sources"][k], sc, parity.FLOOR[ft], parity.CEIL[ft],
                              parity.CTOL[ft] / parity.RTOL[ft])[keep]
        worst[k] = float(np.nan_to_num(e, nan=np.inf).max())
        assert worst[k] <= parity.RTOL[ft], (k, worst[k])
    print(f"[1M source parity] {ft} {optset}: worst {max(worst.values()):.2e} ({max(worst, key=worst.get)})")


def test_config1_autoconversion_and_terminal_velocity_1e6_f64(dev, oracle):
    """BASELINE config 1: conv_q_lcl_to_q_rai + terminal_velocity over 1e6 random (ρ, q) Float64 points."""
    import cmx
    n = 1_000_000
    g = torch.Generator().manual_seed(1234)
    u = lambda: torch.rand(n, dtype=torch.float64, generator=g)  # noqa: E731
    rho = 0.3 + u()
    sel = u()
    q = torch.where(sel < 0.5, torch.exp(math.log(1e-8) + u() * math.log(5e5)), torch.where(sel < 0.95, torch.zeros(n, dtype=torch.float64), -1e-9 * u()))
    mp, tps = P.Microphysics1MParams("f64"), P.ThermodynamicsParameters("f64")
    z = torch.zeros(n, dtype=torch.float64)
    d = lambda t: t.to(dev)  # noqa: E731
    src = cmx.microphysics_source_terms_1m(mp, tps, d(rho), d(z + 280.0), d(q), d(q), d(z), d(z), d(z))
    vel = cmx.terminal_velocity_1m(mp, d(rho), d(q), d(q), chen=True)
    torch.cuda.synchronize()
    ref = oracle.mp1m(_abi.F64, mp.c, tps, mp.flags, rho.numpy(), (z + 280.0).numpy(), q.numpy(), q.numpy(), z.numpy(), z.numpy(), z.numpy())
    rv = oracle.mp1m_terminal_velocity(_abi.F64, mp.c, P.Chen2022VelTypeRain("f64"), rho.numpy(), q.numpy(), q.numpy())
    np.testing.assert_allclose(src.S_acnv_lcl_rai.cpu().numpy(), ref["sources"]["S_acnv_lcl_rai"], rtol=1e-6, atol=1e-300)
    for k in ("vt_rai_blk1m", "vt_sno_blk1m", "vt_rai_chen"):
        np.testing.assert_allclose(getattr(vel, k).cpu().numpy(), rv[k], rtol=1e-6, atol=0, err_msg=k)
    assert (vel.vt_rai_blk1m[q.to(dev) <= 0] == 0).all()


@pytest.mark.parametrize("ft", ["f32", "f64"])
def test_non_default_slope_exponents_take_the_general_kernels(dev, oracle, ft):
    """The default option set with Δv = 0.1 on the rain and snow fall-speed relations: the slope-parameter exponents are no longer the
    multiples of ¼ / ⅛ the default instantiation multiplies out (csrc/cmx_mp1m_kernels.hip kDefExpBit), so the entry points must take
    the general exp2(e·log2 λ⁻¹) kernels — same oracle, same bound."""
    import cmx
    from cmx import synthetic
    n = 100_003
    st = synthetic.mp1m_state(n, dtype=DT[ft], seed=77)
    tps = P.ThermodynamicsParameters(ft)
    mp, mp64 = P.Microphysics1MParams(ft), P.Microphysics1MParams("f64")
    for m in (mp, mp64):
        m.c.vel_rain.delta_v = 0.1
        m.c.vel_snow.delta_v = 0.1
    dcols = [c.to(dev) for c in st]
    tend = cmx.bulk_microphysics_tendencies_1m(cmx.Instantaneous(), cmx.Microphysics1Moment(), mp, tps, *dcols)
    lin = cmx.bulk_microphysics_tendencies_1m(cmx.LinearizedAverage(), cmx.Microphysics1Moment(), mp, tps, *dcols, 20.0, 2)
    torch.cuda.synchronize()
    ref = oracle.mp1m(_abi.F64, mp64.c, P.ThermodynamicsParameters("f64"), mp64.flags, *[c.numpy().astype(np.float64) for c in st],
                      float32_gates=(ft == "f32"), nthreads=8)
    tf = P.DEFAULT_PARAMETERS["temperature_water_freeze"]
    ref["near_branch"] = np.abs(st[1].numpy().astype(np.float64) - tf) < (1e-4 if ft == "f32" else 1e-11)
    got = {k: getattr(tend, k).cpu().numpy() for k in TN}
    rep = parity.assert_parity(got, ref, parity.RTOL[ft], names=TN, what=f"1M {ft} delta_v=0.1")
    print(f"\n[1M parity, non-default exponents] {ft}: {rep}")
    assert all(bool(torch.isfinite(getattr(lin, k)).all()) for k in TN)
    # and the result differs from the default-parameter one (the perturbation is seen)
    base = cmx.bulk_microphysics_tendencies_1m(cmx.Instantaneous(), cmx.Microphysics1Moment(), P.Microphysics1MParams(ft), tps, *dcols)
    assert not torch.equal(base.dq_rai_dt, tend.dq_rai_dt)


@pytest.mark.parametrize("ft", ["f32", "f64"])
@pytest.mark.parametrize("n", [0, 1, 3, 5, 257, 1023])
def test_ragged_sizes(dev, oracle, ft, n):
    import cmx
    from cmx import synthetic
    st = [c[:n].contiguous() for c in synthetic.mp1m_state(max(n, 1), dtype=DT[ft], seed=n + 3)]
    mp, tps = P.Microphysics1MParams(ft), P.ThermodynamicsParameters(ft)
    t = cmx.bulk_microphysics_tendencies_1m(cmx.Instantaneous(), cmx.Microphysics1Moment(), mp, tps, *[c.to(dev) for c in st])
    assert t.dq_lcl_dt.shape == (n,)
    if n:
        ref = _oracle(oracle, ft, {}, [c.numpy() for c in st], want_sources=False)
        parity.assert_parity({k: getattr(t, k).cpu().numpy() for k in TN}, ref, parity.RTOL[ft], names=TN, what=f"n={n}")


def test_unaligned_and_errors(dev):
    import cmx
    from cmx import synthetic
    st = [c.to(dev) for c in synthetic.mp1m_state(10_002, seed=2)]
    mp, tps = P.Microphysics1MParams("f32"), P.ThermodynamicsParameters("f32")
    call = lambda cols: cmx.bulk_microphysics_tendencies_1m(cmx.Instantaneous(), cmx.Microphysics1Moment(), mp, tps, *cols)  # noqa: E731
    a, b = call([c[1:] for c in st]), call([c[1:].clone() for c in st])
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    with pytest.raises(TypeError):
        cmx.bulk_microphysics_tendencies_1m(cmx.Instantaneous(), cmx.Microphysics1Moment(), P.Microphysics1MParams("f64"), tps, *st)
    with pytest.raises(ValueError):
        call([c.cpu() for c in st])


def test_full_size_1e8_f32_properties(dev, oracle):
    import cmx
    from cmx import sharding, synthetic
    n = 100_000_000
    st = synthetic.mp1m_state(n, dtype=torch.float32, device=dev, seed=1234)
    mp, tps = P.Microphysics1MParams("f32"), P.ThermodynamicsParameters("f32")
    call = lambda cols: cmx.bulk_microphysics_tendencies_1m(cmx.Instantaneous(), cmx.Microphysics1Moment(), mp, tps, *cols)  # noqa: E731
    full = call(st)
    torch.cuda.synchronize()
    for k, v in full._asdict().items():
        assert bool(torch.isfinite(v).all()), k
    for lo, hi in ((0, 4096), (12_345_677, 12_400_001), (n - 1_000_003, n)):
        part = call([c[lo:hi] for c in st])
        for a, b in zip(full, part):
            assert torch.equal(a[lo:hi], b), (lo, hi)
    tot = cmx.column_sums(list(full))
    acc = torch.zeros_like(tot)
    for r in range(8):
        lo, hi = sharding.shard_bounds(n, r, 8)
        acc += cmx.column_sums([c[lo:hi] for c in full])
    assert torch.allclose(tot, acc, rtol=1e-9, atol=0)
    stride = 101
    samp = [c[::stride].contiguous().cpu().numpy() for c in st]
    ref = _oracle(oracle, "f32", {}, samp, want_sources=False)
    got = {k: getattr(full, k)[::stride].contiguous().cpu().numpy() for k in TN}
    rep = parity.assert_parity(got, ref, parity.RTOL["f32"], names=TN, what="1M 1e8 f32 sample")
    print(f"\n[1M parity 1e8 f32, {samp[0].size} sampled points] {rep}")


@pytest.mark.parametrize("ft", ["f64", "f32"])
def test_sedimentation_velocities(dev, oracle, ft):
    """cmx_sedimentation_velocities_*: the reference's KATs (test/gpu_tests.jl:608-630) through the C ABI + random parity."""
    import cmx
    dt = DT[ft]
    g = G["chen2022_sedimentation_velocities"]
    mp = P.Microphysics1MParams(ft)
    par = (P.StokesRegimeVelType(ft), P.Chen2022VelTypeRain(ft), P.Chen2022VelTypeIce(ft))
    col = lambda v: torch.tensor(v, dtype=dt, device=dev)  # noqa: E731
    r = cmx.sedimentation_velocities(mp, *par, col([g["rho"]]), col([g["q_lcl"]]), col([g["q_icl"]]), col([g["q_rai"]]), col([g["q_sno"]]))
    for k in ("w_lcl", "w_icl", "w_rai", "w_sno"):
        assert float(getattr(r, k)[0]) == pytest.approx(g[k], rel=1e-10 if ft == "f64" else 2e-5), k
    only = cmx.sedimentation_velocities(mp, par[0], None, None, col([0.95]), q_lcl=col([0.004]))
    assert only.w_icl is None and float(only.w_lcl[0]) == pytest.approx(g["w_lcl"], rel=1e-5)
    n = 100_003
    gen = torch.Generator().manual_seed(8)
    rho = (0.3 + torch.rand(n, generator=gen, dtype=torch.float64)).to(dt)
    qq = lambda: torch.where(torch.rand(n, generator=gen, dtype=torch.float64) < 0.15, torch.zeros(n, dtype=torch.float64),  # noqa: E731
                             10 ** (-9 + 6.5 * torch.rand(n, generator=gen, dtype=torch.float64))).to(dt)
    qs = [qq() for _ in range(4)]
    got = cmx.sedimentation_velocities(mp, *par, rho.to(dev), *[q.to(dev) for q in qs])
    mp64 = P.Microphysics1MParams("f64")
    ref = oracle.sedimentation_velocities(_abi.F64, mp64.c, P.StokesRegimeVelType("f64"), P.Chen2022VelTypeRain("f64"),
                                          P.Chen2022VelTypeIce("f64"), rho.numpy().astype(np.float64),
                                          *[q.numpy().astype(np.float64) for q in qs], float32_gates=(ft == "f32"))
    # the Chen-2022 ice curves are differences of two terms that cancel near the zero crossing (E + F e^{−cD} with E ≈ −F):
    # conditioning scale = the positive term alone (oracle evaluated with the negative amplitude switched off)
    pos = P.Chen2022VelTypeIce("f64")
    pos.small_ice.F[0] = -1e30                                   # Fs = −exp(F₀ − …) → 0
    pos.large_ice.E[0], pos.large_ice.E[1], pos.large_ice.E[2] = 0.0, 0.0, 0.0   # El → 0 (second large-ice amplitude)
    scale = oracle.sedimentation_velocities(_abi.F64, mp64.c, P.StokesRegimeVelType("f64"), P.Chen2022VelTypeRain("f64"), pos,
                                            rho.numpy().astype(np.float64), *[q.numpy().astype(np.float64) for q in qs],
                                            float32_gates=(ft == "f32"))
    for k in ("w_lcl", "w_icl", "w_rai", "w_sno"):
        x, rr = getattr(got, k).cpu().numpy().astype(np.float64), ref[k]
        sc = scale[k] if k in ("w_icl", "w_sno") else np.abs(rr)
        tol = parity.RTOL[ft] * np.abs(rr) + parity.CTOL[ft] * sc
        assert np.all(np.abs(x - rr) <= tol + 1e-300), (k, float(np.max(np.abs(x - rr) / (tol + 1e-300))))
        assert np.mean((x == 0) == (rr == 0)) > 0.999, k          # the max(0, ·) clamp agrees except within rounding of the crossing


@pytest.mark.parametrize("ft", ["f32", "f64"])
def test_degenerate_states(dev, oracle, ft):
    """All-zero tracers, negative inputs (clamped, BMT:147-152), values straddling ϵ = cbrt(floatmin), huge contents, temperatures around T_freeze
    and far from it: the Float64 point function uses the finite-argument exp2 / reciprocal forms — every state must stay finite where
    the oracle is finite and agree with it."""
    import cmx
    eps = float(np.cbrt(np.finfo(np.float32 if ft == "f32" else np.float64).tiny))
    rows = []
    for q_lcl in (0.0, -1e-6, eps * 0.5, eps * 2, 1e-3):
        for q_icl in (0.0, eps * 2, 2e-4):
            for q_rai in (0.0, -1e-7, eps * 0.5, 5e-3, 0.2):
                for q_sno in (0.0, eps * 2, 3e-3):
                    for T, q_tot, rho in ((290.0, 1.5e-2, 1.1), (273.16, 4e-3, 1.0), (273.14, 4e-3, 0.9), (240.0, 3e-4, 0.6), (215.0, 0.0, 0.3), (305.0, 0.3, 1.25)):
                        rows.append((rho, T, q_tot, q_lcl, q_icl, q_rai, q_sno))
    arr = np.array(rows, dtype=np.float64).T
    cols = [torch.tensor(a, dtype=DT[ft]) for a in arr]
    for optset in ("default", "tdep"):
        opts = OPTION_SETS[optset]
        mp, tps = P.Microphysics1MParams(ft, **opts), P.ThermodynamicsParameters(ft)
        tend = cmx.bulk_microphysics_tendencies_1m(cmx.Instantaneous(), cmx.Microphysics1Moment(), mp, tps, *[c.to(dev) for c in cols])
        torch.cuda.synchronize()
        ref = _oracle(oracle, ft, opts, [c.numpy() for c in cols])
        got = {k: getattr(tend, k).cpu().numpy() for k in TN}
        for k in TN:
            assert np.all(np.isfinite(got[k]) | ~np.isfinite(ref[k]) | (np.abs(ref[k]) > parity.CEIL[ft])), (optset, k)
        # the conditioned metric is asserted as everywhere; the PLAIN relative bound is a statement about typical states — this set sits on the
        # cancellation points on purpose (q_v ≈ q_sat, T = T_freeze ± 0.01 K: T_freeze itself is a Float32-rounded parameter in the Float32 kernel; ≈ 10 % of the Float32 points lose more than three digits there)
        parity.assert_parity(got, ref, parity.RTOL[ft], names=TN, what=f"1M degenerate {ft} {optset}", min_frac=0.85 if ft == "f32" else None)
