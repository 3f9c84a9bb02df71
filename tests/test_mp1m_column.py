"""The operational 1-moment column step (cmx_mp1m_column_tendencies_sedimentation_*): 1-moment tendencies
(Instantaneous / LinearizedAverage) + the four sedimentation velocities + the host model's upwind flux divergence in one pass.

CPU: the kernel's point functions compiled for the host (tests/native/point_host.cpp) against the column oracle.
GPU (-m gpu): through the C ABI against the column oracle (tendencies and fall speeds pinned by the reference's KATs; the flux
divergence is the host model's operator — parity unpinned, oracle/cmx_oracle_column_impl.h), against the UNFUSED product sequence
(pointwise entries + divergence in torch), ragged shapes / misaligned columns (bit-identical across paths), NaN propagation at and
away from tile boundaries, and mass conservation at 1e8 points."""
import ctypes as C

import numpy as np
import pytest
import torch

import parity
from cmx import _abi
from cmx import parameters as P

DT = {"f32": torch.float32, "f64": torch.float64}
NPT = {"f32": np.float32, "f64": np.float64}
CT = {"f32": C.c_float, "f64": C.c_double}
NAMES = ("dq_lcl_dt", "dq_icl_dt", "dq_rai_dt", "dq_sno_dt")
Q_MIN = P.DEFAULT_PARAMETERS["specific_humidity_minimum"]
T_FREEZE = P.DEFAULT_PARAMETERS["temperature_water_freeze"]


def _state(n_col, n_lev, ft, seed=3):
    from cmx import synthetic
    st = synthetic.mp1m_state(n_col * n_lev, dtype=DT[ft], seed=seed)
    return [c.reshape(n_col, n_lev) for c in st]


def _inv_dz(n_lev, ft, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (1.0 / (30.0 + 470.0 * torch.rand(n_lev, generator=g, dtype=torch.float64))).to(DT[ft])


def _vel_params(ft):
    return P.StokesRegimeVelType(ft), P.Chen2022VelTypeRain(ft), P.Chen2022VelTypeIce(ft)


def _oracle(oracle, ft, inv_dz, cols, dt=0.0, nsub=0, opts=None):
    mp = P.Microphysics1MParams("f64", **(opts or {}))
    pos = P.Chen2022VelTypeIce("f64")           # conditioning scale of the Chen-2022 ice curves: the negative amplitudes switched off
    pos.small_ice.F[0] = -1e30
    pos.large_ice.E[0], pos.large_ice.E[1], pos.large_ice.E[2] = 0.0, 0.0, 0.0
    ref = oracle.mp1m_column_tendencies_sedimentation(
        _abi.F64, mp.c, P.ThermodynamicsParameters("f64"), *_vel_params("f64"), mp.flags, inv_dz.numpy().astype(np.float64),
        *[c.numpy().astype(np.float64) for c in cols], q_min=Q_MIN, dt=dt, nsub=nsub, chen_ice_scale=pos, float32_gates=(ft == "f32"), nthreads=8)
    ref["near_branch"] = np.abs(cols[1].numpy().astype(np.float64).reshape(-1) - T_FREEZE) < (1e-3 if ft == "f32" else 1e-9)
    return ref


def _check(ft, got, ref, cols, dt, what, min_frac=None):
    """Instantaneous: the library's parity metric; LinearizedAverage: the same + the rounding floor of the difference quotient."""
    if dt:
        eps = {"f64": 2.2e-16, "f32": 1.2e-7}[ft]
        # the substeps couple the species (ice grown in substep 1 converts to snow in substep 2): conditioning scale = Σ over the four
        # species, as in tests/test_mp1m_linearized.py
        scale = sum(ref["scale"][k] for k in NAMES)
        for k, q0 in zip(NAMES, cols[3:]):
            q0 = q0.numpy().astype(np.float64).reshape(-1)
            x, r = np.asarray(got[k], dtype=np.float64), ref[k]
            tol = parity.RTOL[ft] * np.abs(r) + parity.CTOL[ft] * scale + 8 * eps * (np.abs(q0) + np.abs(r) * dt) / dt   # slightly negative q occur
            e = (np.abs(x - r) / np.maximum(tol, 1e-300))[~ref["near_branch"]]
            assert np.all(np.isfinite(x)) and e.max() <= 1.0, (what, k, float(e.max()))
        return {}
    return parity.assert_parity(got, ref, parity.RTOL[ft], names=NAMES, what=what, min_frac=min_frac)


# ---- CPU: host build of the point functions ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("ft", ["f32", "f64"])
@pytest.mark.parametrize("dt,nsub", [(0.0, 0), (30.0, 2)])
def test_host_build_column_step_matches_oracle(oracle, ft, dt, nsub):
    from test_point_host import _ptrs
    import subprocess
    from pathlib import Path
    repo = Path(__file__).resolve().parent.parent
    so = repo / "tests" / "native" / "_build" / "libpoint_host.so"
    so.parent.mkdir(exist_ok=True)
    subprocess.run(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-o", str(so), str(repo / "tests" / "native" / "point_host.cpp")],
                   check=True)
    host = C.CDLL(str(so))
    n_col, n_lev = 400, 37
    cols = _state(n_col, n_lev, ft)
    inv_dz = _inv_dz(n_lev, ft)
    mp, tps = P.Microphysics1MParams(ft), P.ThermodynamicsParameters(ft)
    st, cr, ci = _vel_params(ft)
    x = [np.ascontiguousarray(c.numpy().reshape(-1)) for c in cols]
    y = [np.empty_like(x[0]) for _ in range(4)]
    pr, ps = np.empty(n_col, NPT[ft]), np.empty(n_col, NPT[ft])
    dz = np.ascontiguousarray(inv_dz.numpy())
    P_ = C.POINTER(CT[ft])
    fn = getattr(host, f"host_mp1m_column_{ft}")
    fn.restype = C.c_int32
    fn(C.byref(mp.c), C.byref(tps), C.byref(st), C.byref(cr), C.byref(ci), C.c_uint32(mp.flags), CT[ft](Q_MIN), CT[ft](dt), C.c_int32(nsub),
       C.c_int64(n_col), C.c_int32(n_lev), dz.ctypes.data_as(P_), _ptrs(x, ft), _ptrs(y, ft), pr.ctypes.data_as(P_), ps.ctypes.data_as(P_))
    ref = _oracle(oracle, ft, inv_dz, cols, dt, nsub)
    _check(ft, dict(zip(NAMES, y)), ref, cols, dt, f"host-build 1M column {ft} dt={dt} nsub={nsub}")
    np.testing.assert_allclose(pr, ref["precip_rai"], rtol=10 * parity.RTOL[ft], atol=parity.FLOOR[ft])
    np.testing.assert_allclose(ps, ref["precip_sno"], rtol=10 * parity.RTOL[ft], atol=1e-12)
    assert pr.max() > 0 and ps.max() > 0


# ---- GPU ------------------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _run(dev, ft, inv_dz, cols, dt=None, nsub=1, opts=None, **kw):
    import cmx
    mp, tps = P.Microphysics1MParams(ft, **(opts or {})), P.ThermodynamicsParameters(ft)
    mode = cmx.LinearizedAverage() if dt else cmx.Instantaneous()
    return cmx.column_tendencies_sedimentation_1m(mode, cmx.Microphysics1Moment(), mp, tps, *_vel_params(ft), inv_dz.to(dev), *[c.to(dev) for c in cols],
                                                  dt, nsub, **kw)


@pytest.mark.gpu
@pytest.mark.parametrize("ft", ["f32", "f64"])
@pytest.mark.parametrize("dt,nsub,opts", [(None, 0, None), (30.0, 2, None), (None, 0, dict(cloud_ice_formation=P.TemperatureDependent(), rain_autoconversion=P.PrescribedNd()))])
def test_column_step_matches_oracle(dev, oracle, ft, dt, nsub, opts):
    n_col, n_lev = 2703, 74                                       # 200 022 points; 74 levels = the RCEMIP column of the reference's test
    cols = _state(n_col, n_lev, ft)
    inv_dz = _inv_dz(n_lev, ft)
    got = _run(dev, ft, inv_dz, cols, dt, nsub, opts)
    torch.cuda.synchronize()
    ref = _oracle(oracle, ft, inv_dz, cols, dt or 0.0, nsub, opts)
    rep = _check(ft, {k: getattr(got, k).reshape(-1).cpu().numpy() for k in NAMES}, ref, cols, dt, f"1M column {ft} dt={dt} nsub={nsub} opts={'alt' if opts else 'default'}")
    print(f"\n[1M column parity] {ft} dt={dt} nsub={nsub}: {rep}")
    np.testing.assert_allclose(got.precip_rai.cpu().numpy().astype(np.float64), ref["precip_rai"], rtol=10 * parity.RTOL[ft], atol=parity.FLOOR[ft])
    np.testing.assert_allclose(got.precip_sno.cpu().numpy().astype(np.float64), ref["precip_sno"], rtol=10 * parity.RTOL[ft], atol=1e-12)
    assert float(got.precip_rai.max()) > 0 and float(got.precip_sno.max()) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("ft", ["f32", "f64"])
@pytest.mark.parametrize("lin", [False, True])
def test_fused_equals_unfused_product_sequence(dev, ft, lin):
    """What a host model does today with the pointwise entries: tendencies, the four sedimentation velocities, then the upwind divergence
    (here in torch).  Same library, same point functions → agreement to rounding of the divergence arithmetic."""
    import cmx
    n_col, n_lev = 1500, 37
    cols = [c.to(dev) for c in _state(n_col, n_lev, ft, seed=12)]
    inv_dz = _inv_dz(n_lev, ft, seed=2).to(dev)
    mp, tps = P.Microphysics1MParams(ft), P.ThermodynamicsParameters(ft)
    dt, nsub = (40.0, 2) if lin else (None, 1)
    mode = cmx.LinearizedAverage() if lin else cmx.Instantaneous()
    fused = cmx.column_tendencies_sedimentation_1m(mode, cmx.Microphysics1Moment(), mp, tps, *_vel_params(ft), inv_dz, *cols, dt, nsub)
    flat = [c.reshape(-1) for c in cols]
    pt = cmx.bulk_microphysics_tendencies_1m(mode, cmx.Microphysics1Moment(), mp, tps, *flat, dt, nsub)
    rho = torch.clamp(cols[0], min=0)
    qs = [torch.clamp(c, min=0) for c in cols[3:]]
    w = cmx.sedimentation_velocities(mp, *_vel_params(ft), rho.reshape(-1), *[q.reshape(-1) for q in qs])
    torch.cuda.synchronize()
    sh = (n_col, n_lev)
    for name, chi, wk in zip(NAMES, qs, w):
        F = (rho * chi) * wk.reshape(sh)
        up = torch.cat([F[:, 1:], torch.zeros_like(F[:, :1])], dim=1)
        base = getattr(pt, name).reshape(sh)
        expect = base + (up - F) * (inv_dz[None, :] / rho)
        err = (getattr(fused, name) - expect).abs()
        tol = (8 if ft == "f32" else 64) * torch.finfo(DT[ft]).eps * (base.abs() + (up + F) * (inv_dz[None, :] / rho))
        assert bool((err <= tol + torch.finfo(DT[ft]).tiny).all()), (name, float((err - tol).max()))
    assert torch.equal(fused.precip_rai, ((rho * qs[2]) * w.w_rai.reshape(sh))[:, 0])


@pytest.mark.gpu
@pytest.mark.parametrize("ft", ["f32", "f64"])
@pytest.mark.parametrize("shape", [(1, 1), (3, 1), (1, 2), (5, 3), (7, 5), (11, 74), (1, 1000), (129, 63), (64, 64)])
def test_ragged_shapes_and_tile_invariance(dev, oracle, ft, shape):
    """Small / odd shapes (levels shorter than a lane vector, columns straddling workgroup tiles, single columns) against the oracle and
    against the same call on misaligned copies (scalar head / tail, one-point-per-lane path): the bits must not depend on the path."""
    import cmx
    n_col, n_lev = shape
    cols = _state(n_col, n_lev, ft, seed=31 + n_col)
    inv_dz = _inv_dz(n_lev, ft, seed=n_lev)
    got = _run(dev, ft, inv_dz, cols)
    torch.cuda.synchronize()
    ref = _oracle(oracle, ft, inv_dz, cols)
    _check(ft, {k: getattr(got, k).reshape(-1).cpu().numpy() for k in NAMES}, ref, cols, None, f"1M column {ft} {shape}")
    n = n_col * n_lev
    mp, tps = P.Microphysics1MParams(ft), P.ThermodynamicsParameters(ft)
    fn = getattr(cmx._lib.lib(), f"cmx_mp1m_column_tendencies_sedimentation_{ft}")
    dz = inv_dz.to(dev)
    for offs in ([1] * 11, [0, 1, 2, 3, 0, 1, 2, 3, 0, 1, 2]):
        bufs = [torch.empty(n + 4, dtype=DT[ft], device=dev) for _ in range(11)]
        views = [b[o:o + n] for b, o in zip(bufs, offs)]
        for d, s_ in zip(views[:7], cols):
            d.copy_(s_.reshape(-1))
        st, cr, ci = _vel_params(ft)
        rc = fn(C.byref(mp.c), C.byref(tps), C.byref(st), C.byref(cr), C.byref(ci), mp.flags, 0.0, 0.0, 0, n_col, n_lev, C.c_void_p(dz.data_ptr()),
                (C.c_void_p * 7)(*[v.data_ptr() for v in views[:7]]), (C.c_void_p * 4)(*[v.data_ptr() for v in views[7:]]), None, None, None)
        assert rc == 0
        torch.cuda.synchronize()
        for k, v in zip(NAMES, views[7:]):
            assert torch.equal(v, getattr(got, k).reshape(-1)), (k, offs)


@pytest.mark.gpu
@pytest.mark.parametrize("ft", ["f32", "f64"])
def test_nan_rule_does_not_depend_on_tile_boundaries(dev, ft):
    """A NaN in ρ or in a species' q poisons that species' flux → the same species' tendency in the cell BELOW, and all four tendencies
    of the point itself; a NaN in T or q_tot poisons the point only.  Identical at a tile boundary (flat index a multiple of
    128·VEC) and at an interior point (the 2-moment column kernel treated the two differently)."""
    n_col, n_lev = 6, 1024
    vec = 4 if ft == "f32" else 1
    boundary = 128 * vec * 3            # first point of a workgroup tile
    base = _state(n_col, n_lev, ft, seed=5)
    inv_dz = _inv_dz(n_lev, ft)
    clean = _run(dev, ft, inv_dz, base, precip=False)
    for flat_idx in (boundary, boundary + 37 * vec + 1):
        for col_i, name in ((1, "T"), (2, "q_tot"), (5, "q_rai"), (0, "rho")):
            cols = [c.clone() for c in base]
            cols[col_i].reshape(-1)[flat_idx] = float("nan")
            got = _run(dev, ft, inv_dz, cols, precip=False)
            torch.cuda.synchronize()
            for k, species_col in zip(NAMES, (3, 4, 5, 6)):
                g, c0 = getattr(got, k).reshape(-1), getattr(clean, k).reshape(-1)
                assert torch.isnan(g[flat_idx]), (name, k, flat_idx)                    # the point itself: every tendency
                below_poisoned = name == "rho" or col_i == species_col
                assert bool(torch.isnan(g[flat_idx - 1])) == below_poisoned, (name, k, flat_idx)
                mask = torch.ones_like(g, dtype=torch.bool)
                mask[flat_idx - 1:flat_idx + 1] = False
                assert torch.equal(g[mask], c0[mask]), (name, k, flat_idx)              # nothing else moves


@pytest.mark.gpu
def test_full_size_conservation_1e8(dev):
    """1e8 Float32 points (BASELINE size): with the microphysics switched off the column step is pure sedimentation, and the mass of
    each species changes only through the surface: Σ_k ρ_k Δz_k ∂χ_k/∂t = −F_0."""
    import cmx
    n_lev, n_col = 64, 1_562_500
    ft = "f32"
    cols = [c.to(dev) for c in _state(n_col, n_lev, ft, seed=77)]
    inv_dz = _inv_dz(n_lev, ft, seed=4).to(dev)
    none = {k: None for k in P.Microphysics1MOptions._defaults}
    mp, tps = P.Microphysics1MParams(ft, **none), P.ThermodynamicsParameters(ft)
    got = cmx.column_tendencies_sedimentation_1m(cmx.Instantaneous(), cmx.Microphysics1Moment(), mp, tps, *_vel_params(ft), inv_dz, *cols)
    torch.cuda.synchronize()
    rho = torch.clamp(cols[0], min=0).double()
    dz = (1.0 / inv_dz.double())[None, :]
    for k, pr in (("dq_rai_dt", got.precip_rai), ("dq_sno_dt", got.precip_sno)):
        col_sum = (rho * dz * getattr(got, k).double()).sum(dim=1)
        scale = (rho * dz * getattr(got, k).double().abs()).sum(dim=1) + pr.double().abs()
        assert bool(((col_sum + pr.double()).abs() <= 1e-5 * scale + 1e-30).all()), k


# ---- argument validation: the 2-D tensors themselves are checked, before anything touches the GPU ---------------------------
def test_column_entry_refuses_transposed_and_misshapen_tensors():
    import cmx
    ft = "f32"
    n_col, n_lev = 8, 64
    cols = _state(n_col, n_lev, ft)
    mp, tps = P.Microphysics1MParams(ft), P.ThermodynamicsParameters(ft)
    call = lambda c: cmx.column_tendencies_sedimentation_1m(cmx.Instantaneous(), cmx.Microphysics1Moment(), mp, tps, *_vel_params(ft), _inv_dz(n_lev, ft), *c)  # noqa: E731
    # a transposed view has the right shape after .T.T only; as (n_lev, n_col).T it is (n_col, n_lev) but NOT contiguous
    tr = [c.clone() for c in cols]
    tr[3] = cols[3].t().contiguous().t()
    assert tr[3].shape == cols[3].shape and not tr[3].is_contiguous()
    with pytest.raises(ValueError, match="q_lcl must be a contiguous"):
        call(tr)
    # same number of elements, other shape
    bad = [c.clone() for c in cols]
    bad[5] = cols[5].reshape(n_lev, n_col)
    with pytest.raises(ValueError, match="q_rai: shape"):
        call(bad)
    # a non-row-major rho: the outputs would have inherited its strides (torch.empty_like) while the kernel writes row-major
    rho_t = [cols[0].t().contiguous().t()] + [c.clone() for c in cols[1:]]
    with pytest.raises(ValueError, match="rho must be a contiguous"):
        call(rho_t)
    with pytest.raises(ValueError, match="shape"):
        call([c.reshape(-1) for c in cols])


@pytest.mark.gpu
def test_column_entry_refuses_parameter_structs_of_the_other_float_type():
    import cmx
    dev = torch.device("cuda:0")
    ft, other = "f32", "f64"
    n_col, n_lev = 4, 32
    cols = [c.to(dev) for c in _state(n_col, n_lev, ft)]
    mp, tps = P.Microphysics1MParams(ft), P.ThermodynamicsParameters(ft)
    st, cr, ci = _vel_params(ft)
    so, co, io = _vel_params(other)
    for args in ((so, cr, ci), (st, co, ci), (st, cr, io)):
        with pytest.raises(TypeError, match="struct of the state's float type"):
            cmx.column_tendencies_sedimentation_1m(cmx.Instantaneous(), cmx.Microphysics1Moment(), mp, tps, *args, _inv_dz(n_lev, ft).to(dev), *cols)
    ok = cmx.column_tendencies_sedimentation_1m(cmx.Instantaneous(), cmx.Microphysics1Moment(), mp, tps, st, cr, ci, _inv_dz(n_lev, ft).to(dev), *cols)
    assert all(getattr(ok, k).is_contiguous() and getattr(ok, k).shape == (n_col, n_lev) for k in NAMES)
