"""GPU parity tests of the fused column kernel (cmx_sb2006_column_tendencies_sedimentation_*) through the C ABI:
against the column oracle (tendencies and fall speeds pinned by the reference's KATs; the flux divergence itself is the host
model's scheme — parity unpinned, oracle/cmx_oracle_column_impl.h), against the UNFUSED product sequence (pointwise entry +
velocities, divergence in torch), for ragged shapes, misaligned columns, and at the full BASELINE size through conservation."""
import numpy as np
import pytest
import torch

import parity
from cmx import _abi
from cmx import parameters as P

pytestmark = pytest.mark.gpu
DT = {"f32": torch.float32, "f64": torch.float64}
NAMES = ("dq_lcl_dt", "dn_lcl_dt", "dq_rai_dt", "dn_rai_dt")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _state(n_col, n_lev, ft, seed=3):
    from cmx import synthetic
    st = synthetic.sb2006_state(n_col * n_lev, dtype=DT[ft], seed=seed)
    return [c.reshape(n_col, n_lev) for c in st]


def _inv_dz(n_lev, ft, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (1.0 / (30.0 + 470.0 * torch.rand(n_lev, generator=g, dtype=torch.float64))).to(DT[ft])


def _vel(name):
    import cmx
    return cmx.SB2006VelType if name == "sb" else cmx.Chen2022VelTypeRain


def _oracle(oracle, ft, limited, vel, cloud, inv_dz, cols):
    flags = (_abi.CMX_SB2006_LIMITED if limited else 0) | (_abi.CMX_VEL_SB2006 if vel == "sb" else _abi.CMX_VEL_CHEN2022)
    return oracle.sb2006_column_tendencies_sedimentation(
        _abi.F64, P.WarmRainParams2M("f64", limited).c, P.ThermodynamicsParameters("f64"), P.rain_vel_params("f64"),
        P.StokesRegimeVelType("f64") if cloud else None, flags, inv_dz.numpy().astype(np.float64),
        *[c.numpy().astype(np.float64) for c in cols], float32_gates=(ft == "f32"), nthreads=8,
        branch_margin=1e-5 if ft == "f32" else 1e-11)


@pytest.mark.parametrize("ft", ["f32", "f64"])
@pytest.mark.parametrize("limited", [True, False])
@pytest.mark.parametrize("vel,cloud", [("sb", False), ("sb", True), ("chen", True)])
def test_column_step_matches_oracle(dev, oracle, ft, limited, vel, cloud):
    import cmx
    n_col, n_lev = 2703, 74                                       # 200 022 points; 74 levels = the RCEMIP column of the reference's test
    cols = _state(n_col, n_lev, ft)
    inv_dz = _inv_dz(n_lev, ft)
    mp, tps = P.Microphysics2MParams(ft, is_limited=limited), P.ThermodynamicsParameters(ft)
    got = cmx.column_tendencies_sedimentation(mp, tps, inv_dz.to(dev), *[c.to(dev) for c in cols], vel=_vel(vel),
                                              cloud_vel=P.StokesRegimeVelType(ft) if cloud else None, want_precip_flux=True)
    torch.cuda.synchronize()
    ref = _oracle(oracle, ft, limited, vel, cloud, inv_dz, cols)
    g = {k: getattr(got, k).reshape(-1).cpu().numpy() for k in NAMES}
    rep = parity.assert_parity(g, ref, parity.RTOL[ft], names=NAMES, what=f"column {ft} limited={limited} vel={vel} cloud={cloud}")
    print(f"\n[column parity] {ft} limited={limited} vel={vel} cloud={cloud}: {rep}")
    pf = got.precip_flux.cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(pf, ref["precip_flux"], rtol=parity.RTOL[ft], atol=parity.FLOOR[ft])
    assert pf.max() > 0


@pytest.mark.parametrize("ft", ["f32", "f64"])
def test_fused_equals_unfused_product_sequence(dev, ft):
    """The fused kernel against what a host model does today with the pointwise entry: tendencies + velocities, then the upwind
    divergence (here in torch).  Same library, same point function → agreement to rounding of the divergence arithmetic."""
    import cmx
    n_col, n_lev = 1500, 37
    cols = [c.to(dev) for c in _state(n_col, n_lev, ft, seed=12)]
    inv_dz = _inv_dz(n_lev, ft, seed=2).to(dev)
    mp, tps = P.Microphysics2MParams(ft), P.ThermodynamicsParameters(ft)
    fused = cmx.column_tendencies_sedimentation(mp, tps, inv_dz, *cols, vel=cmx.SB2006VelType)
    pt = cmx.bulk_microphysics_tendencies(cmx.Microphysics2Moment(), mp, tps, *[c.reshape(-1) for c in cols], vel=cmx.SB2006VelType)
    torch.cuda.synchronize()
    assert torch.equal(fused.dq_lcl_dt.reshape(-1), pt.dq_lcl_dt) and torch.equal(fused.dn_lcl_dt.reshape(-1), pt.dn_lcl_dt)   # no cloud sedimentation
    rho, q_rai, n_rai = [torch.clamp(c, min=0) for c in (cols[0], cols[5], cols[6])]
    sh = (n_col, n_lev)
    for name, chi, w in (("dq_rai_dt", q_rai, pt.vt_rai_m), ("dn_rai_dt", n_rai, pt.vt_rai_n)):
        F = (rho * chi) * w.reshape(sh)
        up = torch.cat([F[:, 1:], torch.zeros_like(F[:, :1])], dim=1)
        expect = getattr(pt, name).reshape(sh) + (up - F) * (inv_dz[None, :] / rho)
        got = getattr(fused, name)
        err = (got - expect).abs()
        tol = (8 if ft == "f32" else 64) * torch.finfo(DT[ft]).eps * (getattr(pt, name).reshape(sh).abs() + (up + F) * (inv_dz[None, :] / rho))
        assert bool((err <= tol + torch.finfo(DT[ft]).tiny).all()), (name, float((err - tol).max()))


@pytest.mark.parametrize("ft", ["f32", "f64"])
@pytest.mark.parametrize("shape", [(1, 1), (3, 1), (1, 2), (5, 3), (7, 5), (11, 74), (1, 1000), (129, 63), (64, 64)])
def test_ragged_shapes_and_tile_invariance(dev, oracle, ft, shape):
    """Small / odd shapes: levels shorter than a lane vector, columns that straddle workgroup tiles, single columns.  Each is checked
    against the oracle AND against the same call on misaligned copies (scalar path): the bits must not depend on the path."""
    import cmx
    n_col, n_lev = shape
    cols = _state(n_col, n_lev, ft, seed=31 + n_col)
    inv_dz = _inv_dz(n_lev, ft, seed=n_lev)
    mp, tps = P.Microphysics2MParams(ft), P.ThermodynamicsParameters(ft)
    stokes = P.StokesRegimeVelType(ft)
    dcols = [c.to(dev) for c in cols]
    got = cmx.column_tendencies_sedimentation(mp, tps, inv_dz.to(dev), *dcols, vel=cmx.SB2006VelType, cloud_vel=stokes, want_precip_flux=True)
    torch.cuda.synchronize()
    ref = _oracle(oracle, ft, True, "sb", True, inv_dz, cols)
    g = {k: getattr(got, k).reshape(-1).cpu().numpy() for k in NAMES}
    parity.assert_parity(g, ref, parity.RTOL[ft], names=NAMES, what=f"column {ft} {shape}")
    # misaligned twins: (a) every column shifted by one element (scalar head + vector body + scalar tail),
    # (b) columns at different offsets modulo 16 B (one point per lane throughout)
    n = n_col * n_lev
    for offs in ([1] * 11, [0, 1, 2, 3, 0, 1, 2, 3, 0, 1, 2]):
        bufs = [torch.empty(n + 4, dtype=DT[ft], device=dev) for _ in range(11)]
        ins = [b[o:o + n].view(n_col, n_lev) for b, o in zip(bufs[:7], offs[:7])]
        for d, s_ in zip(ins, dcols):
            d.copy_(s_)
        outs = cmx.ColumnTendencies2M(*[b[o:o + n].view(n_col, n_lev) for b, o in zip(bufs[7:], offs[7:])], None)
        cmx.column_tendencies_sedimentation(mp, tps, inv_dz.to(dev), *ins, vel=cmx.SB2006VelType, cloud_vel=stokes, out=outs)
        torch.cuda.synchronize()
        for k in NAMES:
            assert torch.equal(getattr(outs, k), getattr(got, k)), (k, offs)


def test_nan_inputs_poison_the_point_and_the_level_below(dev):
    import cmx
    ft, n_col, n_lev = "f32", 9, 20
    cols = [c.to(dev) for c in _state(n_col, n_lev, ft, seed=77)]
    cols[5][:, :] = torch.clamp(cols[5], min=1e-5)                  # rain everywhere, so every flux is live
    inv_dz = _inv_dz(n_lev, ft).to(dev)
    mp, tps = P.Microphysics2MParams(ft), P.ThermodynamicsParameters(ft)
    clean = cmx.column_tendencies_sedimentation(mp, tps, inv_dz, *cols, vel=cmx.SB2006VelType)
    cols[5][4, 10] = float("nan")
    bad = cmx.column_tendencies_sedimentation(mp, tps, inv_dz, *cols, vel=cmx.SB2006VelType)
    torch.cuda.synchronize()
    assert all(torch.isnan(getattr(bad, k)[4, 10]) for k in NAMES)
    assert torch.isnan(bad.dq_rai_dt[4, 9]) and torch.isnan(bad.dn_rai_dt[4, 9])          # receives the NaN flux from above
    assert not torch.isnan(bad.dq_lcl_dt[4, 9])
    mask = torch.ones(n_col, n_lev, dtype=torch.bool, device=dev)
    mask[4, 9:11] = False
    for k in NAMES:
        assert torch.equal(getattr(bad, k)[mask], getattr(clean, k)[mask]), k


def test_errors(dev):
    import cmx
    ft = "f32"
    cols = [c.to(dev) for c in _state(4, 6, ft)]
    mp, tps = P.Microphysics2MParams(ft), P.ThermodynamicsParameters(ft)
    inv_dz = _inv_dz(6, ft).to(dev)
    with pytest.raises(ValueError):
        cmx.column_tendencies_sedimentation(mp, tps, inv_dz, *cols, vel=None)
    with pytest.raises(ValueError):
        cmx.column_tendencies_sedimentation(mp, tps, inv_dz[:5], *cols)
    with pytest.raises(ValueError):
        cmx.column_tendencies_sedimentation(mp, tps, inv_dz, *[c.reshape(-1) for c in cols])
    with pytest.raises(TypeError):
        cmx.column_tendencies_sedimentation(mp, tps, inv_dz, *cols, cloud_vel=P.StokesRegimeVelType("f64"))
    z = cmx.column_tendencies_sedimentation(mp, tps, inv_dz, *[c[:0] for c in cols])       # no columns
    assert z.dq_rai_dt.shape == (0, 6)


def test_full_size_1e8_f32_conservation_and_sample(dev, oracle):
    """BASELINE size (1 351 351 columns × 74 levels ≈ 1e8 f32 points): finite outputs, the flux-form conservation law per column
    (Σ_k ρ Δz · sedimentation part = −surface flux) against the pointwise entry, and the oracle on a strided sample of columns."""
    import cmx
    from cmx import synthetic
    ft, n_lev = "f32", 74
    n_col = 100_000_000 // n_lev
    st = synthetic.sb2006_state(n_col * n_lev, dtype=DT[ft], device=dev, seed=2024)
    cols = [c.reshape(n_col, n_lev) for c in st]
    inv_dz = _inv_dz(n_lev, ft).to(dev)
    mp, tps = P.Microphysics2MParams(ft), P.ThermodynamicsParameters(ft)
    got = cmx.column_tendencies_sedimentation(mp, tps, inv_dz, *cols, vel=cmx.SB2006VelType, want_precip_flux=True)
    pt = cmx.bulk_microphysics_tendencies(cmx.Microphysics2Moment(), mp, tps, *st)
    torch.cuda.synchronize()
    for k in NAMES:
        assert bool(torch.isfinite(getattr(got, k)).all()), k
    rho = torch.clamp(cols[0], min=0).double()
    sed = (got.dq_rai_dt - pt.dq_rai_dt.reshape(n_col, n_lev)).double() * rho / inv_dz.double()[None, :]
    integral, mag = sed.sum(dim=1), sed.abs().sum(dim=1) + got.precip_flux.double()
    # the sedimentation part is recovered as a DIFFERENCE of two f32 tendencies: its rounding scales with the tendency itself
    slack = (pt.dq_rai_dt.reshape(n_col, n_lev).abs().double() * rho / inv_dz.double()[None, :]).sum(dim=1) * 4e-7
    assert bool(((integral + got.precip_flux.double()).abs() <= 2e-5 * mag + slack + 1e-30).all())
    idx = torch.arange(0, n_col, 997, device=dev)[:2000]
    sample = [c[idx].cpu() for c in cols]
    ref = _oracle(oracle, ft, True, "sb", False, inv_dz.cpu(), sample)
    g = {k: getattr(got, k)[idx].reshape(-1).cpu().numpy() for k in NAMES}
    parity.assert_parity(g, ref, parity.RTOL[ft], names=NAMES, what="column 1e8 f32 sample")


@pytest.mark.parametrize("ft", ["f32", "f64"])
def test_nan_rule_does_not_depend_on_tile_boundaries(dev, ft):
    """a NaN in T or q_tot poisoned the flux of the point itself inside a tile (→ the cell below got NaN) but not when the
    point was the one evaluated after a tile (halo).  One rule now: every input poisons the point's tendencies, only the flux operands
    (ρ, q_lcl, n_lcl, q_rai, n_rai) poison its fluxes — identical at a tile boundary (flat index a multiple of 128·VEC) and inside."""
    import cmx
    n_col, n_lev = 6, 1024
    vec = 4 if ft == "f32" else 1
    boundary = 128 * vec * 3
    base = _state(n_col, n_lev, ft, seed=5)
    inv_dz = _inv_dz(n_lev, ft).to(dev)
    mp, tps = P.Microphysics2MParams(ft), P.ThermodynamicsParameters(ft)
    run = lambda cols: cmx.column_tendencies_sedimentation(mp, tps, inv_dz, *[c.to(dev) for c in cols], vel=cmx.SB2006VelType,  # noqa: E731
                                                           cloud_vel=P.StokesRegimeVelType(ft))
    clean = run(base)
    for flat_idx in (boundary, boundary + 37 * vec + 1):
        for col_i, name, flux_operand in ((1, "T", False), (2, "q_tot", False), (5, "q_rai", True), (0, "rho", True)):
            cols = [c.clone() for c in base]
            cols[col_i].reshape(-1)[flat_idx] = float("nan")
            got = run(cols)
            torch.cuda.synchronize()
            for k in NAMES:
                g, c0 = getattr(got, k).reshape(-1), getattr(clean, k).reshape(-1)
                assert torch.isnan(g[flat_idx]), (name, k, flat_idx)
                assert bool(torch.isnan(g[flat_idx - 1])) == flux_operand, (name, k, flat_idx)
                mask = torch.ones_like(g, dtype=torch.bool)
                mask[flat_idx - 1:flat_idx + 1] = False
                assert torch.equal(g[mask], c0[mask]), (name, k, flat_idx)
