"""CPU tests of the column-step oracle (oracle/cmx_oracle_column_impl.h).  The flux scheme is the host model's, not the
reference's (parity unpinned for that step), so the oracle is checked against (i) an independent numpy restatement built from the
PINNED pointwise oracle outputs, (ii) the conservation law of a flux-form scheme, (iii) its limits."""
import numpy as np
import pytest

from cmx import _abi
from cmx import parameters as P
from cmx import synthetic

F64 = _abi.F64


def _state(n_col, n_lev, seed=3):
    st = synthetic.sb2006_state(n_col * n_lev, seed=seed)
    return [c.numpy().astype(np.float64).reshape(n_col, n_lev) for c in st]


def _dz(n_lev, seed=0):
    rng = np.random.default_rng(seed)
    return 1.0 / rng.uniform(30.0, 500.0, n_lev)


@pytest.mark.parametrize("limited", [True, False])
@pytest.mark.parametrize("vel", ["sb", "chen"])
@pytest.mark.parametrize("cloud", [False, True])
def test_column_oracle_is_pointwise_oracle_plus_upwind_divergence(oracle, limited, vel, cloud):
    n_col, n_lev = 57, 23
    cols = _state(n_col, n_lev)
    inv_dz = _dz(n_lev)
    wr, tps, rv = P.WarmRainParams2M("f64", limited).c, P.ThermodynamicsParameters("f64"), P.rain_vel_params("f64")
    stokes = P.StokesRegimeVelType("f64") if cloud else None
    flags = (_abi.CMX_SB2006_LIMITED if limited else 0) | (_abi.CMX_VEL_SB2006 if vel == "sb" else _abi.CMX_VEL_CHEN2022)
    got = oracle.sb2006_column_tendencies_sedimentation(F64, wr, tps, rv, stokes, flags, inv_dz, *cols)
    # independent restatement: pointwise tendencies and fall speeds from the pinned oracle, divergence in numpy
    pt = oracle.sb2006_warm_rain_tendencies(F64, wr, tps, rv, flags, *[c.reshape(-1) for c in cols])
    rho, _, _, q_lcl, n_lcl, q_rai, n_rai = [np.maximum(c, 0.0) for c in cols]
    sh = (n_col, n_lev)
    Fq = rho * q_rai * pt["vt_rai_m"].reshape(sh)
    Fn = rho * n_rai * pt["vt_rai_n"].reshape(sh)
    up = lambda F: np.concatenate([F[:, 1:], np.zeros((n_col, 1))], axis=1)  # noqa: E731
    w = inv_dz[None, :] / rho
    np.testing.assert_allclose(got["dq_rai_dt"].reshape(sh), pt["dq_rai_dt"].reshape(sh) + (up(Fq) - Fq) * w, rtol=1e-13, atol=1e-300)
    np.testing.assert_allclose(got["dn_rai_dt"].reshape(sh), pt["dn_rai_dt"].reshape(sh) + (up(Fn) - Fn) * w, rtol=1e-13, atol=1e-300)
    np.testing.assert_array_equal(got["precip_flux"], Fq[:, 0])
    if cloud:
        pdf_c = P.SB2006("f64", limited).pdf_c
        cv = oracle.sb2006_cloud_terminal_velocity(F64, pdf_c, stokes, q_lcl.reshape(-1), rho.reshape(-1), (rho * n_lcl).reshape(-1))
        Fql = rho * q_lcl * np.asarray(cv[1]).reshape(sh)
        Fnl = rho * n_lcl * np.asarray(cv[0]).reshape(sh)
        np.testing.assert_allclose(got["dq_lcl_dt"].reshape(sh), pt["dq_lcl_dt"].reshape(sh) + (up(Fql) - Fql) * w, rtol=1e-13, atol=1e-300)
        np.testing.assert_allclose(got["dn_lcl_dt"].reshape(sh), pt["dn_lcl_dt"].reshape(sh) + (up(Fnl) - Fnl) * w, rtol=1e-13, atol=1e-300)
        assert np.abs(Fql).max() > 0
    else:
        np.testing.assert_array_equal(got["dq_lcl_dt"], pt["dq_lcl_dt"])
        np.testing.assert_array_equal(got["dn_lcl_dt"], pt["dn_lcl_dt"])
    assert np.abs(Fq).max() > 0 and np.all(got["scale"]["dq_rai_dt"] >= pt["scale"]["dq_rai_dt"])


def test_flux_form_conserves_the_column_integral(oracle):
    """Σ_k ρ_k Δz_k · (sedimentation part of ∂q_rai/∂t) = −F_0: what leaves the column is exactly the surface precipitation flux."""
    n_col, n_lev = 40, 74
    cols = _state(n_col, n_lev, seed=8)
    inv_dz = _dz(n_lev, seed=1)
    wr, tps, rv = P.WarmRainParams2M("f64").c, P.ThermodynamicsParameters("f64"), P.rain_vel_params("f64")
    flags = _abi.CMX_SB2006_LIMITED | _abi.CMX_VEL_SB2006
    got = oracle.sb2006_column_tendencies_sedimentation(F64, wr, tps, rv, None, flags, inv_dz, *cols)
    pt = oracle.sb2006_warm_rain_tendencies(F64, wr, tps, rv, flags, *[c.reshape(-1) for c in cols])
    sed = (got["dq_rai_dt"] - pt["dq_rai_dt"]).reshape(n_col, n_lev)
    rho = np.maximum(cols[0], 0.0)
    integral = (sed * rho / inv_dz[None, :]).sum(axis=1)
    scale = np.abs(sed * rho / inv_dz[None, :]).sum(axis=1) + got["precip_flux"]
    assert np.all(np.abs(integral + got["precip_flux"]) <= 1e-9 * scale + 1e-300)
    assert got["precip_flux"].max() > 0 and np.all(got["precip_flux"] >= 0)


def test_single_level_columns_and_dry_columns(oracle):
    wr, tps, rv = P.WarmRainParams2M("f64").c, P.ThermodynamicsParameters("f64"), P.rain_vel_params("f64")
    flags = _abi.CMX_SB2006_LIMITED | _abi.CMX_VEL_SB2006
    cols = _state(31, 1, seed=4)                                      # n_lev = 1: the only level is top and bottom at once
    got = oracle.sb2006_column_tendencies_sedimentation(F64, wr, tps, rv, None, flags, np.array([0.01]), *cols)
    pt = oracle.sb2006_warm_rain_tendencies(F64, wr, tps, rv, flags, *[c.reshape(-1) for c in cols])
    rho, q_rai = np.maximum(cols[0], 0).reshape(-1), np.maximum(cols[5], 0).reshape(-1)
    np.testing.assert_allclose(got["dq_rai_dt"], pt["dq_rai_dt"] - q_rai * pt["vt_rai_m"] * 0.01, rtol=1e-13, atol=1e-300)
    cols = _state(5, 9, seed=5)
    cols[5][:] = 0.0                                                  # no rain anywhere: the column step is the pointwise step
    got = oracle.sb2006_column_tendencies_sedimentation(F64, wr, tps, rv, None, flags, _dz(9), *cols)
    pt = oracle.sb2006_warm_rain_tendencies(F64, wr, tps, rv, flags, *[c.reshape(-1) for c in cols])
    for k in ("dq_lcl_dt", "dn_lcl_dt", "dq_rai_dt"):
        np.testing.assert_array_equal(got[k], pt[k])
    assert np.all(got["precip_flux"] == 0)
