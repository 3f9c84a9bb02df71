"""GPU tests of the layout adapters (cmx_sb2006_warm_rain_tendencies_fields_*): ClimaCore field storage in, ClimaCore
field storage or the reference's array of NamedTuples out.  The adapters run the same per-point instruction sequence as the SoA
entry, so every comparison is BIT-exact against `bulk_microphysics_tendencies` on the gathered points."""
import numpy as np
import pytest
import torch

from cmx import parameters as P

pytestmark = pytest.mark.gpu
DT = {"f32": torch.float32, "f64": torch.float64}
NAMES = ("rho", "T", "q_tot", "q_lcl", "n_lcl", "q_rai", "n_rai")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _state_field(Nh, Nf, S, ft, dev, seed=5):
    """A ClimaCore VIJFH array (Nv, Ni, Nj, Nf, Nh) seen from C order: (Nh, Nf, S = Nv·Ni·Nj).  Components 0..6 hold the state."""
    from cmx import synthetic
    st = synthetic.sb2006_state(Nh * S, dtype=DT[ft], seed=seed)
    Y = torch.full((Nh, Nf, S), float("nan"), dtype=DT[ft])
    for f, k in enumerate(NAMES):
        Y[:, f, :] = getattr(st, k).reshape(Nh, S)
    return Y.to(dev)


@pytest.mark.parametrize("ft", ["f32", "f64"])
@pytest.mark.parametrize("shape", [(24, 9, 74 * 16), (7, 8, 63 * 9 + 1), (1, 7, 4096), (5, 7, 1)])   # RCEMIP box, odd run length, single run
def test_vijfh_fields_in_and_out(dev, ft, shape):
    import cmx
    Nh, Nf, S = shape
    Y = _state_field(Nh, Nf, S, ft, dev)
    mp, tps = P.Microphysics2MParams(ft), P.ThermodynamicsParameters(ft)
    cols = [Y[:, f, :] for f in range(7)]                                    # strided views, no copy
    assert Nf == 7 and Nh == 1 or not cols[0].is_contiguous() or Nh == 1
    ref = cmx.bulk_microphysics_tendencies(cmx.Microphysics2Moment(), mp, tps, *[c.contiguous().reshape(-1) for c in cols])
    # tendency field with its own number of components (Nf_out = 5, tendencies in components 1..4)
    Yt = torch.full((Nh, 5, S), float("nan"), dtype=DT[ft], device=dev)
    got = cmx.bulk_microphysics_tendencies_fields(cmx.Microphysics2Moment(), mp, tps, *cols, out=[Yt[:, k, :] for k in (1, 2, 3, 4)])
    torch.cuda.synchronize()
    for k, name in enumerate(("dq_lcl_dt", "dn_lcl_dt", "dq_rai_dt", "dn_rai_dt")):
        assert torch.equal(Yt[:, k + 1, :].reshape(-1), getattr(ref, name)), name
        assert getattr(got, name).data_ptr() == Yt[:, k + 1, :].data_ptr()
    assert torch.isnan(Yt[:, 0, :]).all()                                     # nothing written outside the four components
    # array-of-NamedTuples result from the same strided inputs
    aos = cmx.bulk_microphysics_tendencies_fields(cmx.Microphysics2Moment(), mp, tps, *cols, aos=True)
    assert aos.shape == (Nh * S, 8)
    for k, name in enumerate(("dq_lcl_dt", "dn_lcl_dt", "dq_rai_dt", "dn_rai_dt")):
        assert torch.equal(aos[:, k], getattr(ref, name)), name
    assert float(aos[:, 4:].abs().max()) == 0.0


@pytest.mark.parametrize("ft", ["f32", "f64"])
@pytest.mark.parametrize("limited", [True, False])
def test_aos_from_contiguous_columns_ragged_sizes(dev, ft, limited):
    import cmx
    from cmx import synthetic
    mp, tps = P.Microphysics2MParams(ft, is_limited=limited), P.ThermodynamicsParameters(ft)
    for n in (1, 3, 255, 1024, 1027, 50_001):
        st = synthetic.sb2006_state(n, dtype=DT[ft], device=dev, seed=n)
        ref = cmx.bulk_microphysics_tendencies(cmx.Microphysics2Moment(), mp, tps, *st)
        aos = cmx.bulk_microphysics_tendencies_fields(cmx.Microphysics2Moment(), mp, tps, *st, aos=True)
        for k, name in enumerate(("dq_lcl_dt", "dn_lcl_dt", "dq_rai_dt", "dn_rai_dt")):
            assert torch.equal(aos[:, k], getattr(ref, name)), (n, name)
        assert float(aos[:, 4:].abs().max()) == 0.0
        # misaligned columns (an odd slice) take the one-point-per-lane variant: same bits
        if n > 3:
            sl = [c[1:] for c in st]
            a2 = cmx.bulk_microphysics_tendencies_fields(cmx.Microphysics2Moment(), mp, tps, *sl, aos=True)
            assert torch.equal(a2, aos[1:])


def test_large_aos_and_validation(dev):
    import cmx
    from cmx import synthetic
    ft = "f32"
    mp, tps = P.Microphysics2MParams(ft), P.ThermodynamicsParameters(ft)
    n = 20_000_000
    st = synthetic.sb2006_state(n, dtype=DT[ft], device=dev, seed=9)
    ref = cmx.bulk_microphysics_tendencies(cmx.Microphysics2Moment(), mp, tps, *st)
    aos = cmx.bulk_microphysics_tendencies_fields(cmx.Microphysics2Moment(), mp, tps, *st, aos=True)
    assert torch.equal(aos[:, 0], ref.dq_lcl_dt) and torch.equal(aos[:, 3], ref.dn_rai_dt) and float(aos[:, 4:].abs().max()) == 0.0
    with pytest.raises(TypeError):
        cmx.bulk_microphysics_tendencies_fields(cmx.Microphysics2Moment(), mp, tps, *[c[::2] for c in st], aos=True)       # element stride 2
    with pytest.raises(ValueError):
        cmx.bulk_microphysics_tendencies_fields(cmx.Microphysics2Moment(), mp, tps, st[0][:10], *st[1:], aos=True)


@pytest.mark.parametrize("ft", ["f32", "f64"])
def test_mp1m_fields_and_aos(dev, ft):
    """cmx_mp1m_tendencies_fields_*: VIJFH components in place and the reference's 4-field NamedTuple rows, bit-exact vs the SoA entry;
    default and non-default option sets (compile-time and run-time flag instantiations)."""
    import cmx
    from cmx import synthetic
    tps = P.ThermodynamicsParameters(ft)
    Nh, Nf, S = 11, 9, 74 * 16
    st = synthetic.mp1m_state(Nh * S, dtype=DT[ft], seed=21)
    Y = torch.full((Nh, Nf, S), float("nan"), dtype=DT[ft])
    for f, c in enumerate(st):
        Y[:, f + 1, :] = c.reshape(Nh, S)
    Y = Y.to(dev)
    cols = [Y[:, f + 1, :] for f in range(7)]
    flat = [c.contiguous().reshape(-1) for c in cols]
    for mp in (P.Microphysics1MParams(ft), P.Microphysics1MParams(ft, snow_autoconversion=P.WithSupersaturation, snow_deposition_sublimation=P.SublimationOnly)):
        ref = cmx.bulk_microphysics_tendencies_1m(cmx.Instantaneous(), cmx.Microphysics1Moment(), mp, tps, *flat)
        Yt = torch.full((Nh, 6, S), float("nan"), dtype=DT[ft], device=dev)
        got = cmx.bulk_microphysics_tendencies_1m_fields(cmx.Instantaneous(), cmx.Microphysics1Moment(), mp, tps, *cols,
                                                         out=[Yt[:, k, :] for k in (0, 2, 3, 5)])
        aos = cmx.bulk_microphysics_tendencies_1m_fields(cmx.Instantaneous(), cmx.Microphysics1Moment(), mp, tps, *cols, aos=True)
        assert aos.shape == (Nh * S, 4)
        for k, (name, comp) in enumerate(zip(("dq_lcl_dt", "dq_icl_dt", "dq_rai_dt", "dq_sno_dt"), (0, 2, 3, 5))):
            assert torch.equal(Yt[:, comp, :].reshape(-1), getattr(ref, name)), name
            assert torch.equal(aos[:, k], getattr(ref, name)), name
            assert getattr(got, name).data_ptr() == Yt[:, comp, :].data_ptr()
        assert torch.isnan(Yt[:, 1, :]).all() and torch.isnan(Yt[:, 4, :]).all()
    for n in (1, 5, 1023, 4099):      # ragged contiguous sizes through the AoS path
        s1 = synthetic.mp1m_state(n, dtype=DT[ft], device=dev, seed=n)
        mp = P.Microphysics1MParams(ft)
        ref = cmx.bulk_microphysics_tendencies_1m(cmx.Instantaneous(), cmx.Microphysics1Moment(), mp, tps, *s1)
        aos = cmx.bulk_microphysics_tendencies_1m_fields(cmx.Instantaneous(), cmx.Microphysics1Moment(), mp, tps, *s1, aos=True)
        assert torch.equal(aos[:, 0], ref.dq_lcl_dt) and torch.equal(aos[:, 3], ref.dq_sno_dt)


@pytest.mark.parametrize("ft", ["f32", "f64"])
def test_mp1m_linearized_average_fields_and_aos(dev, ft):
    """cmx_mp1m_linearized_average_fields_* (round 3): the operational LinearizedAverage mode on VIJFH components in place and as the
    reference's NamedTuple rows, bit-exact vs the SoA entry; ragged sizes through the AoS path."""
    import cmx
    from cmx import synthetic
    tps, mp = P.ThermodynamicsParameters(ft), P.Microphysics1MParams(ft)
    mode, scheme = cmx.LinearizedAverage(), cmx.Microphysics1Moment()
    Nh, Nf, S = 7, 9, 37 * 16
    st = synthetic.mp1m_state(Nh * S, dtype=DT[ft], seed=22)
    Y = torch.full((Nh, Nf, S), float("nan"), dtype=DT[ft])
    for f, c in enumerate(st):
        Y[:, f + 1, :] = c.reshape(Nh, S)
    Y = Y.to(dev)
    cols = [Y[:, f + 1, :] for f in range(7)]
    flat = [c.contiguous().reshape(-1) for c in cols]
    for dt, nsub in ((20.0, 1), (60.0, 3)):
        ref = cmx.bulk_microphysics_tendencies_1m(mode, scheme, mp, tps, *flat, dt, nsub)
        Yt = torch.full((Nh, 6, S), float("nan"), dtype=DT[ft], device=dev)
        cmx.bulk_microphysics_tendencies_1m_fields(mode, scheme, mp, tps, *cols, dt, nsub, out=[Yt[:, k, :] for k in (0, 2, 3, 5)])
        aos = cmx.bulk_microphysics_tendencies_1m_fields(mode, scheme, mp, tps, *cols, dt, nsub, aos=True)
        for k, (name, comp) in enumerate(zip(("dq_lcl_dt", "dq_icl_dt", "dq_rai_dt", "dq_sno_dt"), (0, 2, 3, 5))):
            assert torch.equal(Yt[:, comp, :].reshape(-1), getattr(ref, name)), name
            assert torch.equal(aos[:, k], getattr(ref, name)), name
        assert torch.isnan(Yt[:, 1, :]).all() and torch.isnan(Yt[:, 4, :]).all()
    for n in (1, 5, 1023, 4099):
        s1 = synthetic.mp1m_state(n, dtype=DT[ft], device=dev, seed=n)
        ref = cmx.bulk_microphysics_tendencies_1m(mode, scheme, mp, tps, *s1, 30.0, 2)
        aos = cmx.bulk_microphysics_tendencies_1m_fields(mode, scheme, mp, tps, *s1, 30.0, 2, aos=True)
        assert torch.equal(aos[:, 0], ref.dq_lcl_dt) and torch.equal(aos[:, 3], ref.dq_sno_dt)
    with pytest.raises(ValueError):
        cmx.bulk_microphysics_tendencies_1m_fields(mode, scheme, mp, tps, *flat)            # LinearizedAverage needs dt


# ---- the layout entries against the ORACLE (not only against the SoA kernel) ------------------------------------------------------
# Reference layouts: ClimaCore fields of test/gpu_clima_core_test.jl:16-30,100-114 (VIJFH storage) and the Vector{NamedTuple} result of
# benchmark_2m_bulk_tendencies_kernel! / benchmark_1m_bulk_tendencies_kernel! (test/gpu_performance.jl:138-182,212-216).
@pytest.mark.parametrize("ft", ["f32", "f64"])
@pytest.mark.parametrize("layout", ["vijfh", "aos"])
def test_2m_layout_entries_match_oracle(dev, oracle, ft, layout):
    import cmx
    import parity
    from cmx import _abi
    Nh, Nf, S = 41, 9, 74 * 16 + (0 if layout == "vijfh" else 3)      # the AoS case also gets a run length that is not a multiple of 4
    Y = _state_field(Nh, Nf, S, ft, dev, seed=77)
    mp, tps = P.Microphysics2MParams(ft), P.ThermodynamicsParameters(ft)
    cols = [Y[:, f, :] for f in range(7)]
    names = ("dq_lcl_dt", "dn_lcl_dt", "dq_rai_dt", "dn_rai_dt")
    if layout == "vijfh":
        Yt = torch.full((Nh, 5, S), float("nan"), dtype=DT[ft], device=dev)
        cmx.bulk_microphysics_tendencies_fields(cmx.Microphysics2Moment(), mp, tps, *cols, out=[Yt[:, k, :] for k in (1, 2, 3, 4)])
        torch.cuda.synchronize()
        got = {name: Yt[:, k + 1, :].reshape(-1).cpu().numpy() for k, name in enumerate(names)}
    else:
        aos = cmx.bulk_microphysics_tendencies_fields(cmx.Microphysics2Moment(), mp, tps, *cols, aos=True)
        torch.cuda.synchronize()
        got = {name: aos[:, k].cpu().numpy() for k, name in enumerate(names)}
        assert float(aos[:, 4:].abs().max()) == 0.0                     # the four identically-zero ice fields, BMT:852-853
    cols_np = [c.contiguous().reshape(-1).cpu().numpy().astype(np.float64) for c in cols]
    ref = oracle.sb2006_warm_rain_tendencies(_abi.F64, P.WarmRainParams2M("f64").c, P.ThermodynamicsParameters("f64"), None,
                                             _abi.CMX_SB2006_LIMITED, *cols_np, float32_gates=(ft == "f32"), nthreads=8,
                                             branch_margin=1e-5 if ft == "f32" else 1e-11)
    rep = parity.assert_parity(got, ref, parity.RTOL[ft], names=names, what=f"2M {layout} {ft}")
    print(f"\n[layout parity] 2M {layout} {ft} n={Nh * S}: {rep}")


@pytest.mark.parametrize("ft", ["f32", "f64"])
@pytest.mark.parametrize("layout", ["vijfh", "aos"])
def test_1m_layout_entries_match_oracle(dev, oracle, ft, layout):
    import cmx
    import parity
    from cmx import _abi, synthetic
    tps = P.ThermodynamicsParameters(ft)
    Nh, Nf, S = 37, 9, 74 * 16
    st = synthetic.mp1m_state(Nh * S, dtype=DT[ft], seed=78)
    Y = torch.full((Nh, Nf, S), float("nan"), dtype=DT[ft])
    for f, c in enumerate(st):
        Y[:, f + 1, :] = c.reshape(Nh, S)
    Y = Y.to(dev)
    cols = [Y[:, f + 1, :] for f in range(7)]
    mp = P.Microphysics1MParams(ft)
    names = ("dq_lcl_dt", "dq_icl_dt", "dq_rai_dt", "dq_sno_dt")
    if layout == "vijfh":
        Yt = torch.full((Nh, 6, S), float("nan"), dtype=DT[ft], device=dev)
        cmx.bulk_microphysics_tendencies_1m_fields(cmx.Instantaneous(), cmx.Microphysics1Moment(), mp, tps, *cols,
                                                   out=[Yt[:, k, :] for k in (0, 2, 3, 5)])
        torch.cuda.synchronize()
        got = {name: Yt[:, comp, :].reshape(-1).cpu().numpy() for name, comp in zip(names, (0, 2, 3, 5))}
    else:
        aos = cmx.bulk_microphysics_tendencies_1m_fields(cmx.Instantaneous(), cmx.Microphysics1Moment(), mp, tps, *cols, aos=True)
        torch.cuda.synchronize()
        got = {name: aos[:, k].cpu().numpy() for k, name in enumerate(names)}
    mp64 = P.Microphysics1MParams("f64")
    ref = oracle.mp1m(_abi.F64, mp64.c, P.ThermodynamicsParameters("f64"), mp64.flags, *[c.numpy().astype(np.float64) for c in st],
                      float32_gates=(ft == "f32"), nthreads=8, want_sources=False)
    tf = P.DEFAULT_PARAMETERS["temperature_water_freeze"]
    ref["near_branch"] = np.abs(st[1].numpy().astype(np.float64) - tf) < (1e-4 if ft == "f32" else 1e-11)      # is_warm routing, BMT:171
    rep = parity.assert_parity(got, ref, parity.RTOL[ft], names=names, what=f"1M {layout} {ft}")
    print(f"\n[layout parity] 1M {layout} {ft} n={Nh * S}: {rep}")
