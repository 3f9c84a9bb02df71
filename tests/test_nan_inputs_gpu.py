"""NaN inputs.  The reference sanitises with Julia's `max(0, x)` (= NaN for a NaN x, src/Utilities.jl:296) and carries the NaN to the
tendencies; the device clamps use v_max, which would return 0.  The bulk-tendency entries therefore poison every output of a point
whose inputs hold a NaN, and leave all other points untouched (bit-identical to a run without the NaN)."""
import pytest
import torch

from cmx import parameters as P

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda", 0)


def _nan(t, negative):
    """A quiet NaN of t's dtype, with the sign bit set if `negative`: the Float32 max0() is an integer max on the bits, which turns a
    sign-bit NaN into +0 — the any_nan poison of the entries must catch it all the same."""
    x = torch.full((1,), float("nan"), dtype=t.dtype, device=t.device)
    return -x if negative else x                 # negation flips the sign bit of a NaN (IEEE 754 §5.5.1: a sign-bit operation)


def _check(call, cols, n_out, lo=0, negative_nan=False):
    clean = call(cols)
    n = cols[0].numel()
    for k in range(len(cols)):
        bad = [c.clone() for c in cols]
        idx = torch.arange(k, n, 37, device=cols[0].device)
        bad[k][idx] = _nan(bad[k], negative_nan)
        if negative_nan:
            assert bool(torch.signbit(bad[k][idx]).all())
        out = call(bad)
        mask = torch.zeros(n, dtype=torch.bool, device=cols[0].device)
        mask[idx] = True
        mask = mask[lo:]
        for a, b in zip(list(out)[:n_out], list(clean)[:n_out]):
            assert bool(torch.isnan(a[mask]).all()), f"input column {k}: NaN not propagated"
            assert torch.equal(torch.nan_to_num(a[~mask], nan=-7.0), torch.nan_to_num(b[~mask], nan=-7.0)), f"input column {k}: clean points changed"


@pytest.mark.parametrize("sfx", ["f32", "f64"])
def test_nan_in_any_input_poisons_the_point(dev, sfx):
    import cmx
    from cmx import synthetic
    dt = torch.float32 if sfx == "f32" else torch.float64
    n = 10_007
    tps = P.ThermodynamicsParameters(sfx)
    mp2, mp1, mp0 = P.Microphysics2MParams(sfx), P.Microphysics1MParams(sfx), P.Microphysics0MParams(sfx)
    st2 = [c.clone() for c in synthetic.sb2006_state(n, dtype=dt, device=dev, seed=21)]
    st1 = [c.clone() for c in synthetic.mp1m_state(n, dtype=dt, device=dev, seed=22)]
    s2, s1, s0 = cmx.Microphysics2Moment(), cmx.Microphysics1Moment(), cmx.Microphysics0Moment()
    _check(lambda c: cmx.bulk_microphysics_tendencies(s2, mp2, tps, *c, vel=cmx.SB2006VelType), st2, 6)
    _check(lambda c: cmx.bulk_microphysics_tendencies(s2, mp2, tps, *[x[1:] for x in c]), st2, 4, lo=1)     # one-point-per-lane path
    _check(lambda c: cmx.bulk_microphysics_tendencies_fields(s2, mp2, tps, *c), st2, 4)
    _check(lambda c: cmx.bulk_microphysics_tendencies_1m(cmx.Instantaneous(), s1, mp1, tps, *c), st1, 4)
    _check(lambda c: cmx.bulk_microphysics_tendencies_1m(cmx.LinearizedAverage(), s1, mp1, tps, *c, 10.0, 2), st1, 4)
    _check(lambda c: cmx.bulk_microphysics_tendencies_1m_fields(cmx.Instantaneous(), s1, mp1, tps, *c), st1, 4)
    _check(lambda c: (cmx.bulk_microphysics_tendencies_0m(s0, mp0, tps, c[0], c[0], c[1]),), [st1[3], st1[4]], 1)
    _check(lambda c: (cmx.bulk_microphysics_tendencies_0m(s0, mp0, tps, c[0], c[0], c[1], c[2]),), [st1[3], st1[4], st1[2]], 1)


@pytest.mark.parametrize("sfx", ["f32", "f64"])
def test_sign_bit_nan_is_poison_too(dev, sfx):
    """A NaN whose sign bit is set: the integer-max clamp of the Float32 kernels would turn it into +0; the entries' NaN rule looks at the raw
    inputs (unordered compares), so the point is poisoned — in the tendencies entries AND in the 1-moment source-term entry, which had no
    poison before round 4."""
    import cmx
    from cmx import synthetic
    dt = torch.float32 if sfx == "f32" else torch.float64
    n = 4099
    tps = P.ThermodynamicsParameters(sfx)
    mp2, mp1 = P.Microphysics2MParams(sfx), P.Microphysics1MParams(sfx)
    st2 = [c.clone() for c in synthetic.sb2006_state(n, dtype=dt, device=dev, seed=51)]
    st1 = [c.clone() for c in synthetic.mp1m_state(n, dtype=dt, device=dev, seed=52)]
    s2, s1 = cmx.Microphysics2Moment(), cmx.Microphysics1Moment()
    _check(lambda c: cmx.bulk_microphysics_tendencies(s2, mp2, tps, *c, vel=cmx.SB2006VelType), st2, 6, negative_nan=True)
    _check(lambda c: cmx.bulk_microphysics_tendencies_1m(cmx.Instantaneous(), s1, mp1, tps, *c), st1, 4, negative_nan=True)
    src = lambda c: tuple(v for v in cmx.microphysics_source_terms_1m(mp1, tps, *c) if v is not None)  # noqa: E731
    n_src = len(src(st1))
    _check(src, st1, n_src, negative_nan=True)
    _check(src, st1, n_src)


def test_nan_in_the_2m_p3_entry(dev):
    import cmx
    from cmx import synthetic
    n = 2048
    mp, tps = P.Microphysics2MParams("f64", with_ice=True), P.ThermodynamicsParameters("f64")
    st = [c.clone() for c in synthetic.sb2006_state(n, dtype=torch.float64, device=dev, seed=31)]
    p3 = synthetic.p3_state(n, dtype=torch.float64, device=dev, seed=32)
    rho = st[0]
    ice = [p3.rho_q_ice / rho, p3.rho_n_ice / rho, p3.rho_q_rim / rho, p3.rho_b_rim / rho]
    ll = cmx.p3_shape(P.ParametersP3("f64"), *p3, want=("log_lambda",)).log_lambda
    _check(lambda c: cmx.bulk_microphysics_tendencies(cmx.Microphysics2Moment(), mp, tps, *c, ll), st + ice, 8)


def test_nan_in_the_other_entries(dev):
    """ARG2000 (a NaN reaches S_max and through it every mode), the P3 shape solver and the SB2006 per-process entry."""
    import cmx
    from cmx import synthetic
    sfx, dt, n = "f32", torch.float32, 4099
    tps = P.ThermodynamicsParameters(sfx)
    ap, aip, ad = P.AerosolActivationParameters(sfx), P.AirProperties(sfx), synthetic.arg_config3_distribution()
    sta = [c.clone() for c in synthetic.arg_state(n, dtype=dt, device=dev, seed=41)]

    def arg(c):
        r = cmx.aerosol_activation(ap, ad, aip, tps, *c, want=("N_act", "M_act", "S_max"))
        return tuple(r.N_act) + tuple(r.M_act) + (r.S_max,)
    _check(arg, sta, 11)
    _check(lambda c: tuple(cmx.aerosol_activation(ap, ad, aip, tps, *c).N_act), sta, 5)                  # the number-only instantiation
    p3 = [c.clone() for c in synthetic.p3_state(n, dtype=torch.float64, device=dev, seed=42)]
    _check(lambda c: tuple(cmx.p3_shape(P.ParametersP3("f64"), *c, want=("F_rim", "rho_rim", "log_lambda", "D_m", "log_N0"))), p3, 5)
    st2 = [c.clone() for c in synthetic.sb2006_state(n, dtype=dt, device=dev, seed=43)]
    rho, T, q_tot, q_lcl, n_lcl, q_rai, n_rai = st2
    mp2 = P.Microphysics2MParams(sfx)
    _check(lambda c: tuple(cmx.sb2006_process_rates(mp2, tps, *c)), [q_tot, q_lcl, q_rai, n_lcl * rho, n_rai * rho, rho, T], 17)


@pytest.mark.parametrize("sfx", ["f32", "f64"])
def test_zero_and_negative_air_density(dev, sfx):
    """ρ ≤ 0 (clamped to 0 like every negative input) is outside the entries' domain — include/cmx.h "Conventions" states ρ > 0 — but the
    result must not look valid: Float32 reproduces the reference's own finite / NaN / ±Inf pattern output by output (its
    hardware reciprocals and logarithms handle 0 and Inf like IEEE division); Float64 — whose finite-argument forms assume
    ρ > 0 — returns a non-finite value wherever the reference does, and may return NaN where the reference still returns a finite number
    or a signed infinity (the 1-moment entries: NaN in every output, test_float64_1m_entries_poison_nonpositive_air_density).  The SB2006 kernel takes ρ^(-1/2) with the full form in both float types (+Inf at ρ = 0, never NaN from the seed)."""
    import numpy as np

    import cmx
    import oracle_binding as ob
    from cmx import _abi
    dt = torch.float32 if sfx == "f32" else torch.float64
    rows2 = [(rho, T, qt, ql, nl, qr, nr) for rho in (0.0, -1.0) for (T, qt, ql, nl, qr, nr) in
             ((290.0, 7e-3, 1e-3, 1e8, 5e-3, 1e4), (290.0, 7e-3, 0.0, 0.0, 0.0, 0.0), (250.0, 1e-4, 1e-3, 1e8, 0.0, 0.0), (300.0, 3e-2, 0.0, 0.0, 2e-3, 5e3))]
    rows1 = [(rho, T, qt, ql, qi, qr, qs) for rho in (0.0, -1.0) for (T, qt, ql, qi, qr, qs) in
             ((290.0, 1.5e-2, 1e-3, 0.0, 5e-3, 0.0), (260.0, 3e-3, 1e-4, 2e-4, 1e-4, 3e-3), (275.0, 4e-3, 0.0, 1e-4, 0.0, 1e-3), (240.0, 3e-4, 0.0, 0.0, 0.0, 0.0))]
    arr2, arr1 = np.array(rows2).T, np.array(rows1).T

    def cls(a):
        a = np.asarray(a, dtype=np.float64)
        return np.where(np.isnan(a), 3, np.where(np.isposinf(a), 2, np.where(np.isneginf(a), 1, 0)))

    got2 = cmx.bulk_microphysics_tendencies(cmx.Microphysics2Moment(), P.Microphysics2MParams(sfx), P.ThermodynamicsParameters(sfx),
                                            *[torch.tensor(a, dtype=dt, device=dev) for a in arr2], vel=cmx.SB2006VelType)
    ref2 = ob.sb2006_warm_rain_tendencies(_abi.F64, P.WarmRainParams2M("f64").c, P.ThermodynamicsParameters("f64"), P.rain_vel_params("f64"),
                                          _abi.CMX_SB2006_LIMITED | _abi.CMX_VEL_SB2006, *arr2, float32_gates=(sfx == "f32"), nthreads=1)
    mp = P.Microphysics1MParams(sfx)
    got1 = cmx.bulk_microphysics_tendencies_1m(cmx.Instantaneous(), cmx.Microphysics1Moment(), mp, P.ThermodynamicsParameters(sfx),
                                               *[torch.tensor(a, dtype=dt, device=dev) for a in arr1])
    mp64 = P.Microphysics1MParams("f64")
    ref1 = ob.mp1m(_abi.F64, mp64.c, P.ThermodynamicsParameters("f64"), mp64.flags, *arr1, nthreads=1, float32_gates=(sfx == "f32"))
    for got, ref in ((got2, ref2), (got1, ref1)):
        for k, v in got._asdict().items():
            g, r = cls(v.cpu().numpy()), cls(ref[k])
            if sfx == "f32":
                assert np.array_equal(g, r), (k, g, r)
            else:
                assert np.all((g != 0) | (r == 0)), (k, g, r)            # non-finite wherever the reference is non-finite
                assert np.all((g == r) | (g == 3)), (k, g, r)            # and otherwise the reference's class, or NaN


def test_float64_1m_entries_poison_nonpositive_air_density(dev):
    """The Float64 1-moment entries (Instantaneous, LinearizedAverage, source terms): ρ ≤ 0 → NaN in EVERY output of the point (cmx_math.hpp
    bad_density, include/cmx.h "Conventions"); the points beside it are bit-identical to a run without it."""
    import cmx
    from cmx import synthetic
    n = 4096
    st = list(synthetic.mp1m_state(n, dtype=torch.float64, device=dev, seed=5))
    mp, tps = P.Microphysics1MParams("f64"), P.ThermodynamicsParameters("f64")
    calls = {"instantaneous": lambda c: tuple(cmx.bulk_microphysics_tendencies_1m(cmx.Instantaneous(), cmx.Microphysics1Moment(), mp, tps, *c)),
             "linearized": lambda c: tuple(cmx.bulk_microphysics_tendencies_1m(cmx.LinearizedAverage(), cmx.Microphysics1Moment(), mp, tps, *c, dt=60.0, nsub=2)),
             "source terms": lambda c: tuple(cmx.microphysics_source_terms_1m(mp, tps, *c))}
    idx = torch.arange(3, n, 41, device=dev)
    mask = torch.zeros(n, dtype=torch.bool, device=dev)
    mask[idx] = True
    for name, call in calls.items():
        clean = call(st)
        for bad_rho in (0.0, -1.0, -0.0):
            cols = [c.clone() for c in st]
            cols[0][idx] = bad_rho
            out = call(cols)
            for a, b in zip(out, clean):
                assert bool(torch.isnan(a[mask]).all()), (name, bad_rho)
                assert torch.equal(torch.nan_to_num(a[~mask], nan=-7.0), torch.nan_to_num(b[~mask], nan=-7.0)), (name, bad_rho)


def test_float64_1m_fall_speed_entries_poison_nonpositive_air_density(dev):
    """… and the 1-moment fall-speed entries (terminal_velocity_1m, sedimentation_velocities): the floored logarithms of the Float64 fall speeds would return finite
    numbers at ρ = 0, where the reference's are ±Inf or NaN."""
    import cmx
    n = 512
    g = torch.Generator(device="cpu").manual_seed(3)
    q = [(1e-4 * torch.rand(n, dtype=torch.float64, generator=g)).to(dev) for _ in range(4)]
    rho = (0.4 + torch.rand(n, dtype=torch.float64, generator=g)).to(dev)
    mp = P.Microphysics1MParams("f64")
    vels = (P.StokesRegimeVelType("f64"), P.Chen2022VelTypeRain("f64"), P.Chen2022VelTypeIce("f64"))
    idx = torch.arange(5, n, 17, device=dev)
    mask = torch.zeros(n, dtype=torch.bool, device=dev)
    mask[idx] = True
    calls = {"terminal_velocity_1m": lambda r: tuple(v for v in cmx.terminal_velocity_1m(mp, r, q[2], q[3], chen=True) if v is not None),
             "sedimentation_velocities": lambda r: tuple(cmx.sedimentation_velocities(mp, *vels, r, *q))}
    for name, call in calls.items():
        clean = call(rho)
        for bad_rho in (0.0, -1.0):
            r = rho.clone()
            r[idx] = bad_rho
            for a, b in zip(call(r), clean):
                assert bool(torch.isnan(a[mask]).all()), (name, bad_rho)
                assert torch.equal(a[~mask], b[~mask]), (name, bad_rho)
