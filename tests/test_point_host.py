"""CPU check of the DEVICE point functions' algebra: csrc/cmx_mp1m.hpp / cmx_mp1m_vel.hpp compiled by g++ for the host
(tests/native/point_host.cpp, CMX_HOST_BUILD: libm stand-ins for v_exp_f32 / v_log_f32 / v_rcp_f32) against the oracle on the same
random states the GPU parity tests use.  This is test infrastructure — it is not linked into libcmx.so, nothing in the package loads
it, and it proves formulas (host-folded constants, log2-domain rewrites, gates), not device bits: the -m gpu suite is the parity gate."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np
import pytest
import torch

import parity
from cmx import _abi, synthetic
from cmx import parameters as P
from test_mp1m_linearized import NAMES, Q_MIN, _random_state, check_linearized_parity

REPO = Path(__file__).resolve().parent.parent
NPT = {"f32": np.float32, "f64": np.float64}
CT = {"f32": C.c_float, "f64": C.c_double}
T_FREEZE = P.DEFAULT_PARAMETERS["temperature_water_freeze"]
OPTION_SETS = {
    "default": {},
    "alt": dict(snow_autoconversion=P.WithSupersaturation(), snow_deposition_sublimation=P.SublimationOnly(),
                rain_autoconversion=P.PrescribedNd()),
    "sparse": dict(rain_snow_accretion=None, cloud_ice_melt=None, cloud_liquid_snow_accretion=None, snow_melt=None),
    "tdep": dict(cloud_ice_formation=P.TemperatureDependent()),
}


@pytest.fixture(scope="module")
def host():
    out = REPO / "tests" / "native" / "_build"
    out.mkdir(exist_ok=True)
    so = out / "libpoint_host.so"
    subprocess.run(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-o", str(so),
                    str(REPO / "tests" / "native" / "point_host.cpp")], check=True)
    return C.CDLL(str(so))


def _ptrs(arrs, ft):
    return (C.POINTER(CT[ft]) * len(arrs))(*[a.ctypes.data_as(C.POINTER(CT[ft])) for a in arrs])


def _call(host, name, ft, head, cols, nout):
    x = [np.ascontiguousarray(c, dtype=NPT[ft]) for c in cols]
    y = [np.empty_like(x[0]) for _ in range(nout)]
    fn = getattr(host, f"{name}_{ft}")
    fn.restype = C.c_int32
    rc = fn(*head, C.c_int64(x[0].size), _ptrs(x, ft), _ptrs(y, ft))
    return rc, y


def _oracle_1m(oracle, ft, opts, cols, **kw):
    mp = P.Microphysics1MParams("f64", **opts)
    r = oracle.mp1m(_abi.F64, mp.c, P.ThermodynamicsParameters("f64"), mp.flags, *[np.asarray(c, dtype=np.float64) for c in cols],
                    float32_gates=(ft == "f32"), nthreads=8, **kw)
    r["near_branch"] = np.abs(np.asarray(cols[1], dtype=np.float64) - T_FREEZE) < (1e-4 if ft == "f32" else 1e-11)
    return r


@pytest.mark.parametrize("ft", ["f32", "f64"])
@pytest.mark.parametrize("optset", list(OPTION_SETS))
def test_mp1m_tendencies_and_sources(host, oracle, ft, optset):
    opts = OPTION_SETS[optset]
    n = 60_000
    st = [c.numpy() for c in synthetic.mp1m_state(n, dtype=getattr(torch, {"f32": "float32", "f64": "float64"}[ft]), seed=1234)]
    mp, tps = P.Microphysics1MParams(ft, **opts), P.ThermodynamicsParameters(ft)
    head = (C.byref(mp.c), C.byref(tps), C.c_uint32(mp.flags))
    rc, tend = _call(host, "host_mp1m_tendencies", ft, head, st, 4)
    assert rc == (1 if optset == "default" else 0)          # the default set takes the default-exponent instantiation
    _, src = _call(host, "host_mp1m_sources", ft, head, st, _abi.CMX_MP1M_NSRC)
    ref = _oracle_1m(oracle, ft, opts, st)
    rep = parity.assert_parity(dict(zip(NAMES, tend)), ref, parity.RTOL[ft], names=list(NAMES), what=f"host-build 1M {ft} {optset}")
    keep = ~ref["near_branch"]
    cancel = {"S_phase_change_vap_lcl": "dq_lcl_dt", "S_phase_change_vap_icl": "dq_icl_dt", "S_phase_change_vap_rai": "dq_rai_dt",
              "S_phase_change_vap_sno": "dq_sno_dt", "S_melt_icl_lcl": "dq_icl_dt", "S_melt_sno_rai": "dq_sno_dt",
              "S_accr_melt_lcl_sno": "dq_sno_dt", "S_accr_melt_rai_sno": "dq_sno_dt", "S_acnv_icl_sno": "dq_icl_dt",
              "S_acnv_lcl_rai": "dq_lcl_dt"}
    for k, got in zip(_abi.MP1M_SOURCE_COLUMNS, src):
        sc = ref["scale"][cancel[k]] if k in cancel else None
        e = parity.scaled_err(got, ref["sources"][k], sc, parity.FLOOR[ft], parity.CEIL[ft], parity.CTOL[ft] / parity.RTOL[ft])[keep]
        assert float(np.nan_to_num(e, nan=np.inf).max()) <= parity.RTOL[ft], (k, optset, ft)
    print(f"\n[host build, 1M {ft} {optset}] worst normalised error per tendency {rep}")


def test_mp1m_general_exponent_instantiation_matches_default(host, oracle):
    """A parameter set one ulp away from the default exponents takes the general exp2(e·log2 λ⁻¹) forms: same numbers."""
    n = 20_000
    st = [c.numpy() for c in synthetic.mp1m_state(n, dtype=torch.float64, seed=7)]
    tps = P.ThermodynamicsParameters("f64")
    mp = P.Microphysics1MParams("f64")
    head = (C.byref(mp.c), C.byref(tps), C.c_uint32(mp.flags))
    rc0, a = _call(host, "host_mp1m_tendencies", "f64", head, st, 4)
    mp2 = P.Microphysics1MParams("f64")
    mp2.c.vel_rain.ve = np.nextafter(mp2.c.vel_rain.ve, 1.0)
    rc1, b = _call(host, "host_mp1m_tendencies", "f64", (C.byref(mp2.c), C.byref(tps), C.c_uint32(mp2.flags)), st, 4)
    assert (rc0, rc1) == (1, 0)
    for x, y in zip(a, b):
        np.testing.assert_allclose(x, y, rtol=1e-9, atol=1e-30)


@pytest.mark.parametrize("ft", ["f64", "f32"])
@pytest.mark.parametrize("dt,nsub", [(1e-2, 1), (20.0, 1), (60.0, 4)])
def test_mp1m_linearized_average(host, oracle, ft, dt, nsub):
    n = 40_000
    st = [c.astype(NPT[ft]) for c in _random_state(n, seed=21)]
    mp, tps = P.Microphysics1MParams(ft), P.ThermodynamicsParameters(ft)
    head = (C.byref(mp.c), C.byref(tps), C.c_uint32(mp.flags), CT[ft](Q_MIN), CT[ft](dt), C.c_int32(nsub))
    _, got = _call(host, "host_mp1m_linearized", ft, head, st, 4)
    c64 = [c.astype(np.float64) for c in st]
    mp64 = P.Microphysics1MParams("f64")
    ref = oracle.mp1m_linearized_average(_abi.F64, mp64.c, P.ThermodynamicsParameters("f64"), mp64.flags, Q_MIN, dt, nsub, *c64,
                                         float32_gates=(ft == "f32"), nthreads=8)
    inst = oracle.mp1m(_abi.F64, mp64.c, P.ThermodynamicsParameters("f64"), mp64.flags, *c64, float32_gates=(ft == "f32"), nthreads=8,
                       want_sources=False)
    check_linearized_parity(ft, dict(zip(NAMES, got)), ref, inst, c64, dt, f"host-build 1M LinearizedAverage {ft} dt={dt} nsub={nsub}")


def _chen_table(ft, variant):
    """default: Table B1;  low_b: an exponent below the round-2 polynomial window (b₃ + 1 = 1.3 — the host-fitted Γ covers it);
    steep: b_ρ = 0.6, Γ's argument moves by 1.2 over 0 ≤ ρ ≤ 2 → the fit misses its accuracy → general (run-time Γ) instantiation."""
    cr = P.Chen2022VelTypeRain(ft)
    if variant == "low_b":
        cr.b[2] = 0.3
    elif variant == "steep":
        cr.b_rho = 0.6
    return cr


@pytest.mark.parametrize("ft", ["f32", "f64"])
@pytest.mark.parametrize("general,variant", [(0, "default"), (1, "default"), (-1, "low_b"), (-1, "steep")])
def test_sedimentation_velocities(host, oracle, ft, general, variant):
    """The four fall speeds a host model precomputes; the Chen-2022 rain term with the host-fitted Γ polynomials (0), the run-time Γ (1),
    and (−1) whichever instantiation the entry points pick for a modified table."""
    n = 50_000
    rng = np.random.default_rng(3)
    rho = rng.uniform(0.3, 1.3, n)
    q = [np.where(rng.random(n) < 0.2, 0.0, 10 ** rng.uniform(-8, -2.5, n)) for _ in range(4)]
    mp = P.Microphysics1MParams(ft)
    stokes, cr, ci = P.StokesRegimeVelType(ft), _chen_table(ft, variant), P.Chen2022VelTypeIce(ft)
    head = (C.byref(mp.c), C.byref(stokes), C.byref(cr), C.byref(ci), C.c_int(general))
    picked_general, w = _call(host, "host_sedimentation", ft, head, [rho] + q, 4)
    assert picked_general == (1 if variant == "steep" else 0)
    mp64 = P.Microphysics1MParams("f64")
    x64 = [np.asarray(c, dtype=NPT[ft]).astype(np.float64) for c in [rho] + q]
    ref = oracle.sedimentation_velocities(_abi.F64, mp64.c, P.StokesRegimeVelType("f64"), _chen_table("f64", variant), P.Chen2022VelTypeIce("f64"),
                                          *x64, float32_gates=(ft == "f32"))
    # the Chen-2022 ice curves are differences of two terms that cancel near the zero crossing (E + F e^{−cD} with E ≈ −F): conditioning
    # scale = the positive term alone (oracle evaluated with the negative amplitude switched off), as in tests/test_mp1m_gpu.py
    pos = P.Chen2022VelTypeIce("f64")
    pos.small_ice.F[0] = -1e30
    pos.large_ice.E[0], pos.large_ice.E[1], pos.large_ice.E[2] = 0.0, 0.0, 0.0
    scale = oracle.sedimentation_velocities(_abi.F64, mp64.c, P.StokesRegimeVelType("f64"), P.Chen2022VelTypeRain("f64"), pos, *x64,
                                            float32_gates=(ft == "f32"))
    for k, got in zip(("w_lcl", "w_icl", "w_rai", "w_sno"), w):
        rr = ref[k]
        sc = scale[k] if k in ("w_icl", "w_sno") else np.abs(rr)
        tol = parity.RTOL[ft] * 0.1 * np.abs(rr) + parity.CTOL[ft] * sc
        assert np.all(np.abs(got.astype(np.float64) - rr) <= tol + 1e-300), (k, float(np.max(np.abs(got - rr) / (tol + 1e-300))))


# ---- round 5: the packed pair instantiations are the one-point arithmetic, bit for bit -----------------------------------------------------------
CLANG = Path("/opt/rocm/lib/llvm/bin/clang++")


@pytest.fixture(scope="module")
def host_clang():
    """The same file built by clang++, which has ext_vector_type: the f32x2 value type of csrc/cmx_math.hpp exists in this build."""
    if not CLANG.exists():
        pytest.skip("no clang++ in this image (the packed value type needs ext_vector_type)")
    out = REPO / "tests" / "native" / "_build"
    out.mkdir(exist_ok=True)
    so = out / "libpoint_host_clang.so"
    subprocess.run([str(CLANG), "-std=c++17", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-o", str(so), str(REPO / "tests" / "native" / "point_host.cpp")], check=True)
    lib = C.CDLL(str(so))
    assert lib.host_have_packed() == 1
    return lib


@pytest.mark.parametrize("optset", list(OPTION_SETS))
def test_packed_pairs_are_bit_identical_to_single_points(host_clang, optset):
    """mp1m_tendencies_point / mp1m_linearized_point / mp1m_sed_fluxes instantiated on the pair type f32x2 (what the Float32 kernels with four points per lane
    run since round 5) against the one-point instantiation of the SAME build, on random states plus NaN, zero and negative entries: identical bits —
    the generic value-type code (lane masks, m_or / m_and, nan_mask, the scalar-broadcast overloads of Math<f32x2>) changes no operation and no order."""
    ft, opts = "f32", OPTION_SETS[optset]
    n = 40_000
    st = [c.numpy().copy() for c in synthetic.mp1m_state(n, dtype=torch.float32, seed=77)]
    rng = np.random.default_rng(5)
    for c in st[2:]:
        k = rng.integers(0, n, 200)
        c[k[:70]] = 0.0; c[k[70:140]] = -1e-9; c[k[140:]] = np.nan
    st[1][rng.integers(0, n, 50)] = np.float32(T_FREEZE)              # exactly at the warm / cold routing
    mp, tps = P.Microphysics1MParams(ft, **opts), P.ThermodynamicsParameters(ft)
    head = (C.byref(mp.c), C.byref(tps), C.c_uint32(mp.flags))
    def same(a, b):
        """identical bits; ±0 are taken as equal HERE (x86 maxss / minss return their second operand for (+0, −0), and the compiler is free to commute the
        operands of fmaxf differently in the two instantiations — the device's v_max_f32 orders −0 < +0 and has no such freedom: the GPU suite compares bits)"""
        ua, ub = a.view(np.uint32), b.view(np.uint32)
        return bool(np.all((ua == ub) | ((a == 0) & (b == 0))))
    rc1, one = _call(host_clang, "host_mp1m_tendencies", ft, head, st, 4)
    rc2, two = _call(host_clang, "host_mp1m_tendencies_pairs", ft, head, st, 4)
    assert rc1 == rc2
    for a, b, name in zip(one, two, NAMES):
        assert same(a, b), (name, int(np.sum(a.view(np.uint32) != b.view(np.uint32))))
    lin = (*head, C.c_float(Q_MIN), C.c_float(30.0), C.c_int32(2))
    _, one = _call(host_clang, "host_mp1m_linearized", ft, lin, st, 4)
    _, two = _call(host_clang, "host_mp1m_linearized_pairs", ft, lin, st, 4)
    for a, b, name in zip(one, two, NAMES):
        assert same(a, b), ("linearized", name)
    if optset == "default":
        stokes, cr, ci = P.StokesRegimeVelType(ft), P.Chen2022VelTypeRain(ft), P.Chen2022VelTypeIce(ft)
        cols = [st[0], st[3], st[4], st[5], st[6]]
        hd = lambda pairs: (C.byref(mp.c), C.byref(stokes), C.byref(cr), C.byref(ci), C.c_int(pairs))  # noqa: E731
        rc, one = _call(host_clang, "host_sed_fluxes", ft, hd(0), cols, 4)
        assert rc == 0
        _, two = _call(host_clang, "host_sed_fluxes", ft, hd(1), cols, 4)
        for a, b in zip(one, two):
            assert same(a, b)


SB_NAMES = ["dq_lcl_dt", "dn_lcl_dt", "dq_rai_dt", "dn_rai_dt", "vt_rai_n", "vt_rai_m"]


@pytest.mark.parametrize("ft", ["f32", "f64"])
@pytest.mark.parametrize("limited", [True, False])
@pytest.mark.parametrize("vel", ["sb2006", "chen2022"])
def test_sb2006_point_function_on_the_host(host, oracle, ft, limited, vel):
    """The north-star point function (csrc/cmx_sb2006.hpp sb2006_point + the fused sums of the tendencies kernel) compiled for the host against the oracle:
    the log2-domain algebra, the host-folded constants (incl. round 5's t*·D_r) and the gates, without a GPU."""
    n = 60_000
    st = [c.numpy() for c in synthetic.sb2006_state(n, dtype=getattr(torch, {"f32": "float32", "f64": "float64"}[ft]), seed=321)]
    wr, tps, velp = P.WarmRainParams2M(ft, limited), P.ThermodynamicsParameters(ft), P.rain_vel_params(ft)
    flags = (_abi.CMX_SB2006_LIMITED if limited else 0) | (_abi.CMX_VEL_CHEN2022 if vel == "chen2022" else _abi.CMX_VEL_SB2006)
    rc, got = _call(host, "host_sb2006", ft, (C.byref(wr.c), C.byref(tps), C.byref(velp), C.c_uint32(flags), C.c_int(0)), st, 6)
    assert rc == 1                                   # the default parameter set takes the integer-exponent instantiation
    ref = oracle.sb2006_warm_rain_tendencies(_abi.F64, P.WarmRainParams2M("f64", limited).c, P.ThermodynamicsParameters("f64"), P.rain_vel_params("f64"), flags,
                                             *[np.asarray(c, dtype=np.float64) for c in st], float32_gates=(ft == "f32"), nthreads=8)
    parity.assert_parity(dict(zip(SB_NAMES, got)), ref, parity.RTOL[ft], names=SB_NAMES, what=f"host-build SB2006 {ft} {'limited' if limited else 'not limited'} {vel}")


@pytest.mark.parametrize("limited", [True, False])
@pytest.mark.parametrize("vel", ["sb2006", "chen2022"])
def test_sb2006_packed_pairs_are_bit_identical(host_clang, limited, vel):
    ft, n = "f32", 40_000
    st = [c.numpy().copy() for c in synthetic.sb2006_state(n, dtype=torch.float32, seed=99)]
    rng = np.random.default_rng(6)
    for c in (st[0], *st[2:]):
        k = rng.integers(0, n, 150)
        c[k[:50]] = 0.0; c[k[50:100]] = -1e-9; c[k[100:]] = np.nan
    wr, tps, velp = P.WarmRainParams2M(ft, limited), P.ThermodynamicsParameters(ft), P.rain_vel_params(ft)
    flags = (_abi.CMX_SB2006_LIMITED if limited else 0) | (_abi.CMX_VEL_CHEN2022 if vel == "chen2022" else _abi.CMX_VEL_SB2006)
    hd = lambda pairs: (C.byref(wr.c), C.byref(tps), C.byref(velp), C.c_uint32(flags), C.c_int(pairs))  # noqa: E731
    _, one = _call(host_clang, "host_sb2006", ft, hd(0), st, 6)
    _, two = _call(host_clang, "host_sb2006", ft, hd(1), st, 6)
    for a, b, name in zip(one, two, SB_NAMES):
        ua, ub = a.view(np.uint32), b.view(np.uint32)
        same = (ua == ub) | ((a == 0) & (b == 0)) | (np.isnan(a) & np.isnan(b))       # ±0: see test_packed_pairs_are_bit_identical_to_single_points
        assert bool(np.all(same)), (name, int(np.sum(~same)))



@pytest.mark.parametrize("ft", ["f32", "f64"])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_sb2006_point_function_with_perturbed_parameter_sets(host, oracle, ft, seed):
    """Every SB2006 / air / relaxation parameter scaled by a random factor in [0.8, 1.25] (limiters kept ordered; the three "integer" exponents become
    general ones → the exp2(e·log2 x) instantiation): the host-folded constants of make_sb_consts and the log2-domain algebra follow ANY parameter set, not
    just the ClimaParams defaults — host build against the oracle evaluated with the same perturbed structs."""
    rng = np.random.default_rng(seed)
    names = [k for k in P.DEFAULT_PARAMETERS if k.startswith("SB2006_") and "distribution_coeff_nu" not in k and "distribution_coeff_mu" not in k]
    names += ["condensation_evaporation_timescale", "thermal_conductivity_of_air", "diffusivity_of_water_vapor", "kinematic_viscosity_of_air"]
    names = [k for k in names if k in P.DEFAULT_PARAMETERS and isinstance(P.DEFAULT_PARAMETERS[k], float)]
    ov = {k: P.DEFAULT_PARAMETERS[k] * rng.uniform(0.8, 1.25) for k in names}
    for lo, hi in (("SB2006_raindrops_min_mass", "SB2006_raindrops_max_mass"), ("SB2006_raindrops_size_distribution_coeff_N0_min", "SB2006_raindrops_size_distribution_coeff_N0_max"),
                   ("SB2006_raindrops_size_distribution_coeff_lambda_min", "SB2006_raindrops_size_distribution_coeff_lambda_max"),
                   ("SB2006_raindrops_breakup_mean_diameter_threshold", "SB2006_raindrops_equilibrium_mean_diameter")):
        assert ov[lo] < ov[hi]
    n = 40_000
    st = [c.numpy() for c in synthetic.sb2006_state(n, dtype=getattr(torch, {"f32": "float32", "f64": "float64"}[ft]), seed=100 + seed)]
    for limited in (True, False):
        wr, wr64 = P.WarmRainParams2M(P.create_toml_dict(ft, ov), limited), P.WarmRainParams2M(P.create_toml_dict("f64", ov), limited)
        tps, velp = P.ThermodynamicsParameters(ft), P.rain_vel_params(ft)
        flags = (_abi.CMX_SB2006_LIMITED if limited else 0) | _abi.CMX_VEL_SB2006
        rc, got = _call(host, "host_sb2006", ft, (C.byref(wr.c), C.byref(tps), C.byref(velp), C.c_uint32(flags), C.c_int(0)), st, 6)
        assert rc == 0                               # perturbed exponents: the general-exponent instantiation
        ref = oracle.sb2006_warm_rain_tendencies(_abi.F64, wr64.c, P.ThermodynamicsParameters("f64"), P.rain_vel_params("f64"), flags,
                                                 *[np.asarray(c, dtype=np.float64) for c in st], float32_gates=(ft == "f32"), nthreads=8)
        parity.assert_parity(dict(zip(SB_NAMES, got)), ref, parity.RTOL[ft], names=SB_NAMES, what=f"host-build SB2006 perturbed parameters {ft} seed {seed} {'limited' if limited else 'not limited'}")
