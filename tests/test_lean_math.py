"""CPU test of csrc/cmx_lean_f64.hpp (the Float64 elementary functions the f64 kernels use instead of OCML): the
same header compiled for the host (tests/native/lean_math_host.cpp; the hardware reciprocal / rsqrt seeds are
replaced by single-precision stand-ins of the same accuracy class) against numpy's libm results."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def lean(tmp_path_factory):
    so = tmp_path_factory.mktemp("lean") / "liblean.so"
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-o", str(so),
                    str(REPO / "tests" / "native" / "lean_math_host.cpp")], check=True)
    lib = C.CDLL(str(so))

    def ev(which, x, pinned=False):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty_like(x)
        (lib.lean_eval_pinned if pinned else lib.lean_eval)(C.c_int(which), C.c_int64(x.size), x.ctypes.data_as(C.c_void_p),
                                                            y.ctypes.data_as(C.c_void_p))
        return y
    return ev


EXP2, LOG2, EXP, LOG, RCP, SQRT, RSQRT, EXPM1, LOG1P, ERFC, LGAMMA, EXP2_FIN, EXP_FIN, RCP_FINITE, RCP_NZ, SQRT_POS, RSQRT_POS, POW_M34, LOG_POS = range(19)


@pytest.mark.parametrize("script", ["gen_lean_tables.py", "gen_erfc_table.py"])
def test_generated_tables_are_current(script):
    """csrc/cmx_lean_tables.inc and csrc/cmx_erfc_table.inc are what their generators (mpmath) write."""
    import sys
    r = subprocess.run([sys.executable, str(REPO / "tools" / script), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def ulps(y, r):
    with np.errstate(all="ignore"):
        return np.nanmax(np.abs(y - r) / np.spacing(np.abs(r)))


def test_accuracy_in_ulps(lean):
    rng = np.random.default_rng(0)
    n = 400_000
    x = rng.uniform(-1000, 1000, n)
    assert ulps(lean(EXP2, x), np.exp2(x)) <= 2
    x = rng.uniform(-700, 700, n)
    assert ulps(lean(EXP, x), np.exp(x)) <= 2
    x = np.exp(rng.uniform(-700, 700, n))
    assert ulps(lean(LOG2, x), np.log2(x)) <= 4 and ulps(lean(LOG, x), np.log(x)) <= 4
    x = 1 + rng.uniform(-0.3, 0.4, n)            # relative accuracy where the result is small
    assert np.max(np.abs(lean(LOG2, x) - np.log2(x)) / np.abs(np.log2(x))) < 1e-15
    x = np.exp(rng.uniform(-80, 80, n))
    assert ulps(lean(RCP, x), 1 / x) <= 1 and ulps(lean(SQRT, x), np.sqrt(x)) <= 1 and ulps(lean(RSQRT, x), 1 / np.sqrt(x)) <= 3
    for lo, hi in ((-1e-5, 1e-5), (-0.5, 0.5), (-40, 40), (-3, 700)):
        x = rng.uniform(lo, hi, n)
        assert ulps(lean(EXPM1, x), np.expm1(x)) <= 3
    for lo, hi in ((-1e-5, 1e-5), (-0.9, 0.9), (0, 1e6)):
        x = rng.uniform(lo, hi, n)
        assert ulps(lean(LOG1P, x), np.log1p(x)) <= 5


def test_erfc(lean):
    """The table-driven erfc of the Float64 ARG kernel against mpmath (scipy's erfc is itself only good to ≈ 3e-15 relative):
    relative error ≤ (2 + x²)·2e-16 on [0, 6.5] — one rounding of x² enters the exponent — absolute ≤ 4.5e-16 for negative arguments
    (2 − erfc|x|), the tail beyond 6.5 within a factor x/6.5 of a value below 4e-20, IEEE special values."""
    import mpmath as mp
    mp.mp.dps = 40
    rng = np.random.default_rng(11)
    x = np.concatenate([rng.uniform(0, 6.5, 4000), np.linspace(0, 6.5, 64 * 16 + 1), 6.5 / 64 * np.arange(65)])   # incl. the interval joints
    y = lean(ERFC, x)
    worst = max(float(abs((mp.mpf(float(yi)) - mp.erfc(mp.mpf(float(xi)))) / mp.erfc(mp.mpf(float(xi)))) / (2 + xi * xi)) for xi, yi in zip(x, y))
    assert worst < 2e-16, worst
    from scipy.special import erfc
    x = rng.uniform(0, 6.5, 400_000)                      # the bulk: against scipy at scipy's own accuracy
    assert np.max(np.abs(lean(ERFC, x) - erfc(x)) / erfc(x)) < 6e-15
    x = -rng.uniform(0, 8, 100_000)
    assert np.max(np.abs(lean(ERFC, x) - erfc(x))) < 4.5e-16
    x = rng.uniform(6.5, 26, 10_000)
    y, r = lean(ERFC, x), erfc(x)
    assert np.all(y >= r * (1 - 1e-13)) and np.all(y <= r * (x / 6.5) * (1 + 1e-13)) and np.all(y < 4.1e-20)
    with np.errstate(all="ignore"):
        np.testing.assert_array_equal(lean(ERFC, [np.inf, -np.inf, np.nan, 40.0, -40.0]), [0.0, 2.0, np.nan, 0.0, 2.0])
        assert np.all(np.abs(lean(ERFC, [0.0, -0.0]) - 1.0) <= 2.3e-16)        # the end of the first interval: within the polynomial's 1 ulp


def test_lgamma_pos(lean):
    """ln Γ(z), z > 0 (Stirling at z + 7): absolute error ≤ 1.5e-14 on (0, 12] — the P3 shape solver's range — and relative ≤ 1e-15
    beyond 12, against mpmath."""
    import mpmath as mp
    mp.mp.dps = 40
    rng = np.random.default_rng(13)
    x = np.concatenate([rng.uniform(1e-6, 12, 4000), [1.0, 2.0, 0.5, 7.999999, 8.0, 8.000001]])
    y = lean(LGAMMA, x)
    assert max(abs(float(mp.mpf(float(yi)) - mp.loggamma(mp.mpf(float(xi))))) for xi, yi in zip(x, y)) < 1.5e-14
    x = np.exp(rng.uniform(np.log(12), np.log(1e12), 2000))
    y = lean(LGAMMA, x)
    assert max(abs(float((mp.mpf(float(yi)) - mp.loggamma(mp.mpf(float(xi)))) / mp.loggamma(mp.mpf(float(xi))))) for xi, yi in zip(x, y)) < 1e-15


def test_pinned_table_forms(lean):
    """exp(x, TabCoefs) / log(x, TabCoefs): the forms the P3 quadrature loops call."""
    rng = np.random.default_rng(3)
    x = rng.uniform(-700, 700, 400_000)
    assert ulps(lean(EXP, x, pinned=True), np.exp(x)) <= 2
    x = np.exp(rng.uniform(-700, 700, 400_000))
    assert ulps(lean(LOG, x, pinned=True), np.log(x)) <= 3
    x = 1 + rng.uniform(-1e-3, 1e-3, 100_000)
    assert ulps(lean(LOG, x, pinned=True), np.log(x)) <= 3
    with np.errstate(all="ignore"):
        np.testing.assert_array_equal(lean(EXP, [0.0, np.inf, -np.inf, np.nan, 800.0, -800.0], pinned=True), [1.0, np.inf, 0.0, np.nan, np.inf, 0.0])
        np.testing.assert_array_equal(lean(LOG, [0.0, np.inf, np.nan, -1.0, 1.0, 5e-324], pinned=True)[:5], [-np.inf, np.inf, np.nan, np.nan, 0.0])


def test_special_values(lean):
    inf, nan = np.inf, np.nan
    with np.errstate(all="ignore"):
        np.testing.assert_array_equal(lean(EXP2, [0.0, inf, -inf, nan, 2000.0, -2000.0, -1074.0]), [1.0, inf, 0.0, nan, inf, 0.0, 5e-324])
        np.testing.assert_array_equal(lean(EXP, [0.0, inf, -inf, nan, 800.0, -800.0]), [1.0, inf, 0.0, nan, inf, 0.0])
        np.testing.assert_array_equal(lean(LOG2, [0.0, -0.0, inf, nan, -1.0, 1.0, 5e-324, 2.0 ** 1000]), [-inf, -inf, inf, nan, nan, 0.0, -1074.0, 1000.0])
        np.testing.assert_array_equal(lean(LOG, [0.0, inf, nan, -1.0, 1.0]), [-inf, inf, nan, nan, 0.0])
        np.testing.assert_array_equal(lean(RCP, [0.0, -0.0, inf, -inf, nan, 4.0]), [inf, -inf, 0.0, -0.0, nan, 0.25])
        np.testing.assert_array_equal(lean(SQRT, [0.0, inf, nan, 4.0]), [0.0, inf, nan, 2.0])
        np.testing.assert_array_equal(lean(RSQRT, [0.0, inf, 4.0]), [inf, 0.0, 0.5])
        np.testing.assert_array_equal(lean(EXPM1, [0.0, inf, -inf, nan, -800.0]), [0.0, inf, -1.0, nan, -1.0])
        np.testing.assert_array_equal(lean(LOG1P, [0.0, inf, nan, -1.0, -2.0]), [0.0, inf, nan, -inf, nan])
        assert np.isnan(lean(SQRT, [-1.0])[0])


# ---- the same functions ON THE DEVICE (cmx_lean_eval_f64: LDS tables, hardware rcp / rsq seeds, v_ldexp / v_frexp) -------------------
# both builds of the polynomial coefficients: LDS reads (cmx_common.hip) and SGPR literals (the production Float64 kernels' units)
@pytest.fixture(scope="module", params=["cmx_lean_eval_f64", "cmx_lean_eval_literal_f64"])
def dev_lean(request):
    import torch

    from cmx import _lib
    assert torch.cuda.is_available()
    lib = _lib.lib()
    fn = getattr(lib, request.param)

    def ev(which, x):
        xd = torch.as_tensor(np.ascontiguousarray(x, dtype=np.float64)).cuda()
        yd = torch.empty_like(xd)
        st = fn(which, xd.numel(), C.c_void_p(xd.data_ptr()), C.c_void_p(yd.data_ptr()), None)
        assert st == 0
        torch.cuda.synchronize()
        return yd.cpu().numpy()
    return ev


@pytest.mark.gpu
def test_device_accuracy_in_ulps(dev_lean):
    test_accuracy_in_ulps.__wrapped__(dev_lean) if hasattr(test_accuracy_in_ulps, "__wrapped__") else test_accuracy_in_ulps(dev_lean)
    rng = np.random.default_rng(5)
    n = 2_000_000
    x = rng.uniform(-1000, 1000, n)
    assert ulps(dev_lean(EXP2, x), np.exp2(x)) <= 2          # round 4: the degree-4 Chebyshev form (1.5e-16 + roundings); ≤ 1 with round 2's degree 5
    x = np.exp(rng.uniform(-700, 700, n))
    assert ulps(dev_lean(LOG2, x), np.log2(x)) <= 2
    x = 1 + rng.uniform(-1e-3, 1e-3, n)                # the table interval around 1 has c = 1: relative accuracy is kept
    assert ulps(dev_lean(LOG2, x), np.log2(x)) <= 2 and ulps(dev_lean(LOG, x), np.log(x)) <= 3
    x = 10.0 ** rng.uniform(-320, -300, 100_000)       # subnormal arguments take the rescue path
    assert ulps(dev_lean(LOG2, x), np.log2(x)) <= 2


def test_finite_argument_forms(lean):
    """exp2_fin / exp_fin / rcp_finite (Math<double>::exp2_fin, rcp_nz): the full forms minus the clamp, the NaN select and the 0 / Inf
    fix-up.  Same accuracy; finite arguments of ANY size still give the right overflow / underflow result; NaN propagates.  (±Inf into
    the exponentials and 0 / ±Inf into the reciprocal are outside their contract: the call sites exclude them.)"""
    rng = np.random.default_rng(21)
    x = rng.uniform(-1000, 1000, 400_000)
    assert ulps(lean(EXP2_FIN, x), np.exp2(x)) <= 2
    np.testing.assert_array_equal(lean(EXP2_FIN, x), lean(EXP2, x))
    x = rng.uniform(-700, 700, 400_000)
    np.testing.assert_array_equal(lean(EXP_FIN, x), lean(EXP, x))
    x = np.concatenate([10.0 ** rng.uniform(-300, 300, 200_000), -10.0 ** rng.uniform(-300, 300, 200_000)])
    np.testing.assert_array_equal(lean(RCP_FINITE, x), lean(RCP, x))
    worst = ulps(lean(RCP_NZ, x), 1.0 / x)                      # seed + one Newton step: 2⁻⁴⁸ relative
    print(f"rcp_nz worst error {worst:.1f} ulp")
    assert worst <= 40
    x = 10.0 ** rng.uniform(-300, 300, 400_000)
    np.testing.assert_array_equal(lean(SQRT_POS, x), lean(SQRT, x))
    np.testing.assert_array_equal(lean(RSQRT_POS, x), lean(RSQRT, x))
    x = 10.0 ** rng.uniform(-100, 100, 400_000)                 # x^(−¾): the ARG S_max sum (3ζ + η: 1e-12 … 1e3)
    assert ulps(lean(POW_M34, x), x ** -0.75) <= 6            # u³: three times the error of u = x^(−¼) plus two roundings
    assert np.isnan(lean(POW_M34, [np.nan])[0]) and np.isnan(lean(SQRT_POS, [np.nan])[0]) and np.isnan(lean(RSQRT_POS, [np.nan])[0])
    with np.errstate(all="ignore"):
        big = [1100.0, 2000.0, 1e6, 1e12, 1e300, -1100.0, -2000.0, -1e6, -1e12, -1e300, -1074.0, -1060.5]
        np.testing.assert_array_equal(lean(EXP2_FIN, big), np.exp2(big))
        eb = [800.0, 1e6, 1e12, -800.0, -1e6, -1e12]            # e^x: the Cody–Waite reduction needs |x|·128/ln2 to be an exact integer (|x| < 2e13)
        np.testing.assert_array_equal(lean(EXP_FIN, eb), np.exp(eb))
        assert np.isnan(lean(EXP2_FIN, [np.nan])[0]) and np.isnan(lean(EXP_FIN, [np.nan])[0]) and np.isnan(lean(RCP_NZ, [np.nan])[0]) and np.isnan(lean(RCP_FINITE, [np.nan])[0])


def test_log_pos(lean):
    """log_pos (round 4): ln x for a positive NORMAL finite x — the main path of log() without the class test and the rescue block; the P3
    quadrature integrands call it on diameters and areas at interior nodes.  Bit-identical to log() on its domain, in the literal-coefficient
    form and in the pinned (TabCoefs) form.  NaN is OUTSIDE its contract (it comes out as a finite number): the integrands also feed x linearly
    into every result, which is what carries a NaN node (tests/test_nan_inputs_gpu.py)."""
    rng = np.random.default_rng(31)
    x = np.concatenate([10.0 ** rng.uniform(-307, 308, 400_000), 1 + rng.uniform(-1e-3, 1e-3, 100_000), [2.2250738585072014e-308, 1.7976931348623157e308, 1.0]])
    np.testing.assert_array_equal(lean(LOG_POS, x), lean(LOG, x))
    np.testing.assert_array_equal(lean(LOG_POS, x, pinned=True), lean(LOG, x, pinned=True))
    assert ulps(lean(LOG_POS, x), np.log(x)) <= 3
    assert np.isfinite(lean(LOG_POS, [np.nan])[0])          # documented: not a NaN-propagating form


@pytest.mark.gpu
def test_device_log_pos(dev_lean):
    rng = np.random.default_rng(32)
    x = np.concatenate([10.0 ** rng.uniform(-307, 308, 400_000), 1 + rng.uniform(-1e-3, 1e-3, 100_000), [2.2250738585072014e-308, 1.7976931348623157e308, 1.0]])
    np.testing.assert_array_equal(dev_lean(LOG_POS, x), dev_lean(LOG, x))
    assert ulps(dev_lean(LOG_POS, x), np.log(x)) <= 3


@pytest.mark.gpu
def test_device_finite_argument_forms(dev_lean):
    test_finite_argument_forms(dev_lean)


@pytest.mark.gpu
def test_device_special_values(dev_lean):
    test_special_values(dev_lean)


@pytest.mark.gpu
def test_device_erfc(dev_lean):
    test_erfc(dev_lean)


@pytest.mark.gpu
def test_device_lgamma_pos(dev_lean):
    test_lgamma_pos(dev_lean)
