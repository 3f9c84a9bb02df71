"""Parity metric shared by the CPU and GPU tests.

north_star tolerance: ≤ 1e-6 (Float64 kernels), ≤ 1e-3 (Float32 kernels) relative to the reference's Float64 CPU
arithmetic on identical inputs.  Several outputs are differences of large terms (S = p_v/p_sat − 1, q_v − q_sat,
T − T_freeze, Σ number tendencies of both signs, aR − bR/(1+cR·D)), for which a pointwise relative error is ill-posed
near the zero crossing: there the achievable accuracy is a few hundred ulps of the OPERANDS, not of
the result.  The bound checked is therefore

    |x − ref|  ≤  RTOL · |ref|  +  CTOL · scale            scale = Σ|cancelling operand terms| (oracle, Float64)

i.e. the north-star relative tolerance on the value itself wherever it is well-conditioned, plus a tight operand-
relative allowance (CTOL = 2e-5 for Float32 ≈ 170 ulp, 1e-12 for Float64) that only matters where the result is a
small difference of large terms.  Reported as the normalised error  |x − ref| / (|ref| + (CTOL/RTOL)·scale) ≤ RTOL.

Points within a stated margin of a genuine DISCONTINUITY of the scheme (Φ_br at Dr = Dr_th, CM2:596; the warm/cold
routing at T = T_freeze, BMT:171) may legitimately land on either branch in another precision: they are excluded,
counted, and must stay a vanishing fraction.
"""
import numpy as np

RTOL = {"f32": 1e-3, "f64": 1e-6}
CTOL = {"f32": 2e-5, "f64": 1e-12}
# magnitudes below this are "zero" for the kernel's float type (≈ floatmin(FT) with headroom for one product):
# the hardware transcendental units flush subnormals, the reference's CPU arithmetic keeps them.
FLOOR = {"f32": 1e-30, "f64": 1e-290}
# …and magnitudes above this overflow the kernel's float type (exp(κbr ΔD) of a 20-cm "mean raindrop" in the
# not-limited PSD is 1e197 in Float64 and Inf in Float32, in the reference's Float32 path too): an infinity of
# the right sign is then the correct Float32 answer.
CEIL = {"f32": 1e30, "f64": 1e300}
OUT_NAMES = ["dq_lcl_dt", "dn_lcl_dt", "dq_rai_dt", "dn_rai_dt", "vt_rai_n", "vt_rai_m"]


def _ft_of(rtol):
    return "f32" if rtol >= 1e-4 else "f64"


def scaled_err(x, ref, scale=None, floor=0.0, ceil=np.inf, kappa=None):
    """Normalised error |x − ref| / (max(|ref|, floor) + kappa·scale); kappa = CTOL/RTOL (default: the Float32 pair)."""
    x = np.asarray(x, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    if kappa is None:
        kappa = CTOL["f32"] / RTOL["f32"]
    den = np.maximum(np.abs(ref), floor)
    if scale is not None:
        den = den + kappa * np.asarray(scale, dtype=np.float64)
    both_zero = (x == 0) & (ref == 0)
    with np.errstate(divide="ignore", invalid="ignore"):
        e = np.abs(x - ref) / den
    e = np.where(both_zero, 0.0, e)
    # identical non-finite values (inf == inf, nan ↔ nan) count as equal
    same_nonfinite = (~np.isfinite(x)) & (~np.isfinite(ref)) & ((x == ref) | (np.isnan(x) & np.isnan(ref)))
    overflow_ok = (np.abs(ref) > ceil) & np.isinf(x) & (np.sign(x) == np.sign(ref))
    return np.where(same_nonfinite | overflow_ok, 0.0, e)


def assert_parity(got: dict, ref: dict, rtol: float, names=OUT_NAMES, what="", floor=None):
    """got/ref: name → array; ref carries 'scale' (name → array) and 'near_branch' (bool mask).  Returns the worst
    normalised error per output (must be ≤ rtol)."""
    ft = _ft_of(rtol)
    near = ref.get("near_branch")
    keep = ~near if near is not None else slice(None)
    report = {}
    if floor is None:
        floor = FLOOR[ft]
    for k in names:
        if got.get(k) is None:
            continue
        sc = ref.get("scale", {}).get(k)
        e = scaled_err(got[k], ref[k], sc, floor, CEIL[ft], CTOL[ft] / RTOL[ft])
        e = np.nan_to_num(e, nan=np.inf)
        worst = float(np.max(e[keep])) if e[keep].size else 0.0
        report[k] = worst
        if not worst <= rtol:
            i = int(np.argmax(np.where(near, 0, e) if near is not None else e))
            raise AssertionError(
                f"{what} {k}: normalised error {worst:.3e} > {rtol:g} at i={i}: got {np.asarray(got[k])[i]!r} "
                f"ref {ref[k][i]!r} scale {(sc[i] if sc is not None else None)!r}")
    if near is not None and near.size:
        assert near.mean() < 1e-4, f"{what}: implausibly many near-branch points ({near.sum()})"
    return report
