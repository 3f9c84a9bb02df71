"""Parity metric shared by the CPU and GPU tests.

north_star tolerance: ≤ 1e-6 (Float64 kernels), ≤ 1e-3 (Float32 kernels) relative to the
reference's Float64 CPU arithmetic on identical inputs.  Several outputs are sums of terms of
opposite sign (S = p_v/p_sat − 1, q_v − q_sat, Σ number tendencies, aR − bR/(1+cR·D)), for which a
pointwise relative error is ill-posed near the zero crossing.  The error is
therefore measured against max(|ref|, scale) where `scale` = Σ|cancelling terms| of that output,
computed by the oracle in Float64:

    err = |x − ref| / max(|ref|, scale)

and points within 1e-5 (relative) of the one genuine discontinuity of the scheme — the breakup
function Φ_br at Dr = Dr_th (CM2:596) — are compared against both branches' neighbourhood
separately (they are counted and must stay a vanishing fraction).
"""
import numpy as np

RTOL = {"f32": 1e-3, "f64": 1e-6}
# magnitudes below this are "zero" for the kernel's float type (≈ floatmin(FT) with headroom for one product):
# the hardware transcendental units flush subnormals, the reference's CPU arithmetic keeps them.
FLOOR = {"f32": 1e-30, "f64": 1e-290}
# …and magnitudes above this overflow the kernel's float type (exp(κbr ΔD) of a 20-cm "mean raindrop" in the
# not-limited PSD is 1e197 in Float64 and Inf in Float32, in the reference's Float32 path too): an infinity of
# the right sign is then the correct Float32 answer.
CEIL = {"f32": 1e30, "f64": 1e300}
OUT_NAMES = ["dq_lcl_dt", "dn_lcl_dt", "dq_rai_dt", "dn_rai_dt", "vt_rai_n", "vt_rai_m"]


def scaled_err(x, ref, scale=None, floor=0.0, ceil=np.inf):
    x = np.asarray(x, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    den = np.maximum(np.abs(ref), floor)
    if scale is not None:
        den = np.maximum(den, np.asarray(scale, dtype=np.float64))
    both_zero = (x == 0) & (ref == 0)
    with np.errstate(divide="ignore", invalid="ignore"):
        e = np.abs(x - ref) / den
    e = np.where(both_zero, 0.0, e)
    # identical non-finite values (inf == inf, nan ↔ nan) count as equal
    same_nonfinite = (~np.isfinite(x)) & (~np.isfinite(ref)) & ((x == ref) | (np.isnan(x) & np.isnan(ref)))
    overflow_ok = (np.abs(ref) > ceil) & np.isinf(x) & (np.sign(x) == np.sign(ref))
    return np.where(same_nonfinite | overflow_ok, 0.0, e)


def assert_parity(got: dict, ref: dict, rtol: float, names=OUT_NAMES, what="", floor=None):
    """got/ref: name → array; ref carries 'scale' (name → array) and 'near_branch' (bool mask)."""
    near = ref.get("near_branch")
    keep = ~near if near is not None else slice(None)
    report = {}
    if floor is None:
        floor = FLOOR["f32"] if rtol >= 1e-4 else FLOOR["f64"]
    ceil = CEIL["f32"] if rtol >= 1e-4 else CEIL["f64"]
    for k in names:
        if got.get(k) is None:
            continue
        e = scaled_err(got[k], ref[k], ref.get("scale", {}).get(k), floor, ceil)
        e = np.nan_to_num(e, nan=np.inf)
        worst = float(np.max(e[keep])) if e[keep].size else 0.0
        report[k] = worst
        if not worst <= rtol:
            i = int(np.argmax(np.where(near, 0, e) if near is not None else e))
            raise AssertionError(
                f"{what} {k}: scaled error {worst:.3e} > {rtol:g} at i={i}: got {np.asarray(got[k])[i]!r} "
                f"ref {ref[k][i]!r} scale {ref.get('scale', {}).get(k, [None] * (i + 1))[i]!r}")
    if near is not None and near.size:
        assert near.mean() < 1e-4, f"{what}: implausibly many near-branch points ({near.sum()})"
    return report
