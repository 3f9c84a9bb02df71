// cmx_lean_f64.hpp — lean Float64 elementary functions for gfx950.
//
// gfx950 has no Float64 transcendental unit; the OCML routines the compiler links for exp2/log2/exp/log cost
// 50–110 VALU instructions each (correctly rounded-ish, every special case, double-double range reduction), which
// makes every Float64 kernel of this library VALU-issue-bound (one f64 instruction = 4 cycles per wave).  The
// functions here spend 20–35 instructions for ≤ 4 ulp — three to four decimal orders below the parity tolerance of
// the Float64 kernels (1e-6) and well inside their measured 1e-13 — and keep the IEEE special values the rate
// functions rely on (log2(0) = −Inf, log2(x<0) = NaN, exp2(±Inf), NaN propagation, gradual underflow via ldexp).
//
// The same source compiles for the host (plain C++) so tests/test_lean_math.py can check it against libm.
#pragma once
#include <cmath>
#include <limits>

#if defined(__HIPCC__)
#define CMX_LEAN_FN __host__ __device__ __forceinline__   // hipcc: both passes see the same overload set
#else
#define CMX_LEAN_FN inline                                // plain host build (tests/native/lean_math_host.cpp)
#endif

namespace cmx {
namespace lean {

CMX_LEAN_FN double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }

// 1/d for finite, non-zero d: hardware seed (≈2⁻²⁴ relative on gfx950) + two Newton steps → ≤ 1 ulp
CMX_LEAN_FN double rcp_finite(double d) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(d);
#else
    double r = (double)(1.0f / (float)d);   // host stand-in for the seed (tests only; same precision class)
    if (!(std::fabs(d) > 1e-37 && std::fabs(d) < 1e37)) r = 1.0 / d;
#endif
    r = fma_(r, fma_(-d, r, 1.0), r);
    return fma_(r, fma_(-d, r, 1.0), r);
}
// 1/d with the IEEE results for d = ±0 (±Inf), ±Inf (±0) and NaN: whenever the seed is 0 or ±Inf the Newton steps
// would produce NaN, so the seed itself is returned (one class test on the seed; subnormal d saturates to ±Inf)
CMX_LEAN_FN double rcp(double d) {
#if defined(__HIP_DEVICE_COMPILE__)
    const double r0 = __builtin_amdgcn_rcp(d);
    double r = fma_(r0, fma_(-d, r0, 1.0), r0);
    r = fma_(r, fma_(-d, r, 1.0), r);
    return __builtin_amdgcn_class(r0, 0x204 | 0x060) ? r0 : r;              // ±Inf (0x204), ±0 (0x060)
#else
    const double inf = std::numeric_limits<double>::infinity();
    const double ad = __builtin_fabs(d);
    const double r = rcp_finite(d);
    const double s = d != d ? d : __builtin_copysign(ad == inf ? 0.0 : inf, d);
    return (ad >= 2.2250738585072014e-308 && ad < inf) ? r : s;
#endif
}
CMX_LEAN_FN double frexp_mant(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_frexp_mant(x);
#else
    int e; return std::frexp(x, &e);
#endif
}
CMX_LEAN_FN int frexp_exp(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_frexp_exp(x);
#else
    int e; std::frexp(x, &e); return e;
#endif
}
CMX_LEAN_FN double ldexp_(double x, int e) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ldexp(x, e);
#else
    return std::ldexp(x, e);
#endif
}

// eᵗ for |t| ≤ 0.3466 (= ln2 / 2): Taylor to t¹³ (truncation 4e-18), Horner
CMX_LEAN_FN double exp_poly(double t) {
    double p = 1.0 / 6227020800.0;
    p = fma_(p, t, 1.0 / 479001600.0);
    p = fma_(p, t, 1.0 / 39916800.0);
    p = fma_(p, t, 1.0 / 3628800.0);
    p = fma_(p, t, 1.0 / 362880.0);
    p = fma_(p, t, 1.0 / 40320.0);
    p = fma_(p, t, 1.0 / 5040.0);
    p = fma_(p, t, 1.0 / 720.0);
    p = fma_(p, t, 1.0 / 120.0);
    p = fma_(p, t, 1.0 / 24.0);
    p = fma_(p, t, 1.0 / 6.0);
    p = fma_(p, t, 0.5);
    p = fma_(p, t, 1.0);
    return fma_(p, t, 1.0);
}
// 2ˣ
CMX_LEAN_FN double exp2(double x) {
    const double xc = __builtin_fmin(__builtin_fmax(x, -1100.0), 1100.0);   // keeps the int conversion in range
    const double n = __builtin_rint(xc);
    const double t = (xc - n) * 0.6931471805599453;                           // xc − n exact, |t| ≤ ln2/2
    const double r = ldexp_(exp_poly(t), (int)n);                             // overflow → +Inf, underflow → denormal/0
    return x != x ? x : r;
}
// eˣ (Cody–Waite reduction with a two-part ln 2)
CMX_LEAN_FN double exp(double x) {
    const double xc = __builtin_fmin(__builtin_fmax(x, -760.0), 760.0);
    const double n = __builtin_rint(xc * 1.4426950408889634);
    double t = fma_(-n, 0.6931471803691238, xc);                              // ln2_hi: 33 significant bits
    t = fma_(-n, 1.9082149292705877e-10, t);                                  // ln2_lo
    const double r = ldexp_(exp_poly(t), (int)n);
    return x != x ? x : r;
}
// ln m for m ∈ [√½, √2):  2 atanh(s), s = (m−1)/(m+1), |s| ≤ 0.1716, series to s²¹ (truncation 2e-17 relative)
CMX_LEAN_FN double log_mant(double m) {
    const double s = (m - 1.0) * rcp_finite(m + 1.0);
    const double z = s * s;
    double p = 1.0 / 21.0;
    p = fma_(p, z, 1.0 / 19.0);
    p = fma_(p, z, 1.0 / 17.0);
    p = fma_(p, z, 1.0 / 15.0);
    p = fma_(p, z, 1.0 / 13.0);
    p = fma_(p, z, 1.0 / 11.0);
    p = fma_(p, z, 1.0 / 9.0);
    p = fma_(p, z, 1.0 / 7.0);
    p = fma_(p, z, 1.0 / 5.0);
    p = fma_(p, z, 1.0 / 3.0);
    // 2s(1 + z p) with the leading term kept exact
    return fma_(2.0 * s * z, p, 2.0 * s);
}
struct MantExp { double m; double e; };
CMX_LEAN_FN MantExp split(double x) {                                        // x = m·2ᵉ, m ∈ [√½, √2)
    double m = frexp_mant(x);
    int e = frexp_exp(x);
    const bool lo = m < 0.7071067811865476;
    m = lo ? m + m : m;
    e = lo ? e - 1 : e;
    return {m, (double)e};
}
CMX_LEAN_FN double log_special(double x, double r) {                          // IEEE special values of log-type results
    const double inf = std::numeric_limits<double>::infinity();
#if defined(__HIP_DEVICE_COMPILE__)
    // one class test; the fix-ups sit behind a branch that no lane takes for ordinary (positive, finite) arguments
    if (__builtin_amdgcn_class(x, 0x200 | 0x060 | 0x01C)) {                    // +Inf | ±0 | −Inf, −normal, −subnormal
        r = x == inf ? inf : r;
        r = x == 0.0 ? -inf : r;
        r = x < 0.0 ? std::numeric_limits<double>::quiet_NaN() : r;
    }
    return r;
#else
    r = x == inf ? inf : r;
    r = x == 0.0 ? -inf : r;
    return x < 0.0 ? std::numeric_limits<double>::quiet_NaN() : r;            // NaN input propagates through r
#endif
}
CMX_LEAN_FN double log2(double x) {
    const MantExp s = split(x);
    return log_special(x, fma_(log_mant(s.m), 1.4426950408889634, s.e));
}
CMX_LEAN_FN double log(double x) {
    const MantExp s = split(x);
    return log_special(x, fma_(s.e, 0.6931471803691238, fma_(s.e, 1.9082149292705877e-10, log_mant(s.m))));
}
// eˣ − 1: with x = n ln2 + t,  eˣ − 1 = 2ⁿ·(eᵗ − 1) + (2ⁿ − 1); eᵗ − 1 = t·q(t) has no cancellation
CMX_LEAN_FN double expm1(double x) {
    const double xc = __builtin_fmin(__builtin_fmax(x, -50.0), 760.0);        // e⁻⁵⁰ − 1 = −1 to the last bit
    const double n = __builtin_rint(xc * 1.4426950408889634);
    double t = fma_(-n, 0.6931471803691238, xc);
    t = fma_(-n, 1.9082149292705877e-10, t);
    double q = 1.0 / 6227020800.0;
    q = fma_(q, t, 1.0 / 479001600.0);
    q = fma_(q, t, 1.0 / 39916800.0);
    q = fma_(q, t, 1.0 / 3628800.0);
    q = fma_(q, t, 1.0 / 362880.0);
    q = fma_(q, t, 1.0 / 40320.0);
    q = fma_(q, t, 1.0 / 5040.0);
    q = fma_(q, t, 1.0 / 720.0);
    q = fma_(q, t, 1.0 / 120.0);
    q = fma_(q, t, 1.0 / 24.0);
    q = fma_(q, t, 1.0 / 6.0);
    q = fma_(q, t, 0.5);
    q = fma_(q, t, 1.0);
    const double em1 = q * t;                                                 // eᵗ − 1
    const double two_n = ldexp_(1.0, (int)n);
    const double r = fma_(two_n, em1, two_n - 1.0);
    return x != x ? x : r;
}
// ln(1 + x): ln(u)·x/(u − 1) with u = fl(1 + x) (Kahan) — exact compensation of the rounding in u
CMX_LEAN_FN double log1p(double x) {
    const double u = 1.0 + x;
    const double d = u - 1.0;
    const double r = log(u) * (x * rcp_finite(d == 0.0 ? 1.0 : d));
    const double inf = std::numeric_limits<double>::infinity();
    return d == 0.0 ? x : (x == inf ? inf : r);                                // x = −1 → −Inf, x < −1 → NaN via log
}
// ---- register-pinned constants -----------------------------------------------------------------------------------------
// A Float64 constant that is not one of the few inline values occupies an SGPR pair.  A loop that calls exp and log a few
// times needs ≈35 such constants on top of the kernel's parameters: beyond the SGPR file the compiler parks them in VGPR
// lanes and re-reads each with v_readlane_b32 — VALU instructions (measured: 96 of 264 per quadrature node in the P3
// fall-speed kernel).  `PinnedCoefs` keeps the polynomial / reduction constants in VGPRs instead (an empty asm makes each
// value opaque, so it cannot be rematerialised as a scalar constant): 62 VGPRs, no spill traffic.
struct PinnedCoefs {
    double e[13];           // 1/13!, 1/12!, …, 1/2!, 1/1! of exp_poly
    double log2e, ln2_hi, ln2_lo, exp_lo, exp_hi;
    double l[10];           // 1/21, 1/19, …, 1/3 of log_mant
    double sqrt_half;
};
CMX_LEAN_FN void pin(double &x) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(x));
#else
    (void)x;
#endif
}
CMX_LEAN_FN PinnedCoefs pinned_coefs() {
    PinnedCoefs k;
    const double f[13] = {1.0 / 6227020800.0, 1.0 / 479001600.0, 1.0 / 39916800.0, 1.0 / 3628800.0, 1.0 / 362880.0, 1.0 / 40320.0,
                          1.0 / 5040.0, 1.0 / 720.0, 1.0 / 120.0, 1.0 / 24.0, 1.0 / 6.0, 0.5, 1.0};
    for (int i = 0; i < 13; ++i) { k.e[i] = f[i]; pin(k.e[i]); }
    k.log2e = 1.4426950408889634; k.ln2_hi = 0.6931471803691238; k.ln2_lo = 1.9082149292705877e-10; k.exp_lo = -760.0; k.exp_hi = 760.0;
    pin(k.log2e); pin(k.ln2_hi); pin(k.ln2_lo); pin(k.exp_lo); pin(k.exp_hi);
    for (int i = 0; i < 10; ++i) { k.l[i] = 1.0 / (21.0 - 2.0 * i); pin(k.l[i]); }
    k.sqrt_half = 0.7071067811865476; pin(k.sqrt_half);
    return k;
}
// eˣ and ln x with the constants taken from `k` (same arithmetic as exp / log above)
CMX_LEAN_FN double exp(double x, const PinnedCoefs &k) {
    const double xc = __builtin_fmin(__builtin_fmax(x, k.exp_lo), k.exp_hi);
    const double n = __builtin_rint(xc * k.log2e);
    double t = fma_(-n, k.ln2_hi, xc);
    t = fma_(-n, k.ln2_lo, t);
    double p = k.e[0];
    for (int i = 1; i < 13; ++i) p = fma_(p, t, k.e[i]);
    p = fma_(p, t, 1.0);
    const double r = ldexp_(p, (int)n);
    return x != x ? x : r;
}
CMX_LEAN_FN double log(double x, const PinnedCoefs &k) {
    double m = frexp_mant(x);
    int e = frexp_exp(x);
    const bool lo = m < k.sqrt_half;
    m = lo ? m + m : m;
    e = lo ? e - 1 : e;
    const double s = (m - 1.0) * rcp_finite(m + 1.0);
    const double z = s * s;
    double p = k.l[0];
    for (int i = 1; i < 10; ++i) p = fma_(p, z, k.l[i]);
    const double lm = fma_(2.0 * s * z, p, 2.0 * s);
    const double ed = (double)e;
    return log_special(x, fma_(ed, k.ln2_hi, fma_(ed, k.ln2_lo, lm)));
}

// √x and 1/√x: hardware rsq seed + two coupled Newton (Goldschmidt) steps; x = 0 / Inf / < 0 follow IEEE
CMX_LEAN_FN double rsqrt_core(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    double y = __builtin_amdgcn_rsq(x);
#else
    double y = (double)(1.0f / std::sqrt((float)x));
    if (!(x > 1e-37 && x < 1e37)) y = 1.0 / std::sqrt(x);
#endif
    // y ← y (1.5 − 0.5 x y²), twice
    double h = 0.5 * x;
    y = fma_(y, fma_(-h * y, y, 0.5), y);
    y = fma_(y, fma_(-h * y, y, 0.5), y);
    return y;
}
CMX_LEAN_FN double sqrt(double x) {
    // scale subnormal/huge inputs out of the seed's weak range is unnecessary here: callers pass physical magnitudes;
    // the special values are restored explicitly
    const double y = rsqrt_core(x);
    double s = x * y;
    s = fma_(fma_(-s, s, x), 0.5 * y, s);                                      // one residual correction → ≤ 1 ulp
    const double inf = std::numeric_limits<double>::infinity();
    s = (x == 0.0 || x == inf) ? x : s;
    return s;                                                                 // x < 0 → NaN from the seed
}
CMX_LEAN_FN double rsqrt(double x) {
    const double inf = std::numeric_limits<double>::infinity();
    const double y = rsqrt_core(x);
    return x == 0.0 ? inf : (x == inf ? 0.0 : y);
}

}  // namespace lean
}  // namespace cmx
