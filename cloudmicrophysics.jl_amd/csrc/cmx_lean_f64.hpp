// cmx_lean_f64.hpp — lean Float64 elementary functions for gfx950.
//
// gfx950 has no Float64 transcendental unit; the OCML routines the compiler links for exp2/log2/exp/log cost
// 50–110 VALU instructions each (correctly rounded-ish, every special case, double-double range reduction), which
// makes every Float64 kernel of this library VALU-issue-bound (one f64 instruction = 4 cycles per wave).  The
// functions here are TABLE-DRIVEN (round 2): exp2 / exp reduce to |r| ≤ 1/256 with a 128-entry table of 2^(j/128) and a
// degree-4 polynomial (13 Float64-rate instructions + 4 integer / LDS ones; the round-1 Cody–Waite + degree-13 version took 23),
// log2 / log index a 128-entry table of (1/c, log2 c) by the top mantissa bits — integer operations on the high word, no
// reciprocal, degree-6 polynomial in r = z/c − 1 (9 Float64-rate instructions + 7 integer / LDS; round 1: ≈30).  On the
// device the tables live in LDS (3 KiB per workgroup, filled by `tables_init()` — every kernel that evaluates Float64
// functions calls Math<FT>::prepare() first); the host build reads them from static arrays.  ≤ 2 ulp (exp2, exp), ≤ 4 ulp
// (log2, log; relative accuracy kept for arguments near 1: the table interval around 1 has c = 1 exactly) — three to four
// decimal orders below the parity tolerance of the Float64 kernels (1e-6) — and the IEEE special values the rate functions
// rely on are kept (log2(0) = −Inf, log2(x<0) = NaN, exp2(±Inf), NaN propagation, gradual underflow via ldexp).
//
// The same source compiles for the host (plain C++) so tests/test_lean_math.py can check it against libm.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>

#include "cmx_lean_tables.inc"
#include "cmx_erfc_table.inc"

#if defined(__HIPCC__)
#define CMX_LEAN_FN __host__ __device__ __forceinline__   // hipcc: both passes see the same overload set
#else
#define CMX_LEAN_FN inline                                // plain host build (tests/native/lean_math_host.cpp)
#endif

namespace cmx {
namespace lean {

CMX_LEAN_FN double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }

#ifndef CMX_LEAN_RCP_NEWTON
#define CMX_LEAN_RCP_NEWTON 2
#endif
// 1/d for finite, non-zero d: hardware seed (≈2⁻²⁴ relative on gfx950) + two Newton steps → ≤ 1 ulp
CMX_LEAN_FN double rcp_finite(double d) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(d);
#else
    double r = (double)(1.0f / (float)d);   // host stand-in for the seed (tests only; same precision class)
    if (!(std::fabs(d) > 1e-37 && std::fabs(d) < 1e37)) r = 1.0 / d;
#endif
    r = fma_(r, fma_(-d, r, 1.0), r);
#if CMX_LEAN_RCP_NEWTON < 2
    return r;                                 // A/B only: ≈ 2⁻⁴⁸ relative (16 ulp)
#endif
    return fma_(r, fma_(-d, r, 1.0), r);
}
// 1/d for finite, non-zero d to 2⁻⁴⁸ relative (≈ 16 ulp): the seed and ONE Newton step — 3 instructions against 8 for the full rcp
// below (second step, class test, two selects).  For the reciprocals of the point functions whose argument is a positive physical
// quantity (Math<double>::rcp_nz): the result multiplies terms that carry the ≈ 1 ulp of every table-driven function anyway, and the
// Float64 parity bound is 1e-6.  CMX_F64_FINITE_FORMS=0 maps it back to the full form.  NaN propagates.
CMX_LEAN_FN double rcp_nz(double d) {
#if defined(__HIP_DEVICE_COMPILE__)
    const double r = __builtin_amdgcn_rcp(d);
#else
    int e; const double m = std::frexp(1.0 / d, &e);
    const double r = std::ldexp((double)(float)m, e);   // host stand-in for the seed (tests only): 1/d rounded to 24 bits
#endif
    return fma_(r, fma_(-d, r, 1.0), r);
}
// 1/d with the IEEE results for d = ±0 (±Inf), ±Inf (±0) and NaN: whenever the seed is 0 or ±Inf the Newton steps
// would produce NaN, so the seed itself is returned (one class test on the seed; subnormal d saturates to ±Inf)
CMX_LEAN_FN double rcp(double d) {
#if defined(__HIP_DEVICE_COMPILE__)
    const double r0 = __builtin_amdgcn_rcp(d);
    double r = fma_(r0, fma_(-d, r0, 1.0), r0);
#if CMX_LEAN_RCP_NEWTON >= 2
    r = fma_(r, fma_(-d, r, 1.0), r);
#endif
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
// ---- tables (cmx_lean_tables.inc, generated by tools/gen_lean_tables.py) and polynomial coefficients ----------------------------
// A Float64 constant that is not one of the few inline values needs an SGPR pair (VOP3 takes no literal): the ≈25 coefficients
// below, on top of a Float64 kernel's 60–100 parameters, overflow the 102-SGPR file, and the compiler then either parks them in
// VGPR lanes (v_writelane / v_readlane: VALU instructions) or rematerialises each with two v_mov_b32 before a v_fmac_f64 — measured
// 36 % of the VALU instructions of the round-1 SB2006 Float64 kernel.  On the device the coefficients therefore live in LDS next to
// the tables (uniform-address ds_read_b64: issued on the LDS port, not the VALU port; the register allocator keeps as many of them
// in VGPRs as the kernel's budget allows).  The host build reads the same struct from a static constant.
struct Log2Entry { double invc, logc; };
struct LeanCoefs {
    double x_lo, x_hi, k128, inv128;            // exp2: clamp, 128, −1/128
    double e2[4];                               // (2ʳ − 1)/r on |r| ≤ 1/256, degree 3, highest power first (round 4: Chebyshev fit, 1.5e-16 relative to 2ʳ)
    double ex_lo, ex_hi, k128_log2e, ln2_128_hi, ln2_128_lo;
    double ee[4];                               // (eʳ − 1)/r on |r| ≤ ln2/256, degree 3
    double l2[6];                               // log2(1 + r)/r on |r| ≤ 2⁻⁸, degree 5 (4.6e-17 relative, ln: 3.2e-17; the constant term is log2 e exactly)
    double ln[6];                               // ln(1 + r)/r likewise (constant term 1, linear term −½ exactly)
    double minus_one, ln2, two64, sixty4;
};
// Round 4: the polynomials are one term shorter than the Taylor forms of round 2 (exp: degree 3 instead of 4 for (2ʳ − 1)/r, log: degree 5
// instead of 6 for log2(1 + r)/r) — Chebyshev interpolants on the reduced interval (mpmath, 200 bits).  What that costs in accuracy (ADVICE r04,
// re-measured with mpmath): the degree-3 fits of (2ʳ − 1)/r and (eʳ − 1)/r carry 1.5e-16 relative to 2ʳ / eʳ where the Taylor truncation carried
// 5e-19 — about 0.7 ulp more, exp2 / exp go from ≤ 1 to ≤ 2 ulp (tests/test_lean_math.py asserts 2); the degree-5 log fits carry 3.2e-17 (ln) and
// 4.6e-17 (log2), still below one ulp, so log2 / log stay ≤ 4 ulp.  Five decimal orders below the Float64 parity bound of 1e-6 either way.  One Float64 FMA less per call: ≈ 22 of the 621
// instructions of an SB2006 point.
#define CMX_LEAN_COEFS                                                                                                          \
    {-1100.0, 1100.0, 128.0, -0.0078125,                                                                                       \
     {0.009618131458020956, 0.05550412901021981, 0.24022650695909623, 0.6931471805599065},                                     \
     -760.0, 760.0, 184.66496523378732, 0x1.62e42fefa0000p-8, 1.2864023111638345e-14,                                        \
     {0.04166667684879449, 0.16666672775943595, 0.4999999999999907, 0.999999999999944},                                        \
     {-0.24045330112179908, 0.2885437254791991, -0.36067376019862224, 0.4808983469359951, -0.7213475204444817,                 \
      1.4426950408889634},                                                                                                     \
     {-0.16666952772890656, 0.20000326978416968, -0.24999999998362882, 0.3333333333146234, -0.5, 1.0},                         \
     -1.0, 0.6931471805599453, 1.8446744073709552e19, 64.0}
#if defined(__HIP_DEVICE_COMPILE__)
static __device__ const double kExp2Tab[128] = {CMX_LEAN_EXP2_TABLE};
static __device__ const Log2Entry kLog2Tab[128] = {CMX_LEAN_LOG2_TABLE};
static __device__ const LeanCoefs kLeanCoefs = CMX_LEAN_COEFS;
// workgroup copies in LDS: a per-lane indexed lookup is one ds_read (b64 / b128) there
__device__ __forceinline__ double *lds_exp2_tab() { __shared__ double t[128]; return t; }
__device__ __forceinline__ Log2Entry *lds_log2_tab() { __shared__ __align__(16) Log2Entry t[128]; return t; }
__device__ __forceinline__ LeanCoefs *lds_coefs() { __shared__ __align__(16) LeanCoefs t; return &t; }
// Where the polynomial coefficients live (per translation unit, csrc/Makefile): 1 = in LDS next to the tables, 0 = literals pinned to
// SGPR pairs (sc() below).  With the kernel constants phase-local (cmx_math.hpp consts_after) the SGPR file has room for them again;
// same-box A/B, round 2, ms per 1e8 Float64 points, LDS → literals: SB2006 3.20 → 3.10, 1-moment 3.29 → 3.17, ice nucleation 0.71 →
// 0.68; the column kernel 3.57 → 3.70 (stays with LDS); ARG 3.07 → 3.13 at first, but 2.79 → 2.59 once its erfc was table-driven too
// (the erfc table reads and the coefficient reads competed for the LDS port): literals there as well.
#ifndef CMX_LEAN_COEFS_IN_LDS
#if defined(CMX_LEAN_COEFS_LIT_TU)
#define CMX_LEAN_COEFS_IN_LDS 0
#else
#define CMX_LEAN_COEFS_IN_LDS 1
#endif
#endif
// every thread of the workgroup calls this once before the first exp2 / log2 (ends with a barrier)
__device__ __forceinline__ void tables_init() {
    double *e = lds_exp2_tab();
    Log2Entry *l = lds_log2_tab();
    for (int i = threadIdx.x; i < 128; i += blockDim.x) {
        e[i] = kExp2Tab[i];
        l[i] = kLog2Tab[i];
    }
#if CMX_LEAN_COEFS_IN_LDS
    double *k = reinterpret_cast<double *>(lds_coefs());
    const double *ks = reinterpret_cast<const double *>(&kLeanCoefs);
    for (int i = threadIdx.x; i < (int)(sizeof(LeanCoefs) / sizeof(double)); i += blockDim.x) k[i] = ks[i];
#endif
    __syncthreads();
}
__device__ __forceinline__ double exp2_tab(int j) { return lds_exp2_tab()[j]; }
__device__ __forceinline__ Log2Entry log2_tab(int j) { return lds_log2_tab()[j]; }
#if CMX_LEAN_COEFS_IN_LDS
__device__ __forceinline__ const LeanCoefs &coefs() { return *lds_coefs(); }
#else
__device__ __forceinline__ LeanCoefs coefs() { return LeanCoefs CMX_LEAN_COEFS; }     // literals (A/B switch)
#endif
__device__ __forceinline__ int32_t hi_word(double x) { return __double2hiint(x); }
__device__ __forceinline__ int32_t lo_word(double x) { return __double2loint(x); }
__device__ __forceinline__ double from_words(int32_t hi, int32_t lo) { return __hiloint2double(hi, lo); }
#else
static const double kExp2Tab[128] = {CMX_LEAN_EXP2_TABLE};
static const Log2Entry kLog2Tab[128] = {CMX_LEAN_LOG2_TABLE};
static const LeanCoefs kLeanCoefs = CMX_LEAN_COEFS;
inline void tables_init() {}
inline double exp2_tab(int j) { return kExp2Tab[j]; }
inline Log2Entry log2_tab(int j) { return kLog2Tab[j]; }
inline const LeanCoefs &coefs() { return kLeanCoefs; }
inline int32_t hi_word(double x) { uint64_t u; std::memcpy(&u, &x, 8); return (int32_t)(u >> 32); }
inline int32_t lo_word(double x) { uint64_t u; std::memcpy(&u, &x, 8); return (int32_t)(uint32_t)u; }
inline double from_words(int32_t hi, int32_t lo) { const uint64_t u = ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo; double x; std::memcpy(&x, &u, 8); return x; }
#endif

// 2ʳ − 1 for |r| ≤ 1/256: r·(C1 + r·(C2 + r·(C3 + r·C4))), a Chebyshev fit of (2ʳ − 1)/r (LeanCoefs::e2; error 1.5e-16 relative to 2ʳ)
// sc(c): a polynomial coefficient as an SGPR pair.  With the coefficients written as literals (CMX_LEAN_COEFS_IN_LDS=0) the
// compiler is free to build each one in a VGPR pair instead (two v_mov_b32 per coefficient in front of a v_fmac_f64 — VALU
// instructions on the port that bounds these kernels); the empty asm pins it to the scalar side (two s_mov_b32, SALU port).
#if defined(__HIP_DEVICE_COMPILE__) && !CMX_LEAN_COEFS_IN_LDS
__device__ __forceinline__ double sc(double c) { asm volatile("" : "+s"(c)); return c; }
#else
CMX_LEAN_FN double sc(double c) { return c; }
#endif
// vc(c): the SECOND coefficient of a polynomial in a VGPR pair that the whole kernel shares.  A VOP3 instruction takes one scalar operand, so
// the first Horner step c0·r + c1 with both coefficients on the scalar side costs a v_mov_b64 per evaluation (19 per SB2006 point); the
// non-volatile asm is a pure function of its operand, so all uses of one constant fold into ONE move per kernel.  Two registers per
// polynomial: translation units whose kernels are at their register limit leave it off (CMX_LEAN_VGPR_C1, per file in the Makefile).
#ifndef CMX_LEAN_VGPR_C1
#define CMX_LEAN_VGPR_C1 0
#endif
#if defined(__HIP_DEVICE_COMPILE__) && !CMX_LEAN_COEFS_IN_LDS && CMX_LEAN_VGPR_C1
__device__ __forceinline__ double vc(double c) { asm("" : "+v"(c)); return c; }
#else
CMX_LEAN_FN double vc(double c) { return sc(c); }
#endif
CMX_LEAN_FN double exp2m1_small(double r, const LeanCoefs &K) {
    double p = sc(K.e2[0]);
    p = fma_(p, r, vc(K.e2[1]));
    p = fma_(p, r, sc(K.e2[2]));
    p = fma_(p, r, sc(K.e2[3]));
    return p * r;
}
// eʳ − 1 for |r| ≤ ln2/256
CMX_LEAN_FN double expm1_small(double r, const LeanCoefs &K) {
    double p = sc(K.ee[0]);
    p = fma_(p, r, vc(K.ee[1]));
    p = fma_(p, r, sc(K.ee[2]));
    p = fma_(p, r, sc(K.ee[3]));
    return p * r;
}
// int conversion that saturates (the device's v_cvt_i32_f64 does; the host cast would be undefined out of range)
CMX_LEAN_FN int sat_int(double kd) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (int)kd;
#else
    return kd >= 2147483647.0 ? 2147483647 : (kd <= -2147483648.0 ? (-2147483647 - 1) : (int)kd);
#endif
}
CMX_LEAN_FN bool is_nan_or_inf(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_class(x, 0x003 | 0x204);                          // NaN | ±Inf
#else
    return !(__builtin_fabs(x) < std::numeric_limits<double>::infinity());
#endif
}
// 2ˣ = 2ᵉ · T[j] · 2ʳ,  x = e + j/128 + r.  Branch-free on purpose: a "rare" special-value branch per call cuts the kernel into
// small basic blocks and, at the 2–4 waves per SIMD a Float64 kernel runs at, exposes the latency of every dependent chain
// (measured: SB2006 Float64 3.15 → 3.72 ms with a branch here).
#ifndef CMX_LEAN_TABLES
#define CMX_LEAN_TABLES 1      // 0: the round-1 formulations (Cody–Waite + degree-13 exp, atanh-series log) — kept for A/B runs (tools/ab_bench.sh)
#endif
CMX_LEAN_FN double exp2(double x) {
#if !CMX_LEAN_TABLES
    {
        const double xc = __builtin_fmin(__builtin_fmax(x, -1100.0), 1100.0);
        const double n = __builtin_rint(xc);
        const double t = (xc - n) * 0.6931471805599453;
        const double r = ldexp_(exp_poly(t), (int)n);
        return x != x ? x : r;
    }
#endif
    const LeanCoefs &K = coefs();
    const double xc = __builtin_fmin(__builtin_fmax(x, K.x_lo), K.x_hi);      // keeps the int conversion in range
    const double kd = __builtin_rint(xc * K.k128);
    const double r = fma_(kd, K.inv128, xc);                                  // exact
    const int ki = sat_int(kd);
    const double s = exp2_tab(ki & 127);
    const double y = ldexp_(fma_(s, exp2m1_small(r, K), s), ki >> 7);         // overflow → +Inf, underflow → denormal/0
    return x != x ? x : y;
}
// 2ˣ for an argument the CALLER knows to be finite or NaN (round 3): no clamp and no NaN select — five of the ≈ 21 instructions.
// Finite x of any size is still right: the int conversion saturates, its low 7 bits pick a table entry in [1, 2) and v_ldexp_f64
// turns the saturated exponent into +Inf / 0; r stays exact.  NaN propagates through r.  What is NOT handled is x = ±Inf (r = Inf − Inf
// → NaN where 2^±Inf = Inf / 0 is meant): call sites are those whose argument is a finite combination of log2 of positive, clamped
// quantities (each site states why; the rule is DESIGN.md §4.3).  CMX_F64_FINITE_FORMS=0 maps them back to the full forms.
#ifndef CMX_F64_FINITE_FORMS
#define CMX_F64_FINITE_FORMS 1
#endif
CMX_LEAN_FN double exp2_fin(double x) {
#if !CMX_F64_FINITE_FORMS || !CMX_LEAN_TABLES
    return exp2(x);
#else
    const LeanCoefs &K = coefs();
    const double kd = __builtin_rint(x * K.k128);
    const double r = fma_(kd, K.inv128, x);                                   // exact
    const int ki = sat_int(kd);
    const double s = exp2_tab(ki & 127);
    return ldexp_(fma_(s, exp2m1_small(r, K), s), ki >> 7);
#endif
}
// eˣ: the same with x·log2e split off by a two-part ln2/128 (Cody–Waite)
CMX_LEAN_FN double exp(double x) {
    const LeanCoefs &K = coefs();
    const double xc = __builtin_fmin(__builtin_fmax(x, K.ex_lo), K.ex_hi);
    const double kd = __builtin_rint(xc * K.k128_log2e);                      // 128·log2 e
    double r = fma_(-kd, K.ln2_128_hi, xc);
    r = fma_(-kd, K.ln2_128_lo, r);
    const int ki = (int)kd;
    const double s = exp2_tab(ki & 127);
    const double y = ldexp_(fma_(s, expm1_small(r, K), s), ki >> 7);
    return x != x ? x : y;
}
// eˣ for a finite-or-NaN argument with |x| < 2e13 (see exp2_fin; beyond that 128 x/ln2 is no longer an exact integer and the
// reduced argument is garbage — the one caller, erfc, passes [−800, 0])
CMX_LEAN_FN double exp_fin(double x) {
#if !CMX_F64_FINITE_FORMS || !CMX_LEAN_TABLES
    return exp(x);
#else
    const LeanCoefs &K = coefs();
    const double kd = __builtin_rint(x * K.k128_log2e);
    double r = fma_(-kd, K.ln2_128_hi, x);
    r = fma_(-kd, K.ln2_128_lo, r);
    const int ki = sat_int(kd);
    const double s = exp2_tab(ki & 127);
    return ldexp_(fma_(s, expm1_small(r, K), s), ki >> 7);
#endif
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
// log2 z = k + logc + log2(1 + r): table index and exponent from the high word (integer ops), r = z·invc − 1 by one fma.
// Valid for positive normal x; everything else goes through log_special / the subnormal rescue.
struct Log2Parts { double hi, r; };
CMX_LEAN_FN Log2Parts log2_reduce(double x, const LeanCoefs &K) {
    const int32_t hw = hi_word(x);
    const int32_t tmp = hw - (int32_t)CMX_LEAN_LOG2_OFF_HI;
    const int32_t k = tmp >> 20;                                              // arithmetic: floor
    const double z = from_words(hw - (int32_t)((uint32_t)tmp & 0xFFF00000u), lo_word(x));   // z ∈ [OFF, 2·OFF)
    const Log2Entry e = log2_tab((tmp >> 13) & 127);
    return {(double)k + e.logc, fma_(z, e.invc, K.minus_one)};
}
// log2(1 + r)/r = B1 + r·(B2 + … r·B6) on |r| ≤ 2⁻⁸, a Chebyshev fit (LeanCoefs::l2; error 1.6e-17 relative, B1 = log2 e exactly); ln likewise
CMX_LEAN_FN double poly6s(double r, const double (&b)[6]) {                   // coefficients on the scalar side (sc above)
    double p = sc(b[0]);
    p = fma_(p, r, vc(b[1]));
    p = fma_(p, r, sc(b[2]));
    p = fma_(p, r, sc(b[3]));
    p = fma_(p, r, sc(b[4]));
    return fma_(p, r, sc(b[5]));
}
CMX_LEAN_FN double poly6(double r, const double (&b)[6]) {
    double p = b[0];
    p = fma_(p, r, b[1]);
    p = fma_(p, r, b[2]);
    p = fma_(p, r, b[3]);
    p = fma_(p, r, b[4]);
    return fma_(p, r, b[5]);
}
CMX_LEAN_FN bool log_needs_rescue(double x) {            // anything but a positive normal number
#if defined(CMX_LEAN_NO_RESCUE)
    return false;                                        // measurement only: what the rescue branches cost (results wrong for specials)
#elif defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_class(x, 0x003 | 0x200 | 0x060 | 0x01C | 0x080);   // NaN | +Inf | ±0 | −Inf, −normal, −subnormal | +subnormal
#else
    return !(x >= 2.2250738585072014e-308 && x < std::numeric_limits<double>::infinity());
#endif
}
// Arguments that are not positive normal numbers (subnormals, ±0, negatives, ±Inf, NaN) are rare.  Two forms, both exact:
//   CMX_LEAN_RESCUE_REPEAT=1 (default): ONE block behind a class test repeats the whole evaluation on the argument scaled by 2⁶⁴ and
//     puts the IEEE special values in.  No lane enters it for ordinary arguments.
//   CMX_LEAN_RESCUE_REPEAT=0: two SMALL blocks — the argument is scaled before the reduction (0, ±Inf, NaN and the sign survive the
//     scaling), 64 is taken off and the special values put in after the polynomial, which is evaluated once.  130 fewer static
//     instructions per SB2006 point and no coefficient shared between a block and the main path — and measured SLOWER (same-box
//     A/B, round 2, ms per 1e8 Float64 points: SB2006 3.23 → 3.37, ARG 3.14 → 3.67, column 3.58 → 4.35): two branches per log cut
//     the main path into twice as many basic blocks, and at 3–4 waves per SIMD the scheduler's freedom inside a block is what
//     hides the 11-cycle dependent-issue latency of a Float64 instruction (tools/valu_probe).  The empty asm in the blocks keeps
//     the optimiser from turning them into selects on the main path.
#ifndef CMX_LEAN_RESCUE_REPEAT
#define CMX_LEAN_RESCUE_REPEAT 1
#endif
CMX_LEAN_FN void rare_block(double &x) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(x));
#else
    (void)x;
#endif
}
CMX_LEAN_FN double log_prescale(double x, double two64) {                     // inside `if (log_needs_rescue(x))`
    double xs = x * two64;
    rare_block(xs);
    return xs;
}
CMX_LEAN_FN double log_postfix(double xs, double y_minus_bias) {              // xs: the scaled argument (sign, 0, Inf, NaN as the original)
    double y = xs != xs ? xs : log_special(xs, y_minus_bias);
    rare_block(y);
    return y;
}
CMX_LEAN_FN double log2(double x) {
#if !CMX_LEAN_TABLES
    {
        const MantExp s = split(x);
        return log_special(x, fma_(log_mant(s.m), 1.4426950408889634, s.e));
    }
#endif
    const LeanCoefs &K = coefs();
#if CMX_LEAN_RESCUE_REPEAT
    const Log2Parts q = log2_reduce(x, K);
    double y = fma_(q.r, poly6s(q.r, K.l2), q.hi);
    if (log_needs_rescue(x)) {                                                // no lane takes this for ordinary arguments
        const Log2Parts s = log2_reduce(x * K.two64, K);                      // positive subnormal: ×2⁶⁴
        y = fma_(s.r, poly6s(s.r, K.l2), s.hi - K.sixty4);
        y = x != x ? x : log_special(x, y);
    }
    return y;
#else
    const bool rare = log_needs_rescue(x);                                    // no lane takes the two blocks for ordinary arguments
    if (rare) x = log_prescale(x, K.two64);
    const Log2Parts q = log2_reduce(x, K);
    double y = fma_(q.r, poly6s(q.r, K.l2), q.hi);
    if (rare) y = log_postfix(x, y - K.sixty4);
    return y;
#endif
}
// log2 x for a POSITIVE NORMAL finite x: the main path of log2() alone — no class test and no rescue block.  For arguments that are ordinarily
// zero (ρ·q of an absent species: half the points of a real field, so nearly every wave would run log2()'s "rare" block — 55 instructions per
// logarithm for the whole wave), floored at the smallest normal number by the caller (cmx_math.hpp log2_floored).  0, negatives, subnormals, Inf
// and NaN are outside the contract (NaN comes out as ≈ 1024).
CMX_LEAN_FN double log2_pos(double x) {
#if !CMX_LEAN_TABLES || !CMX_F64_FINITE_FORMS
    return log2(x);
#else
    const LeanCoefs &K = coefs();
    const Log2Parts q = log2_reduce(x, K);
    return fma_(q.r, poly6s(q.r, K.l2), q.hi);
#endif
}
CMX_LEAN_FN double log(double x);
// ln x for a POSITIVE NORMAL finite x (a diameter or an area at an interior quadrature node): the main path of log() alone — no class test, no
// rescue block, so the call does not end a basic block.  0, negatives, subnormals, Inf AND NaN are outside the contract: a NaN argument is reduced
// like a number of exponent 1024 and comes out as ≈ 710.  The P3 integrands that call it also feed the same x linearly into every result (−λ x in
// the number density that multiplies each integrand), so a NaN node still yields a NaN integral (tests/test_nan_inputs_gpu.py).
CMX_LEAN_FN double log_pos(double x) {
#if !CMX_LEAN_TABLES || !CMX_F64_FINITE_FORMS
    return log(x);
#else
    const LeanCoefs &K = coefs();
    const Log2Parts q = log2_reduce(x, K);
    return fma_(q.hi, K.ln2, q.r * poly6s(q.r, K.ln));
#endif
}
// exp_fin / log_pos with the polynomial's SECOND coefficient handed in by the caller (round 5).  A VOP3 instruction reads one scalar operand, so the
// first Horner step c0·r + c1 with both coefficients on the scalar side costs a v_mov_b64 per evaluation; vc() above removes it with a VGPR pair that
// lives through the whole kernel.  A kernel at its register limit in ONE phase and with slack in another (the 2M + P3 step: 163–168 VGPRs in the
// collision sweep, ≈ 107 in the self-collection / melting sweeps) pins the pair for the slack phase only (cmx_p3.hpp PM<double>::coefs_local) and passes it
// in — the same arithmetic, the same coefficient value.
CMX_LEAN_FN double exp_fin_c1(double x, double c1) {
#if !CMX_F64_FINITE_FORMS || !CMX_LEAN_TABLES
    (void)c1;
    return exp(x);
#else
    const LeanCoefs &K = coefs();
    const double kd = __builtin_rint(x * K.k128_log2e);
    double r = fma_(-kd, K.ln2_128_hi, x);
    r = fma_(-kd, K.ln2_128_lo, r);
    const int ki = sat_int(kd);
    const double s = exp2_tab(ki & 127);
    double p = sc(K.ee[0]);
    p = fma_(p, r, c1);
    p = fma_(p, r, sc(K.ee[2]));
    p = fma_(p, r, sc(K.ee[3]));
    return ldexp_(fma_(s, p * r, s), ki >> 7);
#endif
}
CMX_LEAN_FN double log_pos_c1(double x, double c1) {
#if !CMX_LEAN_TABLES || !CMX_F64_FINITE_FORMS
    (void)c1;
    return log(x);
#else
    const LeanCoefs &K = coefs();
    const Log2Parts q = log2_reduce(x, K);
    double p = sc(K.ln[0]);
    p = fma_(p, q.r, c1);
    p = fma_(p, q.r, sc(K.ln[2]));
    p = fma_(p, q.r, sc(K.ln[3]));
    p = fma_(p, q.r, sc(K.ln[4]));
    p = fma_(p, q.r, sc(K.ln[5]));
    return fma_(q.hi, K.ln2, q.r * p);
#endif
}
CMX_LEAN_FN double log(double x) {
#if !CMX_LEAN_TABLES
    {
        const MantExp s = split(x);
        return log_special(x, fma_(s.e, 0.6931471803691238, fma_(s.e, 1.9082149292705877e-10, log_mant(s.m))));
    }
#endif
    const LeanCoefs &K = coefs();
#if CMX_LEAN_RESCUE_REPEAT
    const Log2Parts q = log2_reduce(x, K);
    double y = fma_(q.hi, K.ln2, q.r * poly6s(q.r, K.ln));
    if (log_needs_rescue(x)) {
        const Log2Parts s = log2_reduce(x * K.two64, K);
        y = fma_(s.hi - K.sixty4, K.ln2, s.r * poly6s(s.r, K.ln));
        y = x != x ? x : log_special(x, y);
    }
    return y;
#else
    const bool rare = log_needs_rescue(x);
    if (rare) x = log_prescale(x, K.two64);
    const Log2Parts q = log2_reduce(x, K);
    const double rp = q.r * poly6s(q.r, K.ln);
    double y = fma_(q.hi, K.ln2, rp);
    if (rare) y = log_postfix(x, fma_(q.hi - K.sixty4, K.ln2, rp));
    return y;
#endif
}
// ---- erfc (round 2): table-driven, for the activated NUMBER fractions of the ARG kernel ------------------------------------------------
// erfc(x) = e^(−x²)·E(x), x ≥ 0, with E = erfcx as a degree-8 polynomial in s = x − centre on 64 intervals of [0, 6.5]
// (cmx_erfc_table.inc, tools/gen_erfc_table.py: ≤ 2.4e-16 relative per interval); erfc(−x) = 2 − erfc(x).  One rounding of x² enters the
// exponent, so the relative error is ≤ (2 + x²)·2e-16 against mpmath (9e-15 at 6.5, where erfc = 4e-20); beyond 6.5 E(6.5) is used (erfc < 4e-20:
// the value is right to a factor x/6.5).  ≈ 36 VALU instructions, branch-free — OCML's erfc is 135, and five of them were 70 % of the
// Float64 ARG kernel.  NOT for arguments whose tiny result must keep full relative accuracy beyond 6.5 (the activated MASS, which the
// reference evaluates with erfc itself, stays on OCML).  The kernel that uses it fills the LDS copy with erfc_tab_fill() before
// Math<double>::prepare() (whose barrier publishes it).
#if defined(__HIP_DEVICE_COMPILE__)
static __device__ __align__(16) const double kErfcTab[CMX_LEAN_ERFC_NINT * CMX_LEAN_ERFC_ROW] = {CMX_LEAN_ERFC_TABLE};
__device__ __forceinline__ double *lds_erfc_tab() { __shared__ __align__(16) double t[CMX_LEAN_ERFC_NINT * CMX_LEAN_ERFC_ROW]; return t; }
__device__ __forceinline__ void erfc_tab_fill() {          // 16 bytes per lane and trip (the table is an even number of doubles)
    static_assert((CMX_LEAN_ERFC_NINT * CMX_LEAN_ERFC_ROW) % 2 == 0, "erfc table: even number of doubles");
    double2 *t = reinterpret_cast<double2 *>(lds_erfc_tab());
    const double2 *g = reinterpret_cast<const double2 *>(kErfcTab);
    for (int i = threadIdx.x; i < CMX_LEAN_ERFC_NINT * CMX_LEAN_ERFC_ROW / 2; i += blockDim.x) t[i] = g[i];
}
__device__ __forceinline__ const double *erfc_row(int j) { return lds_erfc_tab() + j * CMX_LEAN_ERFC_ROW; }
#else
static const double kErfcTab[CMX_LEAN_ERFC_NINT * CMX_LEAN_ERFC_ROW] = {CMX_LEAN_ERFC_TABLE};
inline void erfc_tab_fill() {}
inline const double *erfc_row(int j) { return kErfcTab + j * CMX_LEAN_ERFC_ROW; }
#endif
CMX_LEAN_FN double erfc(double x) {
    const double ax = __builtin_fabs(x);
    const double xc = __builtin_fmin(ax, CMX_LEAN_ERFC_XMAX * (1.0 - 0x1p-52));      // polynomial range [0, 6.5)
    const int j = (int)(xc * CMX_LEAN_ERFC_INV_H);                                  // 0 … NINT−1 (a NaN argument: 0, fixed up below)
    const double sft = fma_((double)j, -CMX_LEAN_ERFC_H, xc) - 0.5 * CMX_LEAN_ERFC_H;   // x − centre of interval j, |s| ≤ h/2
    const double *c = erfc_row(j);
    double p = c[CMX_LEAN_ERFC_DEG];
#pragma unroll
    for (int k = CMX_LEAN_ERFC_DEG - 1; k >= 0; --k) p = fma_(p, sft, c[k]);
#if CMX_F64_FINITE_FORMS
    // e^(−x²) = 0 below −745: one max keeps ±Inf arguments (S_max = 0 ⇒ u = +Inf in the ARG kernel) out of the finite-argument exp; it
    // also swallows a NaN, which the final select puts back
    const double r = exp_fin(__builtin_fmax(-(ax * ax), -800.0)) * p;
    const double y = x < 0.0 ? 2.0 - r : r;
    return x != x ? x : y;
#else
    const double r = exp(-(ax * ax)) * p;
    const double y = x < 0.0 ? 2.0 - r : r;
    return x != x ? x : y;
#endif
}

// ---- ln Γ(z) for z > 0 (round 2): the P3 shape solver calls it three times per residual evaluation -----------------------------------
// Stirling's series at w = z + 7 (z < 8) or w = z:  ln Γ(w) = (w − ½) ln w − w + ½ ln 2π + 1/(12w) − 1/(360w³) + … (8 terms: the first
// omitted one is 4e-16 at w = 8), minus ln(z(z+1)…(z+6)) for the shift.  Branch-free, two table-driven logs and one reciprocal:
// ≈ 95 instructions (OCML's lgamma: 934 static instructions, 16 branches).  ABSOLUTE accuracy ≈ 1e-16·(|w ln w| + |ln Π|) — 5e-15
// for z ≤ 11, measured ≤ 1e-14 on (0, 12]; the relative accuracy at the zeros z = 1, 2 is NOT kept (the P3 sums add it to terms of
// order 10).  z > 0 only (z ≤ 0, NaN: not meaningful).
CMX_LEAN_FN double lgamma_pos(double z) {
    const bool shift = z < 8.0;
    const double w = shift ? z + 7.0 : z;
    double prod = z * (z + 1.0);
    prod *= (z + 2.0) * (z + 3.0);
    prod *= (z + 4.0) * (z + 5.0);
    prod *= z + 6.0;
    prod = shift ? prod : 1.0;
    const double r = rcp_finite(w), r2 = r * r;
    double t = 3617.0 / 122400.0;
    t = fma_(-t, r2, 1.0 / 156.0);
    t = fma_(-t, r2, 691.0 / 360360.0);
    t = fma_(-t, r2, 1.0 / 1188.0);
    t = fma_(-t, r2, 1.0 / 1680.0);
    t = fma_(-t, r2, 1.0 / 1260.0);
    t = fma_(-t, r2, 1.0 / 360.0);
    t = fma_(-t, r2, 1.0 / 12.0);
    const double lw = log(w);
    return fma_(w - 0.5, lw, fma_(t, r, 0.9189385332046727418 - w)) - log(prod);
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
CMX_LEAN_FN double exp_fin(double x, const PinnedCoefs &k) {                  // finite |x| < 2e13 or NaN (see exp2_fin)
#if !CMX_F64_FINITE_FORMS
    return exp(x, k);
#else
    const double n = __builtin_rint(x * k.log2e);
    double t = fma_(-n, k.ln2_hi, x);
    t = fma_(-n, k.ln2_lo, t);
    double p = k.e[0];
    for (int i = 1; i < 13; ++i) p = fma_(p, t, k.e[i]);
    p = fma_(p, t, 1.0);
    return ldexp_(p, sat_int(n));
#endif
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
CMX_LEAN_FN double log_pos(double x, const PinnedCoefs &k) { return log(x, k); }   // (no shorter form in the table-free variant)

// ---- table-driven eˣ / ln x with the coefficients pinned in VGPRs (round 2) -------------------------------------------------------
// The quadrature loops of the P3 kernels evaluate ≈ 1 log + 3–4 exp per node, hundreds of nodes per state: the coefficients must not
// be rematerialised per call (PinnedCoefs above), and the table-driven forms need 9 fewer Float64 FMAs per exp (degree 4 instead of
// 13) and 9 fewer per log (degree 6, no reciprocal) for one LDS lookup each.  18 pinned values (36 VGPRs; PinnedCoefs: 29).
struct TabCoefs {
    double ee[4];                                // (eʳ − 1)/r, degree 3 (LeanCoefs::ee)
    double k128_log2e, ln2_128_hi, ln2_128_lo, ex_lo, ex_hi;
    double ln[6];                                // ln(1 + r)/r, degree 5 (LeanCoefs::ln)
    double ln2, two64, sixty4;
};
CMX_LEAN_FN TabCoefs tab_coefs() {
    TabCoefs k;
    const double ee[4] = {0.04166667684879449, 0.16666672775943595, 0.4999999999999907, 0.999999999999944};
    const double ln[6] = {-0.16666952772890656, 0.20000326978416968, -0.24999999998362882, 0.3333333333146234, -0.5, 1.0};
    for (int i = 0; i < 4; ++i) { k.ee[i] = ee[i]; pin(k.ee[i]); }
    for (int i = 0; i < 6; ++i) { k.ln[i] = ln[i]; pin(k.ln[i]); }
    k.k128_log2e = 184.66496523378732; k.ln2_128_hi = 0x1.62e42fefa0000p-8; k.ln2_128_lo = 1.2864023111638345e-14;
    k.ex_lo = -760.0; k.ex_hi = 760.0; k.ln2 = 0.6931471805599453; k.two64 = 1.8446744073709552e19; k.sixty4 = 64.0;
    pin(k.k128_log2e); pin(k.ln2_128_hi); pin(k.ln2_128_lo); pin(k.ex_lo); pin(k.ex_hi); pin(k.ln2); pin(k.two64); pin(k.sixty4);
    return k;
}
CMX_LEAN_FN double exp(double x, const TabCoefs &k) {
    const double xc = __builtin_fmin(__builtin_fmax(x, k.ex_lo), k.ex_hi);
    const double kd = __builtin_rint(xc * k.k128_log2e);
    double r = fma_(-kd, k.ln2_128_hi, xc);
    r = fma_(-kd, k.ln2_128_lo, r);
    const int ki = (int)kd;
    const double s = exp2_tab(ki & 127);
    double p = k.ee[0];
    for (int i = 1; i < 4; ++i) p = fma_(p, r, k.ee[i]);
    const double y = ldexp_(fma_(s, p * r, s), ki >> 7);
    return x != x ? x : y;
}
CMX_LEAN_FN double exp_fin(double x, const TabCoefs &k) {                     // finite |x| < 2e13 or NaN (see exp2_fin)
#if !CMX_F64_FINITE_FORMS
    return exp(x, k);
#else
    const double kd = __builtin_rint(x * k.k128_log2e);
    double r = fma_(-kd, k.ln2_128_hi, x);
    r = fma_(-kd, k.ln2_128_lo, r);
    const int ki = sat_int(kd);
    const double s = exp2_tab(ki & 127);
    double p = k.ee[0];
    for (int i = 1; i < 4; ++i) p = fma_(p, r, k.ee[i]);
    return ldexp_(fma_(s, p * r, s), ki >> 7);
#endif
}
CMX_LEAN_FN Log2Parts log2_reduce_m1(double x) {          // log2_reduce with the literal −1 (an inline constant of the hardware)
    const int32_t hw = hi_word(x);
    const int32_t tmp = hw - (int32_t)CMX_LEAN_LOG2_OFF_HI;
    const int32_t kk = tmp >> 20;
    const double z = from_words(hw - (int32_t)((uint32_t)tmp & 0xFFF00000u), lo_word(x));
    const Log2Entry e = log2_tab((tmp >> 13) & 127);
    return {(double)kk + e.logc, fma_(z, e.invc, -1.0)};
}
CMX_LEAN_FN double log(double x, const TabCoefs &k) {
#if CMX_LEAN_RESCUE_REPEAT
    const Log2Parts q = log2_reduce_m1(x);
    double y = fma_(q.hi, k.ln2, q.r * poly6(q.r, k.ln));
    if (log_needs_rescue(x)) {
        const Log2Parts s = log2_reduce_m1(x * k.two64);
        y = fma_(s.hi - k.sixty4, k.ln2, s.r * poly6(s.r, k.ln));
        y = x != x ? x : log_special(x, y);
    }
    return y;
#else
    const bool rare = log_needs_rescue(x);
    if (rare) x = log_prescale(x, k.two64);
    const Log2Parts q = log2_reduce_m1(x);
    const double rp = q.r * poly6(q.r, k.ln);
    double y = fma_(q.hi, k.ln2, rp);
    if (rare) y = log_postfix(x, fma_(q.hi - k.sixty4, k.ln2, rp));
    return y;
#endif
}
CMX_LEAN_FN double log_pos(double x, const TabCoefs &k) {                      // positive normal finite x (see log_pos above)
#if !CMX_F64_FINITE_FORMS
    return log(x, k);
#else
    const Log2Parts q = log2_reduce_m1(x);
    return fma_(q.hi, k.ln2, q.r * poly6(q.r, k.ln));
#endif
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
// √x, 1/√x and x^(−¾) for a POSITIVE FINITE argument (or NaN, which propagates): no 0 / Inf fix-ups (round 3, see exp2_fin)
CMX_LEAN_FN double rsqrt_pos(double x) { return rsqrt_core(x); }
CMX_LEAN_FN double sqrt_pos(double x) {
    const double y = rsqrt_core(x);
    const double s = x * y;
    return fma_(fma_(-s, s, x), 0.5 * y, s);
}
// x^(−¾) = u³ with u = x^(−¼): seed rsq(sqrt(x)) from the two hardware instructions (≈ 2⁻²³ relative), two Newton steps
// u ← u + ¼u(1 − x u⁴) (error 2.5 e² per step: 2⁻⁴³, 2⁻⁸⁵) — 14 instructions against 22 for rsqrt + sqrt + product
CMX_LEAN_FN double pow_m34_pos(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    double u = __builtin_amdgcn_rsq(__builtin_amdgcn_sqrt(x));
#else
    int e; const double m = std::frexp(std::pow(x, -0.25), &e);
    double u = std::ldexp((double)(float)m, e);          // host stand-in for the seed (tests only): 24 bits
#endif
    double u2 = u * u;
    u = fma_(0.25 * u, fma_(-x, u2 * u2, 1.0), u);
    u2 = u * u;
    u = fma_(0.25 * u, fma_(-x, u2 * u2, 1.0), u);
    return u * (u * u);
}
CMX_LEAN_FN double rsqrt(double x) {
    const double inf = std::numeric_limits<double>::infinity();
    const double y = rsqrt_core(x);
    return x == 0.0 ? inf : (x == inf ? 0.0 : y);
}

}  // namespace lean
}  // namespace cmx
