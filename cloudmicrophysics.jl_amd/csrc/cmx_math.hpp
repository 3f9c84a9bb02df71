// cmx_math.hpp — per-lane special-function layer for gfx950 (CDNA4).
//
// The rate functions are pointwise and HBM-bound only if the transcendental work
// stays on the hardware quarter-rate units: every pow/cbrt/exp/log of the
// reference is rewritten in the log2 domain (x^y = exp2(y·log2 x)) so that the
// Float32 path issues bare v_log_f32 / v_exp_f32 / v_rcp_f32 / v_sqrt_f32 and
// shares one log2 between all powers of the same base.  Float64 has no hardware
// transcendental unit: it uses the lean routines of cmx_lean_f64.hpp (20–35 VALU
// instructions, ≤ 4 ulp) instead of OCML's 50–150-instruction ones.
#pragma once
// CMX_HOST_BUILD: the point functions compiled by g++ for the HOST (tests/native/point_host.cpp) so that their algebra can be checked
// against the oracle without a GPU.  Test infrastructure only — libcmx.so is never built this way and has no CPU path.
#if defined(CMX_HOST_BUILD)
#include <cmath>
#define __device__
#define __host__
#define __forceinline__ inline
#else
#include <hip/hip_runtime.h>
#endif

#include <cmath>
#include <cstdint>
#include <type_traits>

#include "cmx_lean_f64.hpp"

namespace cmx {

template <typename FT> struct Math;

// the hardware transcendental / median instructions (host build: libm stand-ins of the same functions)
namespace hw {
#if defined(CMX_HOST_BUILD)
inline float exp2(float x) { return std::exp2(x); }
inline float log2(float x) { return std::log2(x); }
inline float rcp(float x) { return 1.0f / x; }
inline float sqrt(float x) { return std::sqrt(x); }
inline float rsq(float x) { return 1.0f / std::sqrt(x); }
inline float med3(float a, float b, float c) { return std::fmax(std::fmin(a, b), std::fmin(std::fmax(a, b), c)); }
#else
__device__ __forceinline__ float exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float log2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float med3(float a, float b, float c) { return __builtin_amdgcn_fmed3f(a, b, c); }
#endif
}  // namespace hw

template <> struct Math<float> {
    using Scalar = float;
    using Mask = bool;                     // the type of a comparison of two values (a lane mask pair for the packed type below)
    static constexpr bool IS_F64 = false;
    static constexpr int VEC = 4;
    static constexpr float eps() { return 1.1920928955078125e-07f; }          // eps(Float32)
    static constexpr float eps_1m() { return 2.2737367544323206e-13f; }       // cbrt(floatmin(Float32))
    static __device__ __forceinline__ void prepare() {}                       // Float32 runs on the hardware transcendental unit: nothing to set up
    static __device__ __forceinline__ float exp2(float x) { return hw::exp2(x); }
    static __device__ __forceinline__ float log2(float x) { return hw::log2(x); }
    static __device__ __forceinline__ float log2_pn(float x) { return hw::log2(x); }
    static __device__ __forceinline__ float rcp(float x) { return hw::rcp(x); }
    // the finite-argument forms of the Float64 side (below): the hardware instructions handle every special value at no cost
    static __device__ __forceinline__ float exp2_fin(float x) { return hw::exp2(x); }
    static __device__ __forceinline__ float rcp_nz(float x) { return hw::rcp(x); }
    static __device__ __forceinline__ float rcp_nz1(float x) { return hw::rcp(x); }
    static __device__ __forceinline__ float sqrt(float x) { return hw::sqrt(x); }
    static __device__ __forceinline__ float rsqrt(float x) { return hw::rsq(x); }
    static __device__ __forceinline__ float sqrt_pos(float x) { return hw::sqrt(x); }
    static __device__ __forceinline__ float rsqrt_pos(float x) { return hw::rsq(x); }
    static __device__ __forceinline__ float div(float a, float b) { return a * rcp(b); }
    static __device__ __forceinline__ float fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
    static __device__ __forceinline__ float max(float a, float b) { return __builtin_fmaxf(a, b); }
    static __device__ __forceinline__ float nan() { return __builtin_nanf(""); }
    static __device__ __forceinline__ float min(float a, float b) { return __builtin_fminf(a, b); }
    // log1p / expm1 accurate near 0 without the OCML double-float expansions (≈250 instructions each):
    // 4-term series below |x| = 1/32, the hardware log2/exp2 above (where 1+x / eˣ−1 no longer cancel: ≤4e-6 rel.)
    static __device__ __forceinline__ float log1p(float x) {
        const float s = x * fma(x, fma(x, fma(x, -0.25f, 1.0f / 3.0f), -0.5f), 1.0f);
        const float g = log2(1.0f + x) * 0.6931471805599453f;
        return __builtin_fabsf(x) < 0.03125f ? s : g;
    }
    static __device__ __forceinline__ float expm1(float x) {
        const float s = x * fma(x, fma(x, fma(x, 1.0f / 24.0f, 1.0f / 6.0f), 0.5f), 1.0f);
        const float g = exp2(x * 1.4426950408889634f) - 1.0f;
        return __builtin_fabsf(x) < 0.03125f ? s : g;
    }
};

template <> struct Math<double> {
    using Scalar = double;
    using Mask = bool;
    static constexpr bool IS_F64 = true;
    static constexpr int VEC = 2;
    static constexpr double eps() { return 2.220446049250313e-16; }            // eps(Float64)
    static constexpr double eps_1m() { return 2.8126442852362996e-103; }      // cbrt(floatmin(Float64))
    // Once per kernel that evaluates Float64 functions, before the first of them and executed by EVERY thread of the workgroup (the
    // streaming kernels issue their column loads first): copies the exp2 / log2 tables of cmx_lean_f64.hpp into LDS (3 KiB) and
    // synchronises
    static __device__ __forceinline__ void prepare() { lean::tables_init(); }
    static __device__ __forceinline__ double exp2(double x) { return lean::exp2(x); }
    static __device__ __forceinline__ double log2(double x) { return lean::log2(x); }
    static __device__ __forceinline__ double log2_pn(double x) { return lean::log2_pos(x); }   // positive NORMAL finite argument (cmx_lean_f64.hpp)
    static __device__ __forceinline__ double rcp(double x) { return lean::rcp(x); }
    // exp2_fin: the argument is finite or NaN (never ±Inf); rcp_nz: the argument is finite and non-zero, or NaN.  They drop the clamp /
    // NaN select (5 instructions) resp. the second Newton step and the 0 / Inf fix-up (5 instructions; 2⁻⁴⁸ relative instead of ≤ 1 ulp)
    // of the full forms; NaN still propagates.  Every call site states why its argument qualifies; CMX_F64_FINITE_FORMS=0
    // (cmx_lean_f64.hpp) maps them back to the full forms for A/B runs.
    static __device__ __forceinline__ double exp2_fin(double x) { return lean::exp2_fin(x); }
    static __device__ __forceinline__ double rcp_nz(double x) { return CMX_F64_FINITE_FORMS ? lean::rcp_nz(x) : lean::rcp(x); }
    // rcp_nz1: the same contract at ≤ 1 ulp (two Newton steps, no 0 / Inf fix-up) — where the reciprocal is amplified afterwards, e.g.
    // the determinants of the implicit step, whose result enters a difference quotient (q_new − q)/Δt
    static __device__ __forceinline__ double rcp_nz1(double x) { return CMX_F64_FINITE_FORMS ? lean::rcp_finite(x) : lean::rcp(x); }
    static __device__ __forceinline__ double sqrt(double x) { return lean::sqrt(x); }
    static __device__ __forceinline__ double rsqrt(double x) { return lean::rsqrt(x); }
    // positive finite argument (or NaN): no 0 / Inf fix-ups
    static __device__ __forceinline__ double sqrt_pos(double x) { return CMX_F64_FINITE_FORMS ? lean::sqrt_pos(x) : lean::sqrt(x); }
    static __device__ __forceinline__ double rsqrt_pos(double x) { return CMX_F64_FINITE_FORMS ? lean::rsqrt_pos(x) : lean::rsqrt(x); }
    static __device__ __forceinline__ double div(double a, double b) { return a * lean::rcp(b); }
    static __device__ __forceinline__ double fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
    static __device__ __forceinline__ double max(double a, double b) { return __builtin_fmax(a, b); }
    static __device__ __forceinline__ double nan() { return __builtin_nan(""); }
    static __device__ __forceinline__ double min(double a, double b) { return __builtin_fmin(a, b); }
    static __device__ __forceinline__ double log1p(double x) { return lean::log1p(x); }
    static __device__ __forceinline__ double expm1(double x) { return lean::expm1(x); }
};

// ---- two Float32 points per value: the PACKED value type (round 5) ------------------------------------------------------------------------
// gfx950 issues v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 — two results — in 4.2–4.8 cycles whatever their operands are, where the one-result forms cost
// 2.4–2.9 cycles only in their cheapest shape and 4.2–4.4 with a scalar (SGPR) operand or three distinct registers (tools/valu_probe, profiles/r05_probe_valu.txt:
// v_pk_mul v,s 4.80 against v_mul s,v 4.41; v_pk_fma v,s,v 4.66 against v_fma s,v,v 4.41; v_pk_fma of three registers 5.48 against 4.17; v_pk_add 4.39 against
// 2.58).  A third of the arithmetic of the Float32 rate kernels multiplies by a kernel constant, i.e. reads an SGPR.  Round 2 tried the compiler's SLP vectorizer
// and lost (it has to MOVE values into adjacent registers first); here the pairing is in the data flow from the start: a lane's 16-byte column vector arrives as
// two register pairs, the point functions are instantiated on the pair type, every multiply / add / fma of the source becomes one packed instruction (a scalar
// constant is broadcast by op_sel, no move), and the transcendentals, compares, selects and min / max — which have no packed Float32 form — run per half in
// place.  Same IEEE operations in the same order: results are bit-identical to the one-point instantiation (tests/test_sb2006_gpu.py, unaligned columns).
#if defined(__clang__)        // ext_vector_type: hipcc and clang++ (the host test build prefers clang++; under g++ the packed type does not exist)
#define CMX_HAVE_PACKED 1
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
#define CMX_LANEWISE1(fn) f32x2{fn(x.x), fn(x.y)}
template <> struct Math<f32x2> {
    using Scalar = float;
    using Mask = i32x2;                    // lane-wise comparison results (−1 / 0); combine with | and &, select with ?:
    static constexpr bool IS_F64 = false;
    static constexpr float eps() { return Math<float>::eps(); }
    static constexpr float eps_1m() { return Math<float>::eps_1m(); }
    static __device__ __forceinline__ void prepare() {}
    static __device__ __forceinline__ f32x2 exp2(f32x2 x) { return CMX_LANEWISE1(hw::exp2); }
    static __device__ __forceinline__ f32x2 log2(f32x2 x) { return CMX_LANEWISE1(hw::log2); }
    static __device__ __forceinline__ f32x2 log2_pn(f32x2 x) { return CMX_LANEWISE1(hw::log2); }
    static __device__ __forceinline__ f32x2 rcp(f32x2 x) { return CMX_LANEWISE1(hw::rcp); }
    static __device__ __forceinline__ f32x2 exp2_fin(f32x2 x) { return CMX_LANEWISE1(hw::exp2); }
    static __device__ __forceinline__ f32x2 rcp_nz(f32x2 x) { return CMX_LANEWISE1(hw::rcp); }
    static __device__ __forceinline__ f32x2 rcp_nz1(f32x2 x) { return CMX_LANEWISE1(hw::rcp); }
    static __device__ __forceinline__ f32x2 sqrt(f32x2 x) { return CMX_LANEWISE1(hw::sqrt); }
    static __device__ __forceinline__ f32x2 rsqrt(f32x2 x) { return CMX_LANEWISE1(hw::rsq); }
    static __device__ __forceinline__ f32x2 sqrt_pos(f32x2 x) { return CMX_LANEWISE1(hw::sqrt); }
    static __device__ __forceinline__ f32x2 rsqrt_pos(f32x2 x) { return CMX_LANEWISE1(hw::rsq); }
    static __device__ __forceinline__ f32x2 log1p(f32x2 x) { return CMX_LANEWISE1(Math<float>::log1p); }
    static __device__ __forceinline__ f32x2 expm1(f32x2 x) { return CMX_LANEWISE1(Math<float>::expm1); }
    // any mix of pair and scalar arguments (a scalar is broadcast: (f32x2)s)
    template <typename A, typename B> static __device__ __forceinline__ f32x2 div(A a, B b) { return (f32x2)a * rcp((f32x2)b); }
    template <typename A, typename B, typename C> static __device__ __forceinline__ f32x2 fma(A a, B b, C c) {
#if defined(CMX_HOST_BUILD)
        const f32x2 u = (f32x2)a, v = (f32x2)b, w = (f32x2)c;
        return f32x2{__builtin_fmaf(u.x, v.x, w.x), __builtin_fmaf(u.y, v.y, w.y)};
#else
        return __builtin_elementwise_fma((f32x2)a, (f32x2)b, (f32x2)c);
#endif
    }
    template <typename A, typename B> static __device__ __forceinline__ f32x2 max(A a, B b) {
        const f32x2 u = (f32x2)a, v = (f32x2)b;
        return f32x2{__builtin_fmaxf(u.x, v.x), __builtin_fmaxf(u.y, v.y)};
    }
    template <typename A, typename B> static __device__ __forceinline__ f32x2 min(A a, B b) {
        const f32x2 u = (f32x2)a, v = (f32x2)b;
        return f32x2{__builtin_fminf(u.x, v.x), __builtin_fminf(u.y, v.y)};
    }
    static __device__ __forceinline__ f32x2 nan() { return (f32x2)__builtin_nanf(""); }
};
#undef CMX_LANEWISE1
#else
#define CMX_HAVE_PACKED 0
#endif
// m_or(a, b, …): "or" of comparison results of any value type — lane masks for the pair type; for bool the non-short-circuit form (both sides are plain
// compares: no branch wanted)
__device__ __forceinline__ bool m_or(bool a, bool b) { return (bool)((int)a | (int)b); }
#if CMX_HAVE_PACKED
__device__ __forceinline__ i32x2 m_or(i32x2 a, i32x2 b) { return a | b; }
#endif
template <typename B, typename... R> __device__ __forceinline__ B m_or(B a, B b, R... r) { return m_or(m_or(a, b), r...); }
__device__ __forceinline__ bool m_and(bool a, bool b) { return a && b; }
#if CMX_HAVE_PACKED
__device__ __forceinline__ i32x2 m_and(i32x2 a, i32x2 b) { return a & b; }
#endif
template <typename B, typename... R> __device__ __forceinline__ B m_and(B a, B b, R... r) { return m_and(m_and(a, b), r...); }
// CMX_F32_PACKED: the Float32 instantiations with four points per lane evaluate two PAIRS of points in packed arithmetic (cmx_math.hpp f32x2); 0 = one
// point at a time as in rounds 1–4 (A/B switch)
#ifndef CMX_F32_PACKED
#define CMX_F32_PACKED 1
#endif
// CMX_F32_PACKED_PHASE_CONSTS: a packed instruction takes no literal, so the ≈ 30 literal constants of the point function occupy SGPRs next to the 95 kernel
// constants — more than the file holds (84 v_writelane + 84 v_readlane per 4 points appeared).  The packed instantiations therefore read the constants
// through the kernel-argument pointer, phase by phase, like the Float64 kernels do (cmx_math.hpp consts_after).
#ifndef CMX_F32_PACKED_PHASE_CONSTS
#define CMX_F32_PACKED_PHASE_CONSTS 1
#endif
// the scalar type and the points per value of a value type
template <typename VT> struct lanes_of { static constexpr int value = 1; };
#if CMX_HAVE_PACKED
template <> struct lanes_of<f32x2> { static constexpr int value = 2; };
#endif

// Where the Float64 kernels' constants live (measured, round 2, one box, tools/ab_bench.sh): a Float64 kernel's 60–100 host-folded
// parameters overflow the 102-SGPR file and the overflow is parked in VGPR lanes (v_writelane / v_readlane).  Staging the parameter
// struct in LDS instead (uniform-address ds_read_b64) removed every readlane — and was SLOWER: the compiler hoists all those loads
// to the top of the kernel, 243 VGPRs, 2 waves per SIMD, and the dependent Float64 chains are no longer covered (SB2006 3.23 → 3.68 ms,
// 1-moment 4.60 → 5.72, ARG 3.23 → 3.45, ice nucleation 0.72 → 0.82); capping the registers turns the hoisted loads into scratch
// spills (14.6 ms).  The parameters therefore stay kernel arguments for both float types.

// `keep(x)`: x is computed HERE, unconditionally.  The gates of the rate functions are selects (`gate ? 0 : rate`), exactly the
// reference's `ifelse`; when `rate` ends in a transcendental the compiler prefers a branch around it (v_exp / v_log are "expensive"
// in its cost model) — per point, 3–4 branches in the SB2006 kernel.  A lane owns 4 independent points whose instruction streams
// should interleave; every branch ends a basic block, costs 4–6 scalar instructions and a pipeline bubble, and in a wave of 64
// different states both sides run anyway.  The empty asm makes the value opaque at this point, so it cannot be sunk behind the gate.
#ifndef CMX_KEEP
#define CMX_KEEP 1
#endif
template <typename FT> __device__ __forceinline__ FT keep(FT x) {
#if CMX_KEEP && !defined(CMX_HOST_BUILD)
    asm volatile("" : "+v"(x));
#endif
    return x;
}

// ---- Γ of the Chen-2022 rain exponents as polynomials in the air density (round 3) ----------------------------------------------------
// The three exponents of a Chen-2022 rain table depend on the state only through the air density, b_i(ρ) = b_i − b_ρ ρ (Common.jl:290-302),
// and over 0 ≤ ρ ≤ 2 kg/m³ each moves by b_ρ·2 ≈ 0.08: Γ(b_i(ρ) + 1) is a short polynomial in ρ there, fitted on the HOST for the
// parameter set at hand (Chebyshev interpolation in t = ρ − 1 ∈ [−1, 1], degree 3 for Float32 / 8 for Float64, converted to monomials;
// the nearest singularity of Γ is two units away, so the error falls by ≈ 100× per degree).  Any parameter set with positive
// arguments is covered — no fixed polynomial window — and a Γ costs 3 (8) fma instead of 13 (24).  make_chen_gamma() returns false if
// the fit misses its accuracy at the probe points (a pole inside the range: b_i + 1 ≤ 2 b_ρ, or a huge b_ρ); the entry points then
// take the GENERAL instantiation (tgamma_general below).  Beyond ρ = kChenGammaRhoMax the fast instantiations return NaN fall speeds.
constexpr double kChenGammaRhoMax = 2.0;
template <typename FT> struct ChenGamma {
    static constexpr int D = sizeof(FT) == 4 ? 3 : 8;
    FT c[3][D + 1];      // Γ(b_i − b_ρ ρ + 1 [+ 3])[/3!] = Σ_m c[i][m] (ρ − 1)^m
};
// KSHIFT = 0: Γ(b + 1) (number-weighted, k = 0);  KSHIFT = 3: Γ(b + 4)/3! (mass-weighted, k = 3 — Chen2022_exponential_pdf, Common.jl:414-422)
template <typename FT, typename CH> inline bool make_chen_gamma(const CH &ch, ChenGamma<FT> &g, int kshift = 0) {
    constexpr int D = ChenGamma<FT>::D, N = D + 1;
    const double pi = 3.14159265358979323846, half = 0.5 * kChenGammaRhoMax;
    bool ok = true;
    for (int i = 0; i < 3; ++i) {
        const double kfact = kshift == 3 ? 6.0 : 1.0;
        auto f = [&](double t) { return std::tgamma((double)ch.b[i] - (double)ch.b_rho * (half + half * t) + 1.0 + kshift) / kfact; };
        if (!((double)ch.b[i] + 1.0 - (double)ch.b_rho * kChenGammaRhoMax > 0.05) || !((double)ch.b[i] + 1.0 < 30.0)) ok = false;
        // Chebyshev coefficients from the N Chebyshev nodes, then Σ a_k T_k(t) → monomials by the recurrence T_{k+1} = 2 t T_k − T_{k−1}
        double a[N], fx[N], mono[N] = {0}, Tkm1[N] = {0}, Tk[N] = {0};
        for (int j = 0; j < N; ++j) fx[j] = f(std::cos(pi * (j + 0.5) / N));
        for (int k = 0; k < N; ++k) {
            double sum = 0;
            for (int j = 0; j < N; ++j) sum += fx[j] * std::cos(pi * k * (j + 0.5) / N);
            a[k] = (k == 0 ? 1.0 : 2.0) / N * sum;
        }
        Tkm1[0] = 1.0;                        // T_0
        for (int m = 0; m < N; ++m) mono[m] += a[0] * Tkm1[m];
        if (N > 1) {
            Tk[1] = 1.0;                      // T_1
            for (int m = 0; m < N; ++m) mono[m] += a[1] * Tk[m];
            for (int k = 2; k < N; ++k) {
                double Tn[N] = {0};
                for (int m = 0; m < N; ++m) Tn[m] = (m > 0 ? 2.0 * Tk[m - 1] : 0.0) - Tkm1[m];
                for (int m = 0; m < N; ++m) { mono[m] += a[k] * Tn[m]; Tkm1[m] = Tk[m]; Tk[m] = Tn[m]; }
            }
        }
        // t = (ρ − half)/half: fold 1/half^m into the coefficients so that the kernel evaluates in (ρ − half) directly
        double scale = 1.0;
        for (int m = 0; m < N; ++m) { g.c[i][m] = (FT)(mono[m] * scale); scale /= half; }
        for (double t : {-0.97, -0.41, 0.13, 0.58, 0.99}) {      // off-node probes
            double p = 0;
            for (int m = N - 1; m >= 0; --m) p = p * t + mono[m];
            if (!(std::fabs(p - f(t)) <= (sizeof(FT) == 4 ? 2e-7 : 4e-15) * std::fabs(f(t)))) ok = false;
        }
    }
    return ok;
}
// ---- round 4: the whole density-dependent coefficient of a Chen-2022 rain term, in the log2 domain --------------------------------------
// Term i of the number- (k = 0) and mass-weighted (k = 3) rain fall speed is (Common.jl:290-302,414-422, CM2:703-719)
//     a_i e^{ρ0 ρ} 1000^{b_i(ρ)} [ρ^{a3_pow}, i = 3] · Γ(b_i(ρ) + k + 1)/k! · λ^{k+1} (λ + 1000 c_i)^{−(b_i(ρ) + k + 1)},   b_i(ρ) = b_i − b_ρ ρ.
// Everything in front of the λ powers depends on the state only through ρ, and its logarithm
//     L_{k,i}(ρ) = log2|a_i| + ρ0 log2(e) ρ + b_i(ρ) log2 1000 + log2 Γ(b_i(ρ) + k + 1) − log2 k!
// is linear in ρ plus log2 Γ of an argument that moves by 2 b_ρ ≈ 0.08 over 0 ≤ ρ ≤ 2 kg/m³ — a polynomial of degree 3 (Float32) / 7
// (Float64) in ρ − 1 to rounding.  The kernel then forms each term as ONE exponential of  L_{k,i}(ρ) + (k+1) log2 λ − (b_i(ρ)+k+1) log2(λ + 1000 c_i)
// instead of a Γ polynomial, a magnitude and an exponential each (≈ 17 instead of 26 instructions per term; PMC round 4: 334 → … per point).
// make_chen_log() returns false where the fit misses its accuracy (a pole of Γ inside the range, a huge b_ρ): the GENERAL instantiation runs.
template <typename FT> struct ChenLog {
    static constexpr int D = sizeof(FT) == 4 ? 3 : 7;
    FT c[2][3][D + 1];    // [k = 0 | 3][term][monomial in (ρ − 1)]
    FT sgn[3];            // sign of a_i (0 for a_i = 0)
    FT b1[3], b4[3];      // b_i + 1, b_i + 4
};
template <typename FT, typename CH> inline bool make_chen_log(const CH &ch, ChenLog<FT> &g) {
    constexpr int D = ChenLog<FT>::D, N = D + 1;
    const double pi = 3.14159265358979323846, half = 0.5 * kChenGammaRhoMax, l2e = 1.4426950408889634074, ln2 = 0.69314718055994530942;
    bool ok = true;
    for (int i = 0; i < 3; ++i) {
        const double ai = (double)ch.a[i];
        g.sgn[i] = (FT)(ai > 0 ? 1.0 : (ai < 0 ? -1.0 : 0.0));
        g.b1[i] = (FT)((double)ch.b[i] + 1.0);
        g.b4[i] = (FT)((double)ch.b[i] + 4.0);
        if (!((double)ch.b[i] + 1.0 - (double)ch.b_rho * kChenGammaRhoMax > 0.05) || !((double)ch.b[i] + 4.0 < 30.0)) ok = false;
        for (int kk = 0; kk < 2; ++kk) {
            const int k = kk == 0 ? 0 : 3;
            auto f = [&](double t) {
                const double rho = half + half * t, b = (double)ch.b[i] - (double)ch.b_rho * rho;
                return (ai != 0 ? std::log2(std::fabs(ai)) : -1000.0) + (double)ch.rho_0 * l2e * rho + b * std::log2(1000.0) +
                       (std::lgamma(b + k + 1.0) - (k == 3 ? std::log(6.0) : 0.0)) / ln2;
            };
            double a[N], fx[N], mono[N] = {0}, Tkm1[N] = {0}, Tk[N] = {0};
            for (int j = 0; j < N; ++j) fx[j] = f(std::cos(pi * (j + 0.5) / N));
            for (int q = 0; q < N; ++q) {
                double sum = 0;
                for (int j = 0; j < N; ++j) sum += fx[j] * std::cos(pi * q * (j + 0.5) / N);
                a[q] = (q == 0 ? 1.0 : 2.0) / N * sum;
            }
            Tkm1[0] = 1.0;
            for (int m = 0; m < N; ++m) mono[m] += a[0] * Tkm1[m];
            if (N > 1) {
                Tk[1] = 1.0;
                for (int m = 0; m < N; ++m) mono[m] += a[1] * Tk[m];
                for (int q = 2; q < N; ++q) {
                    double Tn[N] = {0};
                    for (int m = 0; m < N; ++m) Tn[m] = (m > 0 ? 2.0 * Tk[m - 1] : 0.0) - Tkm1[m];
                    for (int m = 0; m < N; ++m) { mono[m] += a[q] * Tn[m]; Tkm1[m] = Tk[m]; Tk[m] = Tn[m]; }
                }
            }
            double scale = 1.0;
            for (int m = 0; m < N; ++m) { g.c[kk][i][m] = (FT)(mono[m] * scale); scale /= half; }
            for (double t : {-0.97, -0.41, 0.13, 0.58, 0.99}) {      // off-node probes: ABSOLUTE error of the logarithm = relative error of the coefficient
                double p = 0;
                for (int m = N - 1; m >= 0; --m) p = p * t + mono[m];
                // Float64: L is of size 30 — the double-precision evaluation of the probe itself carries ≈ 30·2⁻⁵³ ≈ 3e-15 of rounding, and so does
                // the exponent the kernel forms from it; 2e-14 separates that from a fit that misses (a pole nearby: errors ≥ 1e-9)
                if (!(std::fabs(p - f(t)) <= (sizeof(FT) == 4 ? 2e-7 / ln2 : 2e-14))) ok = false;
            }
        }
    }
    return ok;
}
// (FT: the SCALAR type of the table; the argument may be the packed pair type)
template <typename FT, typename G, typename VT> __device__ __forceinline__ VT chen_log_eval(const G &g, int kk, int i, VT t) {
    VT p = VT(g.c[kk][i][ChenLog<FT>::D]);
#pragma unroll
    for (int m = ChenLog<FT>::D - 1; m >= 0; --m) p = Math<VT>::fma(p, t, g.c[kk][i][m]);
    return p;
}

// (G: ChenGamma<FT>, possibly in the constant address space — the Float64 kernels read their constants through the kernel-argument pointer)
// (FT: the SCALAR type of the table; the argument may be the packed pair type)
template <typename FT, typename G, typename VT> __device__ __forceinline__ VT chen_gamma_eval(const G &g, int i, VT rho_c) {
    const VT t = rho_c - VT(0.5 * kChenGammaRhoMax);
    VT p = VT(g.c[i][ChenGamma<FT>::D]);
#pragma unroll
    for (int m = ChenGamma<FT>::D - 1; m >= 0; --m) p = Math<VT>::fma(p, t, g.c[i][m]);
    return p;
}

// Γ(z) for ANY argument (the reference evaluates SF.gamma at run time, Common.jl:414-422): OCML's tgamma — ≈ 110 (Float32) / 300
// (Float64) instructions, used only by the GENERAL instantiations that parameter sets outside the polynomial window above select
template <typename FT> __device__ __forceinline__ FT tgamma_general(FT z) {
#if defined(CMX_HOST_BUILD)
    return (FT)std::tgamma((double)z);
#else
    if constexpr (sizeof(FT) == 8) return ::tgamma(z);
    else return ::tgammaf(z);
#endif
}
#if CMX_HAVE_PACKED
template <> __device__ __forceinline__ f32x2 tgamma_general<f32x2>(f32x2 z) { return f32x2{tgamma_general<float>(z.x), tgamma_general<float>(z.y)}; }
#endif
template <typename FT> __device__ __forceinline__ FT clampv(FT x, FT lo, FT hi) {
    // Base.clamp: x < lo ? lo : (x > hi ? hi : x)
    return Math<FT>::min(Math<FT>::max(x, lo), hi);
}
// clamp_to_nonneg of a LOADED value: max(0, x) through v_med3_f32(x, 0, +Inf).  A plain fmax on a value that comes straight from
// memory costs two instructions in IEEE mode (the compiler must quiet a possible signalling NaN first: v_max x, x); the median does
// not need that, returns 0 for a NaN like v_max does (the NaN rule is applied separately), and its result counts as canonical.
// (Written as an inline-asm v_med3_f32 so that the optimiser cannot turn it back into canonicalize + v_max — it does in the 1-moment
// kernels — it measured no faster there and 6 % slower in the LinearizedAverage kernel, round 2: the builtin stays.)
// Round 3 (CMX_MAX0_INT, default on): as ONE integer instruction instead — for a float, max(0, x) is v_max_i32 on the bits (negative
// floats, −0 and sign-bit NaNs are negative integers → +0; positive floats, +Inf and the other NaNs pass).  No canonicalisation is
// emitted for an integer operation, and a NaN input is poisoned by the kernels' own rule either way.  Same-box A/B (3 interleaved
// runs, ms per 1e8 Float32 points): 1-moment tendencies 0.788–0.804 → 0.782–0.793, LinearizedAverage 1.79 → 1.76, SB2006 unchanged.
#ifndef CMX_MAX0_INT
#define CMX_MAX0_INT 1
#endif
__device__ __forceinline__ float max0(float x) {
#if CMX_MAX0_INT && !defined(CMX_HOST_BUILD)
    const int i = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, i > 0 ? i : 0);
#else
    return hw::med3(x, 0.0f, __builtin_inff());
#endif
}
__device__ __forceinline__ double max0(double x) { return Math<double>::max(0.0, x); }
#if CMX_HAVE_PACKED
__device__ __forceinline__ f32x2 max0(f32x2 x) { return f32x2{max0(x.x), max0(x.y)}; }
#endif
// log2 of a non-negative quantity that is ORDINARILY zero — ρ·q of an absent species.  Float64: floored at the smallest normal number (−1022
// instead of −Inf: every caller floors the slope parameter it derives far above that, or gates the result on q > ϵ) and evaluated by the
// main path of the logarithm alone; the full form would send nearly every wave of a real field through its rescue block.  A NaN argument does
// not survive the max: the kernels apply their NaN rule to the inputs.  Float32: the hardware logarithm takes 0 as it is.
template <typename FT> __device__ __forceinline__ FT log2_floored(FT x) {
    using M = Math<FT>;
    if constexpr (M::IS_F64) return M::log2_pn(M::max(x, FT(2.2250738585072014e-308)));
    else return M::log2(x);
}
// the same for lo ≤ hi as ONE instruction (v_med3_f32: the median of three is the clamp); Float64 has no med3
__device__ __forceinline__ float clamp_ordered(float x, float lo, float hi) { return hw::med3(x, lo, hi); }
__device__ __forceinline__ double clamp_ordered(double x, double lo, double hi) { return clampv(x, lo, hi); }
#if CMX_HAVE_PACKED
__device__ __forceinline__ f32x2 clamp_ordered(f32x2 x, float lo, float hi) { return f32x2{hw::med3(x.x, lo, hi), hw::med3(x.y, lo, hi)}; }
__device__ __forceinline__ f32x2 clamp_ordered(f32x2 x, f32x2 lo, f32x2 hi) { return f32x2{hw::med3(x.x, lo.x, hi.x), hw::med3(x.y, lo.y, hi.y)}; }
#endif

// ---- phase-local constants (Float64) --------------------------------------------------------------------------------------------
// The constants struct of a Float64 kernel is large: SbConsts<double> is ≈ 95 doubles = 190 SGPRs; the register file has 102.
// Left alone the compiler loads the whole kernel-argument struct in the entry block and spills it to VGPR lanes: the Float64 SB2006
// tendencies kernel carried 72 v_writelane + 136 v_readlane per point (17 % of its VALU instructions — and the Float64 kernels are
// VALU-issue-bound), the 1-moment kernel 148 + 340 of 1656.  The Float64 kernels therefore read the constants through a
// constant-address-space pointer to the kernel-argument segment (front_consts) and pass it through consts_after(c, x) at the phase
// boundaries of the point function: an empty asm that makes the pointer depend on a value the previous phase computed, so the
// scalar loads of a phase cannot be hoisted above it and only that phase's constants are live.  Float32 (95 SGPRs) needs none of it.
#ifndef CMX_PHASE_CONSTS
#define CMX_PHASE_CONSTS 1
#endif
// CMX_PHASE_DEP(late, early): the value a phase's constants wait for — the LAST value of the previous phase (only one phase's
// constants live), or with CMX_PHASE_PREFETCH an EARLY value of the previous phase (the loads overlap that phase's arithmetic; two
// phases' constants live).
#ifndef CMX_PHASE_PREFETCH
#define CMX_PHASE_PREFETCH 0
#endif
#if CMX_PHASE_PREFETCH
#define CMX_PHASE_DEP(late, early) early
#else
#define CMX_PHASE_DEP(late, early) late
#endif
#if defined(CMX_HOST_BUILD)
#undef CMX_PHASE_CONSTS
#define CMX_PHASE_CONSTS 0
template <typename T> using KernArg = const T;
template <typename T> struct is_kernarg { static constexpr bool value = false; };
#else
template <typename T> using KernArg = const __attribute__((address_space(4))) T;
template <typename T> struct is_kernarg { static constexpr bool value = false; };
template <typename T> struct is_kernarg<const __attribute__((address_space(4))) T> { static constexpr bool value = true; };
#endif
// the constants struct must be the FIRST kernel argument
// (ALSO: a Float32 kernel whose constants overflow the SGPR file too — the 1-moment kernels with run-time option flags)
#if defined(CMX_HOST_BUILD)
template <typename FT, bool ALSO = false, typename CT> inline const CT &front_consts(const CT &c) { return c; }
template <typename C, typename FT> inline const C &consts_after(const C &c, FT) { return c; }
#else
template <typename FT, bool ALSO = false, typename CT> __device__ __forceinline__ decltype(auto) front_consts(const CT &c) {
    if constexpr ((sizeof(FT) == 8 || ALSO) && CMX_PHASE_CONSTS) return (*(KernArg<CT> *)__builtin_amdgcn_kernarg_segment_ptr());
    else return (c);
}
// (a plain reference passes through: laundering the address of a by-value kernel argument would force a private copy of the struct)
template <typename C, typename FT> __device__ __forceinline__ const C &consts_after(const C &c, FT dep) {
    if constexpr (is_kernarg<const C>::value) {
        const C *p = &c;
        asm volatile("" : "+s"(p) : "v"(dep));
        return *p;
    } else
        return c;
}
#endif

// NaN inputs.  The reference sanitises with Julia's max(0, x), which returns NaN for a NaN x, and its arithmetic then carries the NaN to
// the tendencies; the hardware v_max returns the other operand and would hide it.  The bulk-tendency kernels therefore poison every output
// of a point whose inputs hold a NaN: one unordered compare per pair of inputs.
#ifndef CMX_NAN_POISON
#define CMX_NAN_POISON 1
#endif
template <typename FT> __device__ __forceinline__ bool any_nan(FT a) { return CMX_NAN_POISON && a != a; }
template <typename FT> __device__ __forceinline__ bool any_nan(FT a, FT b) { return CMX_NAN_POISON && __builtin_isunordered(a, b); }
template <typename FT, typename... R> __device__ __forceinline__ bool any_nan(FT a, FT b, R... r) {
    return (bool)((int)any_nan(a, b) | (int)any_nan(r...));   // bitwise on purpose: no branch
}
// The Float64 1-moment entries poison ρ ≤ 0 as well: the air density must be positive (include/cmx.h), their finite-argument forms assume it, and a
// point outside the domain gets NaN in EVERY output rather than a pattern of NaN, ±Inf and valid-looking numbers.  (Float32 reproduces the
// reference's own pattern there: its hardware functions take 0 and Inf as IEEE division does.)
template <typename FT> __device__ __forceinline__ bool bad_density(FT rho) {
    if constexpr (std::is_same<FT, double>::value) return !(rho > 0.0);
    else return false;
}
// pairs: a lane mask, formed per half with the same unordered compares
#if CMX_HAVE_PACKED
template <typename... R> __device__ __forceinline__ i32x2 any_nan_pair(f32x2 a, R... r) {
    return i32x2{any_nan(a.x, r.x...) ? -1 : 0, any_nan(a.y, r.y...) ? -1 : 0};
}
#endif
// nan_mask(inputs…): Math<VT>::Mask of "one of this point's inputs is a NaN" for any value type
template <typename VT, typename... R> __device__ __forceinline__ typename Math<VT>::Mask nan_mask(VT a, R... r) {
#if CMX_HAVE_PACKED
    if constexpr (lanes_of<VT>::value == 2) return any_nan_pair(a, r...);
    else
#endif
        return any_nan(a, r...);
}

}  // namespace cmx
