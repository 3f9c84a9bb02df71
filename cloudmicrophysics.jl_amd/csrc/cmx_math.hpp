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
    static constexpr int VEC = 4;
    static constexpr float eps() { return 1.1920928955078125e-07f; }          // eps(Float32)
    static constexpr float eps_1m() { return 2.2737367544323206e-13f; }       // cbrt(floatmin(Float32))
    static __device__ __forceinline__ void prepare() {}                       // Float32 runs on the hardware transcendental unit: nothing to set up
    static __device__ __forceinline__ float exp2(float x) { return hw::exp2(x); }
    static __device__ __forceinline__ float log2(float x) { return hw::log2(x); }
    static __device__ __forceinline__ float rcp(float x) { return hw::rcp(x); }
    static __device__ __forceinline__ float sqrt(float x) { return hw::sqrt(x); }
    static __device__ __forceinline__ float rsqrt(float x) { return hw::rsq(x); }
    static __device__ __forceinline__ float div(float a, float b) { return a * rcp(b); }
    static __device__ __forceinline__ float fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
    static __device__ __forceinline__ float max(float a, float b) { return __builtin_fmaxf(a, b); }
    static __device__ __forceinline__ float nan() { return __builtin_nanf(""); }
    static __device__ __forceinline__ float min(float a, float b) { return __builtin_fminf(a, b); }
    // Γ(z), used by the Chen-2022 rain velocity only, where z = b_i(ρ) + 1 ∈ [2.0, 3.4] (b = 1.15 / 2.30 − 0.038 ρ, Common.jl:290-302):
    // a degree-10 interpolating polynomial in t = (z − 2.75)/1.25 on [1.5, 4] (3.3e-7 relative in Float32 Horner; OCML's tgammaf
    // costs ≈ 110 instructions and made the Chen variant of the fused kernel compute-bound: 1.37 ms per 1e8 points in round 1).
    // The argument is clamped to [1.5, 4] (one v_med3): entry points that take Chen-2022 rain parameters check on the host that
    // b_i + 1 stays inside for every air density up to 2 kg/m³ (cmx::chen_rain_gamma_domain_ok) and return CMX_ERR_UNSUPPORTED otherwise.
    // (An OCML fallback behind a never-taken branch was tried first: inlined at 12 call sites it took the kernel to 179 VGPRs, 2 waves
    // per SIMD, 1.17 ms.)
    static __device__ __forceinline__ float tgamma(float z) {
        const float t = fma(hw::med3(z, 1.5f, 4.0f), 0.8f, -2.2f);
        float p = 0.000606461835549f;
        p = fma(p, t, 0.000949070487934f);
        p = fma(p, t, 0.00527334298015f);
        p = fma(p, t, 0.015461039999f);
        p = fma(p, t, 0.0556865757496f);
        p = fma(p, t, 0.142589333556f);
        p = fma(p, t, 0.380736919836f);
        p = fma(p, t, 0.751528528131f);
        p = fma(p, t, 1.39245065912f);
        p = fma(p, t, 1.64635862269f);
        return fma(p, t, 1.60835942199f);
    }
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
    static constexpr int VEC = 2;
    static constexpr double eps() { return 2.220446049250313e-16; }            // eps(Float64)
    static constexpr double eps_1m() { return 2.8126442852362996e-103; }      // cbrt(floatmin(Float64))
    // Once per kernel that evaluates Float64 functions, before the first of them and executed by EVERY thread of the workgroup (the
    // streaming kernels issue their column loads first): copies the exp2 / log2 tables of cmx_lean_f64.hpp into LDS (3 KiB) and
    // synchronises
    static __device__ __forceinline__ void prepare() { lean::tables_init(); }
    static __device__ __forceinline__ double exp2(double x) { return lean::exp2(x); }
    static __device__ __forceinline__ double log2(double x) { return lean::log2(x); }
    static __device__ __forceinline__ double rcp(double x) { return lean::rcp(x); }
    static __device__ __forceinline__ double sqrt(double x) { return lean::sqrt(x); }
    static __device__ __forceinline__ double rsqrt(double x) { return lean::rsqrt(x); }
    static __device__ __forceinline__ double div(double a, double b) { return a * lean::rcp(b); }
    static __device__ __forceinline__ double fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
    static __device__ __forceinline__ double max(double a, double b) { return __builtin_fmax(a, b); }
    static __device__ __forceinline__ double nan() { return __builtin_nan(""); }
    static __device__ __forceinline__ double min(double a, double b) { return __builtin_fmin(a, b); }
    // Γ(z) on [1.5, 4] (see the Float32 twin; argument clamped): degree-22 interpolating polynomial, 9e-15 relative
    static __device__ __forceinline__ double tgamma(double z) {
        const double t = fma(min(max(z, 1.5), 4.0), 0.8, -2.2);
        const double k[23] = {1.6083594219855455584, 1.6463589739909790518, 1.3924499187912875802, 0.75152171904830877759,
                              0.38075160368679774603, 0.14262439063517987378, 0.055606088443515184361, 0.015394396819926214053,
                              0.0054496056318177296227, 0.00098897444374469870313, 0.00044573752011233748607, 0.000010932744047535149967,
                              0.000043139914396103166238, -9.7546600621484357892e-6, 6.4693602774412672403e-6, -2.243298262749311333e-6,
                              1.0854823652885385946e-6, -8.4496532802657781764e-7, 3.894405396736869375e-7, 7.7776820822669398175e-8,
                              -3.6483865756254320843e-8, -8.5261728750953385182e-8, 3.8953586756712331112e-8};
        double p = k[22];
#pragma unroll
        for (int i = 21; i >= 0; --i) p = fma(p, t, k[i]);
        return p;
    }
    static __device__ __forceinline__ double log1p(double x) { return lean::log1p(x); }
    static __device__ __forceinline__ double expm1(double x) { return lean::expm1(x); }
};

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

// Host check for the polynomial Γ above: the Chen-2022 rain exponents b_i(ρ) = b_i − b_ρ ρ (Common.jl:290-302) must keep
// z = b_i(ρ) + 1 inside [1.5, 4] for 0 ≤ ρ ≤ 2 kg/m³ (the reference's Table B1 values: z ∈ [2.07, 3.30]).
template <typename CH> inline bool chen_rain_gamma_domain_ok(const CH &ch) {
    if (!(ch.b_rho >= 0)) return false;
    for (int i = 0; i < 3; ++i)
        if (!((double)ch.b[i] + 1.0 <= 4.0 && (double)ch.b[i] + 1.0 - 2.0 * (double)ch.b_rho >= 1.5)) return false;
    return true;
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
// largest air density for which every z = b_i − b_ρ ρ + 1 of a Chen-2022 rain table stays inside the polynomial window [1.5, 4]
// (b_ρ ≥ 0: z falls with ρ).  The fast instantiations poison the fall speeds of points above it with NaN instead of clamping.
template <typename CH> inline double chen_rain_gamma_rho_max(const CH &ch) {
    double r = 1e300;
    for (int i = 0; i < 3; ++i)
        if ((double)ch.b_rho > 0) r = std::fmin(r, ((double)ch.b[i] + 1.0 - 1.5) / (double)ch.b_rho);
    return r;
}

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
// the same for lo ≤ hi as ONE instruction (v_med3_f32: the median of three is the clamp); Float64 has no med3
__device__ __forceinline__ float clamp_ordered(float x, float lo, float hi) { return hw::med3(x, lo, hi); }
__device__ __forceinline__ double clamp_ordered(double x, double lo, double hi) { return clampv(x, lo, hi); }

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
template <typename FT, size_t OFFSET, typename CT> inline const CT &kernarg_at(const CT &c) { return c; }
template <typename C, typename FT> inline const C &consts_after(const C &c, FT) { return c; }
#else
template <typename FT, bool ALSO = false, typename CT> __device__ __forceinline__ decltype(auto) front_consts(const CT &c) {
    if constexpr ((sizeof(FT) == 8 || ALSO) && CMX_PHASE_CONSTS) return (*(KernArg<CT> *)__builtin_amdgcn_kernarg_segment_ptr());
    else return (c);
}
// the same for a LATER by-value kernel argument: OFFSET = its byte offset in the kernel-argument segment (arguments are laid out in
// order, each at its own alignment)
template <typename FT, size_t OFFSET, typename CT> __device__ __forceinline__ decltype(auto) kernarg_at(const CT &c) {
    if constexpr (sizeof(FT) == 8 && CMX_PHASE_CONSTS)
        return (*(KernArg<CT> *)((const __attribute__((address_space(4))) char *)__builtin_amdgcn_kernarg_segment_ptr() + OFFSET));
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

}  // namespace cmx
