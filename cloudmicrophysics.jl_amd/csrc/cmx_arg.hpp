// cmx_arg.hpp — Abdul-Razzak & Ghan (2000) aerosol activation: host-folded constants and the per-state point functions shared by the two kernels of
// cmx_arg_kernels.hip (shared aerosol distribution / mode descriptors as columns).  Templates on the VALUE type (float, double, the packed pair f32x2);
// a header so that tools/census/f64_census.cpp can instantiate the same source on its counting type (the Float64 instruction floor).
//
// Reference (src = /root/reference/src): AerosolActivation.jl — coeff_of_curvature :35-40, critical_supersaturation :107-118, max_supersaturation :138-200,
// N_activated_per_mode :235-259, M_activated_per_mode :294-321.
#pragma once
#include <algorithm>
#include <cmath>
#include <type_traits>

#include "cmx_math.hpp"
#include "../../include/cmx.h"

namespace cmx {

template <typename FT> struct ArgModeConsts {
    FT l2_sm_c;      // log2 Sm_i = l2_sm_c + 1.5·log2 A          (Sm = 2/√B (A/(3 r_dry))^1.5)
    FT f, g;         // f_i = f1 exp(f2 ln²σ), g_i = g1 + g2 ln σ
    FT l2_N;         // log2 N_i      (η_i = X / N_i)
    FT N, half_N;
    FT u_c;          // u_i = u_c · (log2 Sm_i − log2 S_max),  u_c = 2 ln2 /(3√2 ln σ_i)
    FT fac;          // 3 ln σ_i √2 / 2
    FT half_M;       // Σ M_j w_j / 2
    // mode-only factors of the S_max sum (AA:170-183), so that the per-state loop needs ONE log2 + ONE exp2 per mode:
    FT inv_N;        // η_i = X / N_i
    FT fN;           // f_i (ζ/η_i)^p1 = fN · (ζ/X)^p1,           fN = f_i N_i^p1
    FT gS;           // g_i (Sm_i²/(η_i+3ζ))^p2 = gS · A^(3 p2) · (η_i+3ζ)^(−p2),   gS = g_i (Sm_i² A⁻³)^p2
    FT inv_sm_c;     // 1/Sm_i = inv_sm_c · A^(−3/2)
    FT c1, c2;       // Σ_i (1/Sm_i²)[…] = A⁻³ · ( (ζ/X)^p1 Σ c1_i + A^(3 p2) Σ c2_i (η_i+3ζ)^(−p2) ),  c1 = fN/Sm_c², c2 = gS/Sm_c²
    FT uc_sm;        // u_i = u_c (log2 Sm_i − log2 S_max) = uc_sm + u_c (1.5 log2 A − log2 S_max),  uc_sm = u_c · l2_sm_c
};

template <typename FT> struct ArgConsts {
    int32_t n_modes;
    // thermodynamics
    FT R_v, R_d, Rv_over_Rd, inv_R_v, T_0, LH_v0, LH_s0, dcp_l, dcp_i, ps_c0, psl_a, psl_b, psi_a, psi_b, inv_T_tr;
    FT cp_d, cpm_qt, cpm_ql, cpm_qi;
    FT inv_K, Rv_over_D, eps_1m, inv_eps_1m, eps_ft;
    FT g, rho_w, inv_rho_w, rho_i, A_c, p1, p2, two_pi_rho_w, four_pi, inv_43pi_rho_w, inv_43pi_rho_i;
    FT l2_Ac_Ttr, l2_two_thirds, l2_two_pi_rho_w, l2_3;   // log2(A_c/T_tr), log2(2/3), log2(2π ρw), log2 3
    FT sum_c1;       // Σ_i c1_i in mode order, in FT arithmetic (the first sum of AA:170-183 is a constant of the distribution)
    ArgModeConsts<FT> m[CMX_ARG_MAX_MODES];
};

template <typename FT, typename AP, typename AD, typename AI, typename TH>
static ArgConsts<FT> make_arg_consts(const AP &ap, const AD &ad, const AI &aip, const TH &tp) {
    ArgConsts<FT> c{};
    const double l2e = 1.4426950408889634074, ln2 = 0.69314718055994530942, pi = 3.14159265358979323846;
    const double eps = (double)Math<FT>::eps_1m();
    c.n_modes = ad.n_modes;
    const double Rv = tp.R_v, T0 = tp.T_0;
    const double dcp_l = (double)tp.cp_v - (double)tp.cp_l, dcp_i = (double)tp.cp_v - (double)tp.cp_i;
    c.R_v = (FT)Rv; c.R_d = (FT)tp.R_d; c.Rv_over_Rd = (FT)(Rv / (double)tp.R_d); c.inv_R_v = (FT)(1.0 / Rv);
    c.T_0 = (FT)T0; c.LH_v0 = (FT)tp.LH_v0; c.LH_s0 = (FT)tp.LH_s0; c.dcp_l = (FT)dcp_l; c.dcp_i = (FT)dcp_i;
    c.ps_c0 = (FT)std::log2((double)tp.press_triple);
    c.psl_a = (FT)(dcp_l / Rv); c.psl_b = (FT)(((double)tp.LH_v0 - dcp_l * T0) / Rv * l2e);
    c.psi_a = (FT)(dcp_i / Rv); c.psi_b = (FT)(((double)tp.LH_s0 - dcp_i * T0) / Rv * l2e);
    c.inv_T_tr = (FT)(1.0 / (double)tp.T_triple);
    c.cp_d = (FT)tp.cp_d; c.cpm_qt = (FT)((double)tp.cp_v - (double)tp.cp_d);
    c.cpm_ql = (FT)((double)tp.cp_l - (double)tp.cp_v); c.cpm_qi = (FT)((double)tp.cp_i - (double)tp.cp_v);
    c.inv_K = (FT)(1.0 / std::fmax((double)aip.K_therm, eps));
    c.Rv_over_D = (FT)(Rv / std::fmax((double)aip.D_vapor, eps));
    c.eps_1m = (FT)eps; c.inv_eps_1m = (FT)(1.0 / eps); c.eps_ft = Math<FT>::eps();
    c.g = (FT)ap.g; c.rho_w = (FT)ap.rho_w; c.inv_rho_w = (FT)(1.0 / (double)ap.rho_w); c.rho_i = (FT)ap.rho_i;
    c.A_c = (FT)(2.0 * (double)ap.sigma * (double)ap.M_w / (double)ap.rho_w / (double)ap.R);   // A = A_c / T
    c.p1 = (FT)ap.p1; c.p2 = (FT)ap.p2;
    c.two_pi_rho_w = (FT)(2.0 * pi * (double)ap.rho_w); c.four_pi = (FT)(4.0 * pi);
    c.l2_Ac_Ttr = (FT)std::log2(2.0 * (double)ap.sigma * (double)ap.M_w / (double)ap.rho_w / (double)ap.R / (double)tp.T_triple);
    c.l2_two_thirds = (FT)std::log2(2.0 / 3.0); c.l2_two_pi_rho_w = (FT)std::log2(2.0 * pi * (double)ap.rho_w); c.l2_3 = (FT)std::log2(3.0);
    c.inv_43pi_rho_w = (FT)(1.0 / (4.0 / 3.0 * pi * (double)ap.rho_w));
    c.inv_43pi_rho_i = (FT)(1.0 / (4.0 / 3.0 * pi * (double)ap.rho_i));
    for (int k = 0; k < ad.n_modes && k < CMX_ARG_MAX_MODES; ++k) {
        const auto &m = ad.modes[k];
        const double ls = std::log((double)m.stdev);
        ArgModeConsts<FT> &o = c.m[k];
        o.l2_sm_c = (FT)(std::log2(2.0 / std::sqrt((double)m.hygroscopicity)) - 1.5 * std::log2(3.0 * (double)m.r_dry));
        o.f = (FT)((double)ap.f1 * std::exp((double)ap.f2 * ls * ls));
        o.g = (FT)((double)ap.g1 + (double)ap.g2 * ls);
        o.l2_N = (FT)std::log2((double)m.N);
        o.N = (FT)m.N; o.half_N = (FT)(0.5 * (double)m.N);
        o.u_c = (FT)(2.0 * ln2 / (3.0 * std::sqrt(2.0) * ls));
        o.fac = (FT)(3.0 * ls * std::sqrt(2.0) / 2.0);
        o.half_M = (FT)((double)m.molar_mass_mix / 2.0);
        const double l2_sm_c = std::log2(2.0 / std::sqrt((double)m.hygroscopicity)) - 1.5 * std::log2(3.0 * (double)m.r_dry);
        o.inv_N = (FT)(1.0 / (double)m.N);
        o.fN = (FT)((double)ap.f1 * std::exp((double)ap.f2 * ls * ls) * std::pow((double)m.N, (double)ap.p1));
        o.gS = (FT)(((double)ap.g1 + (double)ap.g2 * ls) * std::exp2(2.0 * (double)ap.p2 * l2_sm_c));
        o.inv_sm_c = (FT)std::exp2(-l2_sm_c);
        const double inv_sm2 = std::exp2(-2.0 * l2_sm_c);
        o.c1 = (FT)(inv_sm2 * (double)ap.f1 * std::exp((double)ap.f2 * ls * ls) * std::pow((double)m.N, (double)ap.p1));
        o.c2 = (FT)(inv_sm2 * ((double)ap.g1 + (double)ap.g2 * ls) * std::exp2(2.0 * (double)ap.p2 * l2_sm_c));
        o.uc_sm = (FT)(2.0 * ln2 / (3.0 * std::sqrt(2.0) * ls) * l2_sm_c);
        c.sum_c1 = k == 0 ? o.c1 : (FT)(c.sum_c1 + o.c1);
    }
    return c;
}

// ---- erfc ----------------------------------------------------------------------------------------------------------------------------
// The activated number is N ½ (1 − erf u) in the reference (AA:257), the activated mass M ½ erfc(u − …) (AA:319); in Float64 the first is
// N ½ erfc(u) to ≥ 3 digits up to u = 5 and exactly 0 beyond u = 5.9.  Both device forms carry RELATIVE accuracy, so that a small activated
// fraction keeps its leading digits (north_star: 1e-3 on the value).
//
// Float32 (and the packed pair): erfc(x) = t·P₆(t)·e^{−x²}, t = 1/(1 + 0.374 x) for x ≥ 0, 2 − erfc(−x) below — the shape of Abramowitz & Stegun
// 7.1.26, but P₆ is the minimax fit of the RELATIVE error over 0 ≤ x ≤ 10 (tools/gen_erfc_f32.py: 5.8e-7; A&S's degree-4 fit levels the ABSOLUTE
// error of erf at 1.5e-7, i.e. a relative error of 1e-3 at x = 2.7 and unbounded beyond — rounds 1–5 used it for the number, which is why
// VERDICT r05 found 54 % of the smallest mode outside 1e-3).  Evaluated in Float32 the exponent −x²·log2 e carries two roundings:
// relative error ≤ 1e-6 for x < 2, 2.5e-6 at 5, 3.5e-6 at 6, 8e-6 at 9.2 (erfc = floatmin); 0 for +Inf, 2 for −Inf, NaN for NaN.
// One reciprocal, one exponential, 6 Horner steps: two steps more than A&S, three fewer than the Numerical-Recipes erfcc of rounds 3–5.
template <typename VT> __device__ __forceinline__ VT vabs(VT x) {
#if CMX_HAVE_PACKED
    if constexpr (lanes_of<VT>::value == 2) return __builtin_elementwise_abs(x);
    else
#endif
        return x < VT(0) ? -x : x;
}
template <> __device__ __forceinline__ float vabs<float>(float x) { return __builtin_fabsf(x); }
template <> __device__ __forceinline__ double vabs<double>(double x) { return __builtin_fabs(x); }
template <typename VT> __device__ __forceinline__ VT erfc_f32(VT x) {
    using M = Math<VT>;
    const VT ax = vabs(x);
    const VT t = M::rcp(M::fma(ax, VT(0.374f), VT(1.0f)));
    VT q = VT(-0.137162139f);
    q = M::fma(q, t, VT(0.399237749f));
    q = M::fma(q, t, VT(-0.143235013f));
    q = M::fma(q, t, VT(0.291505698f));
    q = M::fma(q, t, VT(0.163272839f));
    q = M::fma(q, t, VT(0.215641374f));
    q = M::fma(q, t, VT(0.210738917f));
    const VT e = (q * t) * M::exp2((ax * ax) * VT(-1.4426950408889634f));
    return x >= VT(0.0f) ? e : VT(2.0f) - e;
}
#ifndef CMX_ARG_LEAN_ERFC
#define CMX_ARG_LEAN_ERFC 1      // 0: OCML erfc (A/B switch)
#endif
// erfc_dev: the activated NUMBER.  Float64: the table-driven lean::erfc (cmx_lean_f64.hpp: relative error ≤ (2 + x²)·2e-16 up to 6.5; the
// reference's ½(1 − erf u) is exactly 0 beyond u = 5.9) — 45 instructions against OCML's 135, five calls per state
template <typename VT> __device__ __forceinline__ VT erfc_dev(VT x) {
    if constexpr (sizeof(typename Math<VT>::Scalar) == 8) return CMX_ARG_LEAN_ERFC ? lean::erfc(x) : ::erfc(x);
    else return erfc_f32<VT>(x);
}
// erfc_rel_dev: the activated MASS (relative accuracy over the whole range: the reference evaluates erfc itself).  Float64: OCML
template <typename VT> __device__ __forceinline__ VT erfc_rel_dev(VT x) {
    if constexpr (sizeof(typename Math<VT>::Scalar) == 8) return ::erfc(x);
    else return erfc_f32<VT>(x);
}

#ifndef CMX_ARG_P2_ROOTS
#define CMX_ARG_P2_ROOTS 1      // A/B switch for the p2 = ¾ path of the Float64 S_max sum (arg_point)
#endif
// y^(−¾) for the p2 = ¾ path of the Float64 S_max sum
__device__ __forceinline__ double arg_pow_m34(double y) {
#if CMX_F64_FINITE_FORMS
    return lean::pow_m34_pos(y);
#else
    const double t = lean::rsqrt(y);
    return t * lean::sqrt(t);
#endif
}

template <typename VT, int NM> struct ArgOut { VT smax; VT n[NM]; VT m[NM]; };

// ---- one thermodynamic state, in three stages shared by the two kernels: arg_pre (everything in front of the sum over the modes),
// the mode sum (kernel-specific: host-folded mode constants, or mode descriptors streamed from columns), arg_smax (S_max from the sum).
// Templates on the VALUE type VT (float, double, or the packed pair f32x2 — cmx_math.hpp); the constants are scalars of Math<VT>::Scalar.
//
// Round 6: the S_max sum in the log2 domain.  With A = A_c/T, ζ = ⅔A√(αw/G), X = (αw/G)^1.5/(2πρwγ), η_i = X/N_i (AA:35-40,168-183):
//     1/S_max² = Σ_i (1/Sm_i²)[f_i (ζ/η_i)^p1 + g_i (Sm_i²/(η_i + 3ζ))^p2]
//              = A⁻³(ζ/X)^p1 · Σ c1_i  +  A^(3p2−3)(3ζ)^(−p2) · Σ c2_i (1 + Q/N_i)^(−p2),          Q = X/(3ζ)
//              = 2^l2_G · [ Σ c2_i (1 + Q/N_i)^(−p2)  +  R · Σ c1_i ],
//     l2_G = (2p2 − 2)·log2 A^1.5 − p2·log2(3ζ),     R = 2^( p1·(log2 ζ − log2 X) − 2·log2 A^1.5 − l2_G ),
// so that log2 S_max = −½(l2_G + log2[…]) — which is what the erfc arguments need — costs three transcendentals per state (Q, R, one log2)
// where rounds 1–5 formed ζ, X, A⁻³(ζ/X)^p1, A^(3p2−3), 1/√· and log2 S_max: six.  S_max itself is one more exponential, taken only where
// it is stored or the sink correction needs it.  No intermediate leaves the log2 domain, so the Float32 range is never at stake.
// The sink correction (AA:187-197) S_max = S_ARG (αw − K_ice(ξ − 1)) / (αw + (K_ice ξ + K_liq) S_ARG) depends on the sums only through S_ARG: its three
// state-only terms are formed here, in front of the mode loop, so that nothing of the thermodynamics stays alive across it (round 6: the per-element kernel
// kept 15 values per state through its first pass — 136–152 VGPRs with sinks at 8 Float64 modes).
template <typename VT> struct ArgPre {
    VT l2_A15, Q, R, l2_G;
    VT aw, sink_num, sink_den;        // SINKS only: αw, αw − K_ice(ξ − 1), K_ice ξ + K_liq
};
template <typename VT, bool SINKS, typename C>
__device__ __forceinline__ ArgPre<VT> arg_pre(const C &c, VT T, VT p, VT w, VT q_tot, VT q_liq, VT q_ice, VT N_liq, VT N_ice) {
    using M = Math<VT>;
    ArgPre<VT> s;
    const VT inv_T = M::rcp_nz(T);                  // temperature, pressure, R_m, cp_m: positive and finite
    // TD.gas_constant_air, cp_m, latent heat, air density, vapour pressures — AA:152-160
    const VT R_m = c.R_d * (VT(1) + (c.Rv_over_Rd - 1) * q_tot - c.Rv_over_Rd * (q_liq + q_ice));
    const VT cp_m = M::fma(c.cpm_qi, q_ice, M::fma(c.cpm_ql, q_liq, M::fma(c.cpm_qt, q_tot, c.cp_d)));
    const VT L_v = M::fma(c.dcp_l, T - c.T_0, c.LH_v0);
    const VT inv_Rm = M::rcp_nz(R_m), inv_cp = M::rcp_nz(cp_m);
    const VT rho_air = p * inv_Rm * inv_T;
    const VT p_v = (q_tot - q_liq - q_ice) * rho_air * c.R_v * T;
    const VT l2_TT = M::log2(T * c.inv_T_tr), dinvT = c.inv_T_tr - inv_T;
    const VT l2_pvs = M::fma(c.psl_a, l2_TT, M::fma(c.psl_b, dinvT, c.ps_c0));
    const VT inv_pvs = M::exp2_fin(-l2_pvs);          // overflow → +Inf is still right (capped by inv_eps_1m below)
    const VT LoRT = L_v * c.inv_R_v * inv_T;
    // 1/G_liq = L/(K T)(L/(R_v T) − 1) + R_v T/(D max(p_vs, ϵ))  (Common.jl:47-63); 1/max(p_vs, ϵ) = min(1/p_vs, 1/ϵ).  Only the
    // reciprocal of G = G_liq/ρ_w enters S_max (αw/G), so G itself is formed only for the sink terms.
    const VT inv_G_liq = M::fma(L_v * c.inv_K * inv_T, LoRT - VT(1), c.Rv_over_D * T * M::min(inv_pvs, c.inv_eps_1m));
    const VT ratio = p_v * inv_pvs;
    const VT alpha = ratio * (LoRT * c.g * inv_cp * inv_T - c.g * inv_Rm * inv_T);                       // AA:164
    const VT gamma = M::fma(ratio * R_m * L_v, LoRT * inv_cp * M::rcp_nz(p), c.R_v * T * inv_pvs);       // AA:165
    const VT aw = alpha * w;
    const VT aw_over_G = aw * c.rho_w * inv_G_liq;
    s.aw = aw; s.sink_num = aw; s.sink_den = VT(0);
    if constexpr (SINKS) {   // liquid / ice sink terms — AA:187-196
        const VT L_s = M::fma(c.dcp_i, T - c.T_0, c.LH_s0);
        const VT l2_pvi = M::fma(c.psi_a, l2_TT, M::fma(c.psi_b, dinvT, c.ps_c0));
        const VT p_vi = M::exp2(l2_pvi), p_vs = M::exp2(l2_pvs);
        const VT G = M::rcp(inv_G_liq) * c.inv_rho_w;
        const VT r_liq = N_liq < c.eps_ft ? VT(0) : M::exp2(M::log2(rho_air * q_liq * M::rcp(N_liq) * c.inv_43pi_rho_w) * VT(1.0 / 3.0));
        const VT K_liq = c.four_pi * c.rho_w * N_liq * r_liq * G * gamma;
        const VT gamma_i = M::fma(ratio * R_m * L_v, L_s * c.inv_R_v * inv_cp * inv_T * M::rcp(p), c.R_v * T * inv_pvs);
        const VT r_ice = N_ice < c.eps_ft ? VT(0) : M::exp2(M::log2(rho_air * q_ice * M::rcp(N_ice) * c.inv_43pi_rho_i) * VT(1.0 / 3.0));
        const VT LoRT_s = L_s * c.inv_R_v * inv_T;
        const VT G_ice = M::rcp(M::fma(L_s * c.inv_K * inv_T, LoRT_s - VT(1), c.Rv_over_D * T * M::rcp(M::max(p_vi, c.eps_1m))));
        const VT xi = p_vs * M::rcp(p_vi);
        const VT K_ice = c.four_pi * N_ice * r_ice * G_ice * gamma_i;
        s.sink_num = aw - K_ice * (xi - VT(1));
        s.sink_den = M::fma(K_ice, xi, K_liq);
        // Float64: the sink terms are finished HERE — left alone their table-driven functions are interleaved with the ones below and the whole
        // thermodynamic state stays in registers (the memory clobber keeps the next table reads behind this point, like the erfc loop of arg_point)
        #if !defined(CMX_HOST_BUILD)
        if constexpr (sizeof(typename M::Scalar) == 8) asm volatile("" : "+v"(s.sink_num), "+v"(s.sink_den) : : "memory");
#endif
    }
    // log2 A = log2(A_c/T_tr) − log2(T/T_tr) (already formed for p_vs), log2 ζ and log2 X from log2(αw/G) and log2 γ.  w ≤ 0: the reference
    // gives NaN (ζ/η = 0/0, √ of a negative) — here log2 of a non-positive αw/G: NaN, or −Inf − (−Inf) = NaN in Q
    const VT l2_awG = M::log2(aw_over_G);
    const VT l2_A = c.l2_Ac_Ttr - l2_TT;
    const VT l2_A15 = VT(1.5) * l2_A;
    const VT l2_zeta = c.l2_two_thirds + l2_A + VT(0.5) * l2_awG;
    const VT l2_X = M::fma(VT(1.5), l2_awG, -(c.l2_two_pi_rho_w + M::log2(gamma)));
    const VT l2_3zeta = c.l2_3 + l2_zeta;
    s.Q = M::exp2_fin(l2_X - l2_3zeta);
    s.l2_G = M::fma(c.p2 + c.p2 - 2, l2_A15, -(c.p2 * l2_3zeta));
    s.R = M::exp2_fin(M::fma(c.p1, l2_zeta - l2_X, VT(-2) * l2_A15) - s.l2_G);
    s.l2_A15 = l2_A15;
    return s;
}
// (1 + Q/N_i)^(−p2) of one mode
template <typename VT, typename C> __device__ __forceinline__ VT arg_pow_p2(const C &c, VT y, bool p2_is_34) {
    using M = Math<VT>;
    if constexpr (sizeof(typename M::Scalar) == 8) {
        // Float64 with ARG2000's own exponent p2 = ¾ (a wave-uniform test): y^(−¾) = t·√t with t = 1/√y — a reciprocal square root and
        // a square root (hardware seed + Newton steps, ≈ 30 instructions) instead of a table-driven log2 and exp2 (≈ 45) per mode.
        // y = 1 + Q/N_k ≥ 1 for w > 0; w ≤ 0 gives NaN in the reference as well, so no 0 / Inf cases to keep
        if (p2_is_34) return arg_pow_m34(y);
    }
    return M::exp2(-c.p2 * M::log2(y));
}
// S_max (AA:185-199) from the two sums: returns log2 S_max (the erfc arguments need nothing else); S_max itself → `smax` where `want_smax`
// (a wave-uniform flag: the S_max column is requested) or the sink correction is compiled in
template <typename VT, bool SINKS>
__device__ __forceinline__ VT arg_smax(const ArgPre<VT> &s, VT sum1, VT sum2, bool want_smax, VT &smax) {
    using M = Math<VT>;
    const VT l2_S = VT(-0.5) * (s.l2_G + M::log2(M::fma(s.R, sum1, sum2)));                                   // log2 of AA:185
    if constexpr (SINKS) {   // liquid / ice sink correction — AA:187-197 (terms: arg_pre)
        const VT S_arg = M::exp2(l2_S);
        const VT sm = S_arg * s.sink_num * M::rcp(M::fma(s.sink_den, S_arg, s.aw));
        smax = sm < VT(0) ? VT(0) : sm;   // AA:199 max(0, S_max) with Julia's NaN rule: a NaN from any input reaches every output below
        return M::log2(smax);
    } else {
        // N_liq = N_ice = 0: K_liq = K_ice = 0 ⇒ S_max = S_max_ARG·αw/αw; 2^x ≥ 0, so AA:199's max(0, ·) changes nothing and a NaN stays a NaN
        smax = VT(0);
        if (want_smax) smax = M::exp2(l2_S);
        return l2_S;
    }
}

// one thermodynamic state (or pair of states) of the shared-distribution kernel.  NM = compile-time mode count (1…8)
template <typename VT, int NM, bool SINKS, typename C>
__device__ __forceinline__ ArgOut<VT, NM> arg_point(const C &c, VT T, VT p, VT w, VT q_tot, VT q_liq, VT q_ice, VT N_liq, VT N_ice, bool want_N, bool want_M,
                                                    bool want_smax) {
    using M = Math<VT>;
    using S = typename M::Scalar;
    ArgOut<VT, NM> o;
    const ArgPre<VT> s = arg_pre<VT, SINKS>(c, T, p, w, q_tot, q_liq, q_ice, N_liq, N_ice);
    // per mode one FMA, one log2, one multiply, one exp2 and one accumulating FMA; Σ c1_i is a constant of the distribution
    VT sum2 = VT(0);
    const bool p2_34 = sizeof(S) == 8 && CMX_ARG_P2_ROOTS && c.p2 == S(0.75);
#pragma unroll
    // (a multiply and an add per mode, not an fma: the reference sums separately rounded terms, so its totals do not depend on the ORDER of two modes —
    // test/aerosol_activation_tests.jl:222-234 asserts that with `==` — and a + b = b + a holds for rounded products, not for fma(c, p, a))
    for (int k = 0; k < NM; ++k) sum2 = sum2 + c.m[k].c2 * arg_pow_p2<VT>(c, M::fma(s.Q, c.m[k].inv_N, VT(1)), p2_34);
    const VT l2_smax = arg_smax<VT, SINKS>(s, VT(c.sum_c1), sum2, want_smax, o.smax);
    const VT dl0 = s.l2_A15 - l2_smax;                             // log2(Sm_i / S_max) = l2_sm_c + dl0
    // phase boundary (cmx_math.hpp consts_after; a no-op unless the kernel reads its constants through the kernel-argument pointer): the erfc loop's
    // mode constants are loaded here, not next to the S_max sum's
    const auto *cm = consts_after(c, dl0).m;
#pragma unroll
    for (int k = 0; k < NM; ++k) {
        const VT u = M::fma(cm[k].u_c, dl0, cm[k].uc_sm);       // AA:255   (= ln(sm/smax)/fac, AA:316)
        o.n[k] = want_N ? cm[k].half_N * erfc_dev<VT>(u) : VT(0);                  // N ½ (1 − erf u)      AA:257
        // Float64: one mode's erfc at a time — left alone the NM independent table-driven evaluations are interleaved and all their
        // LDS reads hoisted (422 VGPRs for 5 modes × 2 states: one wave per SIMD).  The asm pins the mode's result here and, with its
        // memory clobber, keeps the next mode's table reads behind it.
#if !defined(CMX_HOST_BUILD)
        if constexpr (sizeof(S) == 8 && CMX_ARG_LEAN_ERFC) asm volatile("" : "+v"(o.n[k]) : : "memory");
#endif
        o.m[k] = want_M ? cm[k].half_M * erfc_rel_dev<VT>(u - cm[k].fac) : VT(0); // M/2 erfc(u − fac)    AA:319
    }
    return o;
}

}  // namespace cmx
