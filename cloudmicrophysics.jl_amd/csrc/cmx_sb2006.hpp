// cmx_sb2006.hpp — Seifert–Beheng 2006 two-moment warm-rain point function.
//
// One evaluation of everything the reference computes per grid point on the path
//   bulk_microphysics_tendencies(::Microphysics2Moment, mp{WR,Nothing}, …)   BMT:820-854
//   └ warm_rain_tendencies_2m                                               BMT:707-782
// (reference = /root/reference/src, BMT = BulkMicrophysicsTendencies.jl, CM2 = Microphysics2M.jl,
// NonEq = MicrophysicsNonEq.jl, TDI = ThermodynamicsInterface.jl), restructured for the GPU:
//   * the saturation vapour pressure (3× in the reference: cond/evap, supersaturation, G) and the
//     rain PSD parameters (3×: evaporation, self-collection, breakup; 4× with velocities) are
//     evaluated ONCE;
//   * every cbrt / pow of the mean drop mass shares one log2(x̄_r); all powers run in the log2
//     domain (cmx_math.hpp);
//   * all parameter-only sub-expressions are folded on the host into SbConsts (double arithmetic,
//     then rounded to FT) and arrive in SGPRs through the kernel-argument segment;
//   * gates are branch-free selects, exactly the reference's `ifelse` predicates.
#pragma once
#include <cmath>
#include <cstdint>

#include "../../include/cmx.h"
#include "cmx_math.hpp"

namespace cmx {

// VEL_CHEN: the density-dependent coefficients of the Chen-2022 terms from host-fitted polynomials in ρ (cmx_math.hpp ChenLog); VEL_CHEN_GEN: run-time Γ for
// parameter sets the fit cannot represent (make_chen_log false)
enum : int { VEL_NONE = 0, VEL_SB = 1, VEL_CHEN = 2, VEL_CHEN_GEN = 3 };
#ifndef CMX_SB_TSTAR_RCP
#define CMX_SB_TSTAR_RCP 1        // A/B switch of the evaporation's t* (below)
#endif

// ---------------------------------------------------------------------------------------------
// Host-folded constants.  Plain FT members only (kernel argument → SGPRs).
// ---------------------------------------------------------------------------------------------
template <typename FT> struct SbConsts {
    // thermodynamics (Thermodynamics.jl restatement, see oracle/cmx_oracle_impl.h)
    FT T_0, LH_v0, dcp, R_v, inv_R_v;
    FT ps_c0, ps_a, ps_b, inv_T_tr;     // log2 p_sat = c0 + a·log2(T/T_tr) + b·(1/T_tr − 1/T)
    FT cp_d, cpm_qt, cpm_ql;            // cp_m = cp_d + cpm_qt·q_tot + cpm_ql·q_liq
    FT tau_ce;                          // CondEvap2M.τ_relax
    FT inv_K, Rv_over_D, eps_1m;        // G_func_liquid (Common.jl:47-63)
    // rain PSD (CM2:67-110)
    FT xr_min, xr_max, N0_min, N0_max, lam_min, lam_max, pi_rho_w;
    // evaporation (CM2:780-828)
    FT l2_6xstar, l2_Drc;               // log2(6 x*), log2(6/(π ρw))
    // log2 of the PSD limiters (the rain PSD parameters are formed in the log2 domain)
    FT l2_pi_rho_w, l2_xr_min, l2_xr_max, l2_N0_min, l2_N0_max, l2_lam_min, l2_lam_max, inv_eps_1m;
    FT ga_c1, ga_e1, ga_c2, ga_e2;      // Γ_incl(−1, t)   (CM2:746-753)
    FT gb_c1, gb_e1, gb_c2, gb_e2;      // Γ_incl(β_vent_0, t)
    FT a_vent_1, bSc_vent_1;            // b·∛Sc folded (the order-0 pair lives in ga_c*, gb_c*)
    FT sqrt_alpha_nu, beta, ev_rho0_q;  // √(α/ν_air), β, ρ0^(1/4)
    FT sqrt_alpha_nu_rho0q;             // their product, folded on the host (a product of two kernel arguments is a VALU multiply per point)
    FT two_pi, l2_gate_N;               // log2(eps(FT)·x*) of the evaporation number gate
    FT tstar_Dr;                        // t*·Dr = ∛(6 x*)·∛(6/(π ρw)) — parameter-only: t* = ∛(6 x*/x̄_r) and Dr = ∛(6 x̄_r/(π ρw)) share x̄_r (round 5)
    // autoconversion / cloud self-collection (CM2:396-427, 488-501)
    FT sqrt_kfac, x_star, inv_x_star, acnv_A, acnv_a, acnv_b, acnv_rho0, ksc;
    // accretion (CM2:445-470)
    FT kcr_s, tau_0, accr_c;            // kcr·√ρ0
    // rain self-collection / breakup (CM2:545-601)
    FT krr_s, kappa_rr_K, self_d;       // krr·√ρ0(pdf_r), κ_rr·∛(π ρw/36)
    FT Deq, Dr_th, kbr, kappa_br_l2e;
    // number adjustment (CM2:882-891)
    FT inv_tau_na, inv_xc_min, inv_xc_max, inv_xr_min, inv_xr_max;
    // SB2006 rain velocity (CM2:685-702, 720-739)
    FT vel_s, aR, bR, cR, rc2, e_rc2cR; // √ρ0(vel), …, 2·r_c, exp(−2 r_c c_R)
    // Chen-2022 rain velocity (Common.jl:290-302, 414-422)
    FT ch_rho0_l2e, ch_a[3], ch_a3_pow, ch_b[3], ch_b_rho, ch_c1000[3], l2_1000;
    ChenLog<FT> chl;           // log2 of the density-dependent coefficient of every term, k = 0 and k = 3, as polynomials in ρ (cmx_math.hpp)
};

// The limited rain PSD clamps with v_med3 (clamp_ordered): every (min, max) pair of the limiters must be ordered and positive.
// Entry points return CMX_ERR_BAD_ARG otherwise (Base.clamp with lo > hi is not a clamp either).
template <typename WR> inline bool sb_limiters_ok(const WR &wr) {
    const auto &p = wr.seifert_beheng.pdf_r;
    return p.xr_min > 0 && p.xr_min <= p.xr_max && p.N0_min > 0 && p.N0_min <= p.N0_max && p.lambda_min > 0 && p.lambda_min <= p.lambda_max;
}

template <typename FT, typename WR, typename TH, typename VL>
inline SbConsts<FT> make_sb_consts(const WR &wr, const TH &tp, const VL *vel, double eps_1m) {
    SbConsts<FT> c{};
    const auto &sb = wr.seifert_beheng;
    const double pi = 3.14159265358979323846;
    const double l2e = 1.4426950408889634074;
    // thermo
    const double dcp = (double)tp.cp_v - (double)tp.cp_l;
    c.T_0 = (FT)tp.T_0;
    c.LH_v0 = (FT)tp.LH_v0;
    c.dcp = (FT)dcp;
    c.R_v = (FT)tp.R_v;
    c.inv_R_v = (FT)(1.0 / (double)tp.R_v);
    c.ps_c0 = (FT)std::log2((double)tp.press_triple);
    c.ps_a = (FT)(dcp / (double)tp.R_v);
    c.ps_b = (FT)(((double)tp.LH_v0 - dcp * (double)tp.T_0) / (double)tp.R_v * l2e);
    c.inv_T_tr = (FT)(1.0 / (double)tp.T_triple);
    c.cp_d = (FT)tp.cp_d;
    c.cpm_qt = (FT)((double)tp.cp_v - (double)tp.cp_d);
    c.cpm_ql = (FT)((double)tp.cp_l - (double)tp.cp_v);
    c.tau_ce = (FT)wr.condevap_tau_relax;
    const double K_safe = std::fmax((double)wr.air_properties.K_therm, eps_1m);
    const double D_safe = std::fmax((double)wr.air_properties.D_vapor, eps_1m);
    c.inv_K = (FT)(1.0 / K_safe);
    c.Rv_over_D = (FT)((double)tp.R_v / D_safe);
    c.eps_1m = (FT)eps_1m;
    // rain PSD
    const auto &pr = sb.pdf_r;
    c.xr_min = (FT)pr.xr_min;
    c.xr_max = (FT)pr.xr_max;
    c.N0_min = (FT)pr.N0_min;
    c.N0_max = (FT)pr.N0_max;
    c.lam_min = (FT)pr.lambda_min;
    c.lam_max = (FT)pr.lambda_max;
    c.pi_rho_w = (FT)(pi * (double)pr.rho_w);
    // evaporation
    const auto &ev = sb.evap;
    const double x_star_ev = (double)pr.xr_min;  // CM2:795
    c.l2_6xstar = (FT)std::log2(6.0 * x_star_ev);
    c.l2_Drc = (FT)std::log2(6.0 / (pi * (double)pr.rho_w));
    c.l2_pi_rho_w = (FT)std::log2(pi * (double)pr.rho_w);
    c.l2_xr_min = (FT)std::log2((double)pr.xr_min); c.l2_xr_max = (FT)std::log2((double)pr.xr_max);
    c.l2_N0_min = (FT)std::log2(std::fmax((double)pr.N0_min, 1e-300)); c.l2_N0_max = (FT)std::log2(std::fmax((double)pr.N0_max, 1e-300));
    c.l2_lam_min = (FT)std::log2(std::fmax((double)pr.lambda_min, 1e-300)); c.l2_lam_max = (FT)std::log2(std::fmax((double)pr.lambda_max, 1e-300));
    c.inv_eps_1m = (FT)(1.0 / eps_1m);
    auto gincl = [](double a, FT &c1, FT &e1, FT &c2, FT &e2) {
        c1 = (FT)(0.33 - 0.7 * a);
        e1 = (FT)(0.08 - 0.93 * a);
        c2 = (FT)(1.34 - 0.1 * a);
        e2 = (FT)(0.8 - a);
    };
    gincl(-1.0, c.ga_c1, c.ga_e1, c.ga_c2, c.ga_e2);
    gincl((double)ev.beta_vent_0, c.gb_c1, c.gb_e1, c.gb_c2, c.gb_e2);
    const double cbrt_Sc = std::cbrt((double)wr.air_properties.nu_air / D_safe);
    // 1/(c1 t^e1 + c2 t^e2) scaled by a_vent_0 resp. b_vent_0·∛Sc: fold the scale into c1, c2
    {
        const double sa = (double)ev.a_vent_0_coeff, sb = (double)ev.b_vent_0_coeff * cbrt_Sc;
        c.ga_c1 = (FT)((0.33 + 0.7) / sa); c.ga_c2 = (FT)((1.34 + 0.1) / sa);
        const double ab = (double)ev.beta_vent_0;
        c.gb_c1 = (FT)((0.33 - 0.7 * ab) / sb); c.gb_c2 = (FT)((1.34 - 0.1 * ab) / sb);
    }
    c.a_vent_1 = (FT)ev.a_vent_1;
    c.bSc_vent_1 = (FT)((double)ev.b_vent_1 * cbrt_Sc);
    c.sqrt_alpha_nu = (FT)std::sqrt((double)ev.alpha / (double)wr.air_properties.nu_air);
    c.beta = (FT)ev.beta;
    c.ev_rho0_q = (FT)std::sqrt(std::sqrt((double)ev.rho_0));
    c.sqrt_alpha_nu_rho0q = (FT)(std::sqrt((double)ev.alpha / (double)wr.air_properties.nu_air) * std::sqrt(std::sqrt((double)ev.rho_0)));
    c.two_pi = (FT)(2.0 * pi);
    c.l2_gate_N = (FT)std::log2((double)Math<FT>::eps() * x_star_ev);
    c.tstar_Dr = (FT)(std::cbrt(6.0 * x_star_ev) * std::cbrt(6.0 / (pi * (double)pr.rho_w)));
    // autoconversion
    const auto &ac = sb.acnv;
    const double nu_c = (double)sb.pdf_c.nu_c;
    const double kfac = (double)ac.kcc / 20.0 / (double)ac.x_star * (nu_c + 2) * (nu_c + 4) /
                        ((nu_c + 1) * (nu_c + 1));
    c.sqrt_kfac = (FT)std::sqrt(kfac);
    c.x_star = (FT)ac.x_star;
    c.inv_x_star = (FT)(1.0 / (double)ac.x_star);
    c.acnv_A = (FT)ac.A;
    c.acnv_a = (FT)ac.a;
    c.acnv_b = (FT)ac.b;
    c.acnv_rho0 = (FT)ac.rho_0;
    c.ksc = (FT)((double)ac.kcc * (nu_c + 2) / (nu_c + 1) * (double)ac.rho_0);
    // accretion
    c.kcr_s = (FT)((double)sb.accr.kcr * std::sqrt((double)sb.accr.rho_0));
    c.tau_0 = (FT)sb.accr.tau_0;
    c.accr_c = (FT)sb.accr.c;
    // rain self-collection / breakup
    c.krr_s = (FT)((double)sb.self.krr * std::sqrt((double)pr.rho_0));
    c.kappa_rr_K = (FT)((double)sb.self.kappa_rr * std::cbrt(pi * (double)pr.rho_w / 36.0));
    c.self_d = (FT)sb.self.d;
    c.Deq = (FT)sb.brek.Deq;
    c.Dr_th = (FT)sb.brek.Dr_th;
    c.kbr = (FT)sb.brek.kbr;
    c.kappa_br_l2e = (FT)((double)sb.brek.kappa_br * l2e);
    // number adjustment
    c.inv_tau_na = (FT)(1.0 / (double)sb.numadj.tau);
    c.inv_xc_min = (FT)(1.0 / (double)sb.pdf_c.xc_min);
    c.inv_xc_max = (FT)(1.0 / (double)sb.pdf_c.xc_max);
    c.inv_xr_min = (FT)(1.0 / (double)pr.xr_min);
    c.inv_xr_max = (FT)(1.0 / (double)pr.xr_max);
    // velocities
    if (vel) {
        const auto &v = vel->sb2006;
        c.vel_s = (FT)std::sqrt((double)v.rho_0);
        c.aR = (FT)v.aR;
        c.bR = (FT)v.bR;
        c.cR = (FT)v.cR;
        const double rc2 = -1.0 / (double)v.cR * std::log((double)v.aR / (double)v.bR);  // 2·r_c
        c.rc2 = (FT)rc2;
        c.e_rc2cR = (FT)std::exp(-rc2 * (double)v.cR);
        const auto &ch = vel->chen2022;
        c.ch_rho0_l2e = (FT)((double)ch.rho_0 * l2e);
        for (int i = 0; i < 3; ++i) {
            c.ch_a[i] = (FT)ch.a[i];
            c.ch_b[i] = (FT)ch.b[i];
            c.ch_c1000[i] = (FT)((double)ch.c[i] * 1000.0);
        }
        c.ch_a3_pow = (FT)ch.a3_pow;
        c.ch_b_rho = (FT)ch.b_rho;
        c.l2_1000 = (FT)std::log2(1000.0);
        (void)make_chen_log<FT>(ch, c.chl);        // the entry points test the fit themselves (chen_vel_kind)
    }
    return c;
}

// ---------------------------------------------------------------------------------------------
// Per-point result: every process rate of SB2006_2M_kernel (test/gpu_tests.jl:220-235) + cond/evap.
// The fused kernel sums them (unused ones are dead-code-eliminated per instantiation).
// ---------------------------------------------------------------------------------------------
template <typename FT> struct SbRates {
    FT cond;                                            // NonEq:117-140 [kg/kg/s]
    FT au_dq_lcl, au_dN_lcl, au_dq_rai, au_dN_rai;      // CM2:396-427
    FT lsc;                                             // CM2:488-501 [1/m3/s]
    FT lsc_plus_au;                                     // lsc + au_dN_lcl = −k_sc/ρ·L² (the autoconversion part cancels), gated like lsc
    FT ac_dq_lcl, ac_dN_lcl, ac_dq_rai;                 // CM2:445-470
    FT rsc, rbr;                                        // CM2:545-601 [1/m3/s]
    FT evN, evq;                                        // CM2:780-828
    FT na_lcl, na_rai;                                  // CM2:882-891 [1/kg/s]
    FT vt_n, vt_m;                                      // CM2:685-719 [m/s]
    FT inv_rho;
};

// ---- CM2.cloud_terminal_velocity — Microphysics2M.jl:647-664 ----------------------------------------------------------------
// With B = (x̄ Γ(z₁)/Γ(z₂))^(−μ) (log_pdf_cloud_parameters_mass :174-192) the two moments collapse to one power of the mean
// droplet mass x̄ = ρq/N:  vt_n = pref·K₂₃·x̄^(2/3),  vt_m = pref·K₅₃·x̄^(5/3)·N/(ρq) = pref·K₅₃·x̄^(2/3),
// K_n = (Γ(z₁)/Γ(z₂))^n · Γ(z₁ + n/μ)/Γ(z₁) parameter-only (host, double).
template <typename FT> struct CloudVelConsts { FT pre_c, rho_w, K23, K53; };
template <typename FT, typename PDF, typename VEL>
inline CloudVelConsts<FT> make_cloud_vel_consts(const PDF &pdf, const VEL &vel) {
    const double pi = 3.14159265358979323846, nu = pdf.nu_c, mu = pdf.mu_c, z1 = (nu + 1.0) / mu;
    const double dlg = (double)pdf.loggamma_z1 - (double)pdf.loggamma_z2;   // log Γ(z₁)/Γ(z₂)
    CloudVelConsts<FT> c;
    c.rho_w = (FT)vel.rho_w;
    c.pre_c = (FT)(1.0 / 18.0 * std::cbrt(std::pow(6.0 / (double)vel.rho_w / pi, 2.0)) * (double)vel.grav / (double)vel.nu_air);
    c.K23 = (FT)std::exp(2.0 / 3.0 * dlg + std::lgamma(z1 + 2.0 / 3.0 / mu) - std::lgamma(z1));
    c.K53 = (FT)std::exp(5.0 / 3.0 * dlg + std::lgamma(z1 + 5.0 / 3.0 / mu) - std::lgamma(z1));
    return c;
}
template <typename FT, typename CV>      // FT: the value type; CV: CloudVelConsts of its scalar type
__device__ __forceinline__ void sb2006_cloud_velocity(const CV &c, FT q, FT r, FT N, FT &vt_n, FT &vt_m) {
    using M = Math<FT>;
    const FT sq = M::max(q, M::eps()), sN = M::max(N, M::eps());
    const FT x23 = M::exp2(FT(2.0 / 3.0) * M::log2(r * sq * M::rcp(sN)));
    const FT pref = c.pre_c * (c.rho_w * M::rcp(r) - FT(1));
    const typename M::Mask none = m_or(N < M::eps(), q < M::eps());
    vt_n = none ? FT(0) : pref * c.K23 * x23;
    vt_m = none ? FT(0) : pref * c.K53 * x23;
}

// The point functions below are templates on the VALUE type `FT`: float, double, or the packed pair f32x2 (two Float32 points per value, cmx_math.hpp) — `FT(x)`
// broadcasts a constant, comparisons give `typename Math<FT>::Mask` (bool, or a lane mask that is combined with | and selected with ?:), and the constants
// struct C always holds SCALARS of Math<FT>::Scalar.
// ---- rain PSD parameters (CM2:67-110) from the safe values, in the log2 domain ------------------------------
// All four limiters (Eq. 94-97) are clamps of monotone power laws of L_rai and N_rai, so they are applied to the
// log2 values: two log2 in, then every derived quantity is one exp2 (11 → 5-6 transcendentals).
// Shared by the point function and by the flux-only evaluation of the column kernel (same instruction sequence → same bits).
template <typename FT> struct SbRainPsd { FT l2_xr, l2_lam; };
template <typename FT, bool LIMITED, typename C>
__device__ __forceinline__ SbRainPsd<FT> sb2006_rain_psd(const C &c, FT L_rai, FT sN_rai) {
    using M = Math<FT>;
    const FT l2_L = M::log2(L_rai), l2_N = M::log2(sN_rai);
    SbRainPsd<FT> p;
    if constexpr (LIMITED) {
        // the limiter pairs are ordered (xr_min ≤ xr_max, N0_min ≤ N0_max, λ_min ≤ λ_max: checked by the entry point), so each clamp
        // is one v_med3_f32
        const FT l2_xt = clamp_ordered(l2_L - l2_N, c.l2_xr_min, c.l2_xr_max);                               // Eq. 94
        const FT l2_N0 = clamp_ordered(l2_N + (c.l2_pi_rho_w - l2_xt) * FT(1.0 / 3.0), c.l2_N0_min, c.l2_N0_max);   // Eq. 95
        p.l2_lam = clamp_ordered((c.l2_pi_rho_w + l2_N0 - l2_L) * FT(0.25), c.l2_lam_min, c.l2_lam_max);     // Eq. 96
        p.l2_xr = clamp_ordered(l2_L + p.l2_lam - l2_N0, c.l2_xr_min, c.l2_xr_max);                          // Eq. 97
    } else {
        p.l2_xr = l2_L - l2_N;
        p.l2_lam = (c.l2_pi_rho_w - p.l2_xr) * FT(1.0 / 3.0);
    }
    return p;
}

// ---- rain terminal velocity (CM2:685-719) from log2 λ; gated like the reference -----------------------------------------------
template <typename FT, bool LIMITED, int VEL, typename C>
__device__ __forceinline__ void sb2006_rain_velocity(const C &c, FT rho, FT rs_rho, FT l2_lam, typename Math<FT>::Mask no_N_rai,
                                                     typename Math<FT>::Mask no_q_rai, FT &vt_n, FT &vt_m) {
    using M = Math<FT>;
    using S = typename M::Scalar;
    vt_n = FT(0);
    vt_m = FT(0);
    if constexpr (VEL == VEL_SB) {   // CM2:685-702, helper :720-739
        const FT Dr_mean = M::exp2_fin(-l2_lam);          // l2_lam: finite combination of log2 of clamped positives
        FT pa0 = FT(1), pb0 = FT(1), pa1 = FT(1), pb1 = FT(1);
        if constexpr (!LIMITED) {
            const FT lam = M::exp2_fin(l2_lam);
            const FT ta = c.rc2 * lam, tb = c.rc2 * (lam + c.cR);
            pa0 = M::exp2_fin(ta * FT(-1.4426950408889634));
            pb0 = pa0 * c.e_rc2cR;
            pa1 = (((ta + FT(3)) * ta + FT(6)) * ta + FT(6)) * pa0 * FT(1.0 / 6.0);
            pb1 = (((tb + FT(3)) * tb + FT(6)) * tb + FT(6)) * pb0 * FT(1.0 / 6.0);
        }
        const FT s = c.vel_s * rs_rho;
        const FT inv_d1 = M::rcp_nz(M::fma(c.cR, Dr_mean, FT(1)));      // ≥ 1
        const FT inv_d2 = inv_d1 * inv_d1;
        const FT vt0 = M::max(FT(0), s * (c.aR * pa0 - c.bR * pb0 * inv_d1));
        const FT vt1 = M::max(FT(0), s * (c.aR * pa1 - c.bR * pb1 * (inv_d2 * inv_d2)));
        vt_n = no_N_rai ? FT(0) : vt0;
        vt_m = no_q_rai ? FT(0) : vt1;
    } else if constexpr (VEL == VEL_CHEN) {   // CM2:703-719, Common.jl:290-302, 414-422 — fitted coefficients (cmx_math.hpp ChenLog)
        const FT lam = M::exp2_fin(l2_lam);
        const FT rho_c = M::max(rho, FT(0));
        const FT t = rho_c - FT(0.5 * kChenGammaRhoMax);
        const FT l2_lam4 = FT(4) * l2_lam;
        FT vt0 = FT(0), vt3 = FT(0);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            // (wave-uniform shortcuts for the published table's c_1 = 0, c_2 = c_3 were tried: if-converted they cost three selects per point
            // and save nothing; as real scalar branches they split the four points' code into 24 blocks — 149 VGPRs instead of 124)
            const FT l2_den = M::log2(lam + c.ch_c1000[i]);     // log2(λ + 1000 c_i)
            FT L0 = chen_log_eval<S>(c.chl, 0, i, t), L3 = chen_log_eval<S>(c.chl, 1, i, t);
            if (i == 2) { const FT ar = c.ch_a3_pow * M::log2(rho_c); L0 += ar; L3 += ar; }      // ρ^a3_pow of the third term (log-singular at 0: not in the fit)
            const FT e0 = M::exp2(M::fma(-M::fma(-c.ch_b_rho, rho_c, c.chl.b1[i]), l2_den, L0 + l2_lam));
            const FT e3 = M::exp2(M::fma(-M::fma(-c.ch_b_rho, rho_c, c.chl.b4[i]), l2_den, L3 + l2_lam4));
            vt0 = M::fma(c.chl.sgn[i], e0, vt0);
            vt3 = M::fma(c.chl.sgn[i], e3, vt3);
        }
        vt0 = M::max(FT(0), vt0);
        vt3 = M::max(FT(0), vt3);
        {   // outside the range of the fit: no silent extrapolation
            const typename M::Mask beyond = rho_c > FT(kChenGammaRhoMax);
            vt0 = beyond ? M::nan() : vt0;
            vt3 = beyond ? M::nan() : vt3;
        }
        vt_n = no_N_rai ? FT(0) : vt0;
        vt_m = no_q_rai ? FT(0) : vt3;
    } else if constexpr (VEL == VEL_CHEN_GEN) {   // any table, any ρ: run-time Γ (OCML tgamma)
        const FT lam = M::exp2_fin(l2_lam);
        const FT rho_c = M::max(rho, FT(0));
        const FT l2_q = c.ch_rho0_l2e * rho_c;                  // log2 exp(ρ0 ρ)
        const FT l2_rho = M::log2(rho_c);
        const FT l2_lam_inv = -l2_lam;                          // log2 Dr_mean
        FT vt0 = FT(0), vt3 = FT(0);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const FT bi = M::fma(-c.ch_b_rho, rho_c, c.ch_b[i]);
            // aiu = a_i·q·(ρ^a3_pow for i = 3)·1000^b_i ; sign kept outside the log
            FT l2_mag = l2_q + bi * c.l2_1000 + (i == 2 ? c.ch_a3_pow * l2_rho : FT(0));
            const FT l2_den = M::log2(lam + c.ch_c1000[i]);     // log2(1/λ_inv + c)
            const FT g1 = tgamma_general<FT>(bi + FT(1));
            // k = 0: δ = 1;  k = 3: δ = 4, Γ(b+4) = (b+3)(b+2)(b+1)Γ(b+1), /3!
            const FT e0 = M::exp2(l2_mag - l2_lam_inv - (bi + FT(1)) * l2_den);
            const FT e3 = M::exp2(l2_mag - FT(4) * l2_lam_inv - (bi + FT(4)) * l2_den);
            vt0 = M::fma(c.ch_a[i] * e0, g1, vt0);
            const FT g4 = g1 * (bi + FT(3)) * (bi + FT(2)) * (bi + FT(1)) * FT(1.0 / 6.0);
            vt3 = M::fma(c.ch_a[i] * e3, g4, vt3);
        }
        vt0 = M::max(FT(0), vt0);
        vt3 = M::max(FT(0), vt3);
        vt_n = no_N_rai ? FT(0) : vt0;
        vt_m = no_q_rai ? FT(0) : vt3;
    }
}

// which Chen-2022 instantiation a parameter set takes (host): the fitted-Γ one, or the general one when the fit is not accurate
template <typename FT, typename CH> inline int chen_vel_kind(const CH &ch) {
    ChenLog<FT> g;
    return make_chen_log<FT>(ch, g) ? VEL_CHEN : VEL_CHEN_GEN;
}

// `n_lcl`, `n_rai` are per-kg numbers (BMT), `N_*` = ρ n_* per m³ (CM2).  No input clamping here:
// the fused entry clamps first (BMT:828-837), the per-process entry passes raw values like the
// reference's KA wrapper does.
// ICE: the 2M+P3 entry passes the ice content into the vapour budget and cp_m (BMT:942 → :731-744); the warm-only entry has q_ice ≡ 0.
// INTPOW: the three SB2006 exponents that are small integers in the published scheme — (1 − τᵃ)ᵇ with b = 3 (Eq. 6), (τ/(τ+τ₀))ᶜ with
// c = 4 (Eq. 8), (1 + κ_rr/Br)ᵈ with d = −5 (Eq. 11) — by multiplication instead of exp2(e·log2 x): exact integer powers, and in
// Float64 three table-driven log2 + exp2 pairs (≈ 45 instructions each) become 2–4 multiplies and one reciprocal.  The entry points set
// it when the parameter struct holds exactly these values (sb_integer_exponents), otherwise the general form runs.
template <typename WR> inline bool sb_integer_exponents(const WR &wr) {
    const auto &sb = wr.seifert_beheng;
    return sb.acnv.b == 3 && sb.accr.c == 4 && sb.self.d == -5;
}
template <typename FT, bool LIMITED, int VEL, bool ICE = false, bool INTPOW = false, typename C>
__device__ __forceinline__ SbRates<FT> sb2006_point(const C &c0, FT rho, FT T, FT q_tot,
                                                    FT q_lcl, FT q_rai, FT N_lcl, FT N_rai,
                                                    FT n_lcl, FT n_rai, FT q_ice = FT(0), FT cpm_qi = FT(0)) {
    using M = Math<FT>;
    const FT eps = FT(M::eps());  // ϵ_numerics_2M_M = ϵ_numerics_2M_N = eps(FT)  (Utilities.jl:325,332)
    SbRates<FT> r;

    const FT rs_rho = M::rsqrt(rho);      // ρ^(-1/2); each √(ρ0/ρ) is (host √ρ0)·rs_rho.  The FULL form: ρ arrives clamped with max0(), so 0 is possible (+Inf
                                          // in both float types; the positive-argument form gave NaN in Float64 only — two instructions)
    const FT inv_rho = rs_rho * rs_rho;   // one hardware transcendental for both
    r.inv_rho = inv_rho;

    // ---- thermodynamics: one p_sat(T) shared by cond/evap, S and G -----------------------------
    const C &c = c0;
    const FT inv_T = M::rcp_nz(T);                                           // a temperature: positive, finite
    const FT L_v = M::fma(c.dcp, T - c.T_0, c.LH_v0);                       // TD.latent_heat_vapor
    const FT l2_ps = M::fma(c.ps_a, M::log2(T * c.inv_T_tr), M::fma(c.ps_b, c.inv_T_tr - inv_T, c.ps_c0));
    const FT p_sat = M::exp2_fin(l2_ps);                                       // TD.saturation_vapor_pressure
    const FT q_liq = q_lcl + q_rai;
    FT q_vap = M::max(FT(0), q_tot - q_liq);                                 // TDI.q_vap (q_ice = q_sno = 0)
    if constexpr (ICE) q_vap = M::max(FT(0), (q_tot - q_liq) - q_ice);
    const FT rho_RvT = rho * (c.R_v * T);
    const FT inv_p_sat = M::rcp(p_sat);
    const FT inv_RvT = c.inv_R_v * inv_T;
    const FT q_sat = p_sat * (inv_rho * inv_RvT);                            // TD.q_vap_saturation
    const FT LoRT = L_v * inv_RvT;                                           // L/(R_v T)
    {   // _conv_q_vap_to_q_lcl_const  NonEq:117-140
        FT cp_air = M::fma(c.cpm_ql, q_liq, M::fma(c.cpm_qt, q_tot, c.cp_d));         // TD.cp_m
        if constexpr (ICE) cp_air = M::fma(cpm_qi, q_ice, cp_air);      // cpm_qi = cp_i − cp_v, passed by the 2M+P3 caller
        const FT dqsl_dT = q_sat * (LoRT * inv_T - inv_T);                   // dqcld_dT NonEq:74-76
        // gamma_helper NonEq:88-90: Γ = 1 + (L/cp)·dq/dT = (cp + L·dq/dT)/cp, so 1/(τΓ) needs one reciprocal, not two
        const FT excess = q_vap - q_sat;
        const FT inv_ts = cp_air * M::rcp_nz(c.tau_ce * M::fma(L_v, dqsl_dT, cp_air));
        const FT evap_lim = -M::min(-excess, M::max(FT(0), q_lcl));
        r.cond = (excess < FT(0) ? evap_lim : excess) * inv_ts;
    }
    const FT S = M::fma(q_vap * rho_RvT, inv_p_sat, FT(-1));                 // TDI.supersaturation_over_liquid
    // G_func_liquid  Common.jl:47-63
    const FT inv_p_safe = M::min(inv_p_sat, c.inv_eps_1m);                  // 1/max(p_sat, ϵ)
    const FT G = M::rcp_nz(M::fma(L_v * c.inv_K * inv_T, LoRT - FT(1), c.Rv_over_D * T * inv_p_safe));

    // ---- cloud side: autoconversion, cloud self-collection, accretion --------------------------
    const FT sq_lcl = M::max(q_lcl, eps);
    const FT sN_lcl = M::max(N_lcl, eps);
    const FT sq_rai = M::max(q_rai, eps);
    const FT sN_rai = M::max(N_rai, eps);
    const FT L_lcl = rho * sq_lcl;
    const FT L_rai = rho * sq_rai;
    const FT x_lcl_raw = L_lcl * M::rcp_nz(sN_lcl);                       // sN ≥ eps
    using B = typename M::Mask;
    const B no_q_lcl = q_lcl < eps, no_N_lcl = N_lcl < eps, no_q_rai = q_rai < eps;
    // τ = 1 − q_l/(q_l+q_r) (Eq. 5) in its cancellation-free form q_r/(q_l+q_r); 1−τ likewise.
    // With q_rai < eps the reference's two τ (max(0,q_r) in CM2:407 vs max(q_r,eps) in :450) differ,
    // but there ϕ_au ≡ 0 and accretion ≡ 0, so one τ serves both.
    const FT inv_qsum = M::rcp_nz(sq_lcl + sq_rai);                       // ≥ 2 eps
    const FT tau = sq_rai * inv_qsum;
    const FT one_m_tau = sq_lcl * inv_qsum;
    const FT l2_tau = M::log2(tau);
    {   // autoconversion CM2:396-427
        const C &c = consts_after(c0, CMX_PHASE_DEP(G, inv_T));
        const FT x_lcl = M::min(c.x_star, x_lcl_raw);
        const FT tau_a = M::exp2_fin(c.acnv_a * l2_tau);                  // τ ∈ [eps², 1]: l2_tau finite
        FT pow_b;
        if constexpr (INTPOW) { const FT u = FT(1) - tau_a; pow_b = u * u * u; }
        else pow_b = M::exp2(c.acnv_b * M::log2(FT(1) - tau_a));
        const FT phi_raw = keep(c.acnv_A * tau_a * pow_b);
        const FT phi_au = no_q_rai ? FT(0) : phi_raw;
        const FT u = (L_lcl * x_lcl) * c.sqrt_kfac;   // √(kcc/20/x*·ν-terms)·L·x̄: keeps L²x̄² inside the f32 range
        const FT inv_omt = M::rcp_nz(one_m_tau);                          // 1 − τ = sq_lcl/(sq_lcl + sq_rai) > 0
        const FT dL_rai = (u * u) * M::fma(phi_au, inv_omt * inv_omt, FT(1)) * (c.acnv_rho0 * inv_rho);
        const FT dN_rai = dL_rai * c.inv_x_star;
        const B gate = m_or(no_q_lcl, no_N_lcl);
        r.au_dq_rai = gate ? FT(0) : dL_rai * inv_rho;
        r.au_dq_lcl = -r.au_dq_rai;
        r.au_dN_rai = gate ? FT(0) : dN_rai;
        r.au_dN_lcl = FT(-2) * r.au_dN_rai;
    }
    {   // cloud_liquid_self_collection CM2:488-501 (raw L_lcl = ρ q_lcl)
        const C &c = consts_after(c0, CMX_PHASE_DEP(G, inv_T));
        const FT Lr = rho * q_lcl;
        const FT sc = -c.ksc * inv_rho * (Lr * Lr) - r.au_dN_lcl;
        r.lsc = no_q_lcl ? FT(0) : sc;
        // with q_lcl present but N_lcl absent autoconversion is gated to 0 and lsc = −k_sc/ρ·L² as well: one expression serves both
        r.lsc_plus_au = no_q_lcl ? FT(0) : -c.ksc * inv_rho * (Lr * Lr);
    }
    {   // accretion CM2:445-470
        const C &c = consts_after(c0, CMX_PHASE_DEP(G, inv_T));
        FT pow_c;
        if constexpr (INTPOW) { const FT t = tau * M::rcp_nz(tau + c.tau_0), t2 = t * t; pow_c = t2 * t2; }
        else pow_c = M::exp2_fin(c.accr_c * (l2_tau - M::log2(tau + c.tau_0)));
        const FT phi_ac = keep(pow_c);
        const FT k_ac = c.kcr_s * rs_rho * L_rai * phi_ac;
        const FT dq = keep(k_ac * L_lcl * inv_rho);          // dL_rai/ρ with dL_rai = kcr √(ρ0/ρ) L_lcl L_rai ϕ_ac
        const FT dN = keep(-k_ac * sN_lcl);                  // −dL_rai / x̄_c with x̄_c = L_lcl / N_lcl: the L_lcl of dL_rai cancels
        const B gate = m_or(no_q_lcl, no_q_rai, no_N_lcl);
        r.ac_dq_rai = gate ? FT(0) : dq;
        r.ac_dq_lcl = -r.ac_dq_rai;
        r.ac_dN_lcl = gate ? FT(0) : dN;
    }

    // ---- rain PSD parameters, once (CM2:67-110), from the safe values ----------
    const C &c_psd = consts_after(c0, CMX_PHASE_DEP(r.ac_dN_lcl, l2_tau));
    const SbRainPsd<FT> psd = sb2006_rain_psd<FT, LIMITED>(c_psd, L_rai, sN_rai);
    const FT l2_xr = psd.l2_xr, l2_lam = psd.l2_lam;
    const FT l2_Dr = (l2_xr + c_psd.l2_Drc) * FT(1.0 / 3.0);
    const FT Dr = M::exp2_fin(l2_Dr);                      // ∛(6 x̄_r/(π ρw)): CM2:588 and :809
    const B no_N_rai = N_rai < eps;
    {   // rain_self_collection CM2:545-560 + rain_breakup CM2:579-601
        const C &c = consts_after(c0, CMX_PHASE_DEP(r.ac_dN_lcl, l2_tau));
        // 1/Br = ∛(x̄_r/6) = Dr·∛(π ρw/36): κ_rr/Br = kappa_rr_K·Dr
        FT pw;
        if constexpr (INTPOW) { const FT r1 = M::rcp_nz(M::fma(c.kappa_rr_K, Dr, FT(1))), r2 = r1 * r1; pw = r2 * r2 * r1; }
        else pw = M::exp2_fin(c.self_d * M::log2(M::fma(c.kappa_rr_K, Dr, FT(1))));
        const FT sc = -c.krr_s * rs_rho * N_rai * L_rai * pw;
        const B gate = m_or(no_q_rai, no_N_rai);
        r.rsc = gate ? FT(0) : sc;
        const FT dD = Dr - c.Deq;
        const FT br_lin = keep(c.kbr * dD), br_exp = keep(M::exp2_fin(c.kappa_br_l2e * dD) - FT(1));
        const FT phi_br = (Dr < c.Dr_th) ? FT(-1) : ((Dr <= c.Deq) ? br_lin : br_exp);
        r.rbr = gate ? FT(0) : -(phi_br + FT(1)) * r.rsc;
    }
    {   // rain_evaporation CM2:780-828
        const C &c = consts_after(c0, CMX_PHASE_DEP(r.rbr, l2_xr));
        const FT l2_t = (c.l2_6xstar - l2_xr) * FT(1.0 / 3.0);            // t* = ∛(6 x*/x̄_r)
        // Float64 (round 5): t*·Dr is a constant of the parameter set, so t* is one Newton reciprocal of the Dr already at hand (≈ 7 instruction
        // slots) instead of a table-driven exponential (≈ 14); Float32 keeps the exponential (v_exp_f32 and v_rcp_f32 cost the same).  Dr > 0 and finite:
        // the exponential of a finite log2.  2⁻⁴⁸ relative on t*, which enters e^{−t*} (absolute exponent error t*·2⁻⁴⁸) and nothing else.
        FT t_star;
        if constexpr (M::IS_F64 && CMX_SB_TSTAR_RCP) t_star = c.tstar_Dr * M::rcp_nz(Dr);
        else t_star = M::exp2_fin(l2_t);
        // e^{−t*}/x̄_r in one exponential: both factors enter the number tendency only (Γ_incl does not appear in the mass one);
        // the ventilation coefficients a_vent_0, b_vent_0·∛Sc are folded into the Γ_incl denominators on the host
        const FT e_tx = M::exp2_fin(M::fma(t_star, FT(-1.4426950408889634), -l2_xr));   // t* ≤ 2^((l2_6x* + 1100)/3): finite
        const FT g_a = M::rcp_nz(M::fma(c.ga_c1, M::exp2_fin(c.ga_e1 * l2_t), c.ga_c2 * M::exp2_fin(c.ga_e2 * l2_t)));   // both coefficients > 0;   // a_vent_0·Γ_incl(−1, t*)·e^{t*}
        const FT g_b = M::rcp(M::fma(c.gb_c1, M::exp2_fin(c.gb_e1 * l2_t), c.gb_c2 * M::exp2_fin(c.gb_e2 * l2_t)));   // b_vent_0 ∛Sc·Γ_incl(β, t*)·e^{t*}
        // √N_Re = √(α/ν)·(ρ0/ρ)^¼·√(x̄^β·Dr)
        const FT sqrt_N_Re = c.sqrt_alpha_nu_rho0q * M::sqrt_pos(rs_rho) *
                             M::exp2_fin(FT(0.5) * M::fma(c.beta, l2_xr, l2_Dr));
        const FT Fv0 = M::fma(g_b, sqrt_N_Re, g_a);
        const FT Fv1 = M::fma(c.bSc_vent_1, sqrt_N_Re, c.a_vent_1);
        const FT common = c.two_pi * G * S * N_rai * Dr;
        const FT dN = M::min(FT(0), common * Fv0 * e_tx);
        const FT dq = M::min(FT(0), common * Fv1 * inv_rho);
        const B gate_q = m_or(no_q_rai, N_rai <= eps, S >= FT(0));
        const B gate_N = m_or(gate_q, l2_xr < c.l2_gate_N);                 // x̄_r/x* < eps(FT) — CM2:824-825
        r.evN = gate_N ? FT(0) : dN;
        r.evq = gate_q ? FT(0) : dq;
    }
    {   // number_tendency_from_mass_limits CM2:882-891 (cloud: xc_min/xc_max, rain: xr_min/xr_max)
        const C &c = consts_after(c0, CMX_PHASE_DEP(r.evq, Dr));
        const FT tl = no_q_lcl ? FT(0) : clampv(n_lcl, q_lcl * c.inv_xc_max, q_lcl * c.inv_xc_min);
        r.na_lcl = (tl - n_lcl) * c.inv_tau_na;
        const FT tr = no_q_rai ? FT(0) : clampv(n_rai, q_rai * c.inv_xr_max, q_rai * c.inv_xr_min);
        r.na_rai = (tr - n_rai) * c.inv_tau_na;
    }
    // ---- rain terminal velocity (optional columns) ----------------------------------------------
    sb2006_rain_velocity<FT, LIMITED, VEL>(consts_after(c0, CMX_PHASE_DEP(r.evq, Dr)), rho, rs_rho, l2_lam, no_N_rai, no_q_rai, r.vt_n, r.vt_m);
    return r;
}

}  // namespace cmx
