// cmx_arg_kernels.hip — Abdul-Razzak & Ghan (2000) aerosol activation, fused per thermodynamic state, for
// gfx950; C-ABI entry points of include/cmx.h §(6).
//
// Reference (src = /root/reference/src): AerosolActivation.jl — coeff_of_curvature :35-40,
// critical_supersaturation :107-118, max_supersaturation :138-200, N_activated_per_mode :235-259,
// M_activated_per_mode :294-321 (which evaluates critical_supersaturation and the thermodynamics twice per call).
//
// HBM-bound pointwise map: 4 (up to 8) state columns in, n_modes (+n_modes +1) columns out — 36 B/state for the
// BASELINE config (4 in, 5 modes out, f32).  The aerosol distribution is shared by all states, so everything that
// depends only on a mode (f_i, g_i, ln σ_i terms, N_i, hygroscopicity, r_dry) is folded on the host into per-mode
// constants and lives in SGPRs; per state the mode loop costs 2 exp2 + 1 log2 for the S_max sum and one erf
// (Float32: Abramowitz–Stegun 7.1.26 on v_exp/v_rcp, |ε| ≤ 1.5e-7 absolute — the result is N_i·½·erfc, compared
// against operands of size N_i) per requested output.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <type_traits>

#include "cmx_launch.hpp"
#include "cmx_math.hpp"

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
    FT l2_Ac_Ttr, l2_two_thirds, l2_two_pi_rho_w;   // log2(A_c/T_tr), log2(2/3), log2(2π ρw)
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
    c.l2_two_thirds = (FT)std::log2(2.0 / 3.0); c.l2_two_pi_rho_w = (FT)std::log2(2.0 * pi * (double)ap.rho_w);
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
    }
    return c;
}

// erfc for the activated NUMBER fractions.  Float32: A&S 7.1.26 (|ε| ≤ 1.5e-7); Float64: table-driven (below).
template <typename FT> __device__ __forceinline__ FT erfc_dev(FT x);
template <> __device__ __forceinline__ float erfc_dev<float>(float x) {
    using M = Math<float>;
    const float ax = __builtin_fabsf(x);
    const float t = M::rcp(M::fma(0.3275911f, ax, 1.0f));
    const float poly = t * M::fma(t, M::fma(t, M::fma(t, M::fma(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
    const float e = poly * M::exp2(-(ax * ax) * 1.4426950408889634f);
    return x >= 0.0f ? e : 2.0f - e;
}
#ifndef CMX_ARG_LEAN_ERFC
#define CMX_ARG_LEAN_ERFC 1      // 0: OCML erfc (A/B switch)
#endif
// Float64: the table-driven lean::erfc (cmx_lean_f64.hpp: relative error ≤ (2 + x²)·2e-16 up to 6.5; the reference's ½(1 − erf u) is
// exactly 0 beyond u = 5.9) — 45 instructions against OCML's 135, five calls per state
template <> __device__ __forceinline__ double erfc_dev<double>(double x) { return CMX_ARG_LEAN_ERFC ? lean::erfc(x) : ::erfc(x); }
// erfc with RELATIVE accuracy, for the activated mass: the reference evaluates M_act with erfc itself (AA:319), so a small activated
// fraction keeps its leading digits there (N_act is ½(1 − erf u), AA:257: absolute accuracy in the reference too, A&S above is its
// match).  Float32: t·exp(−x² + P(t)), t = 1/(1 + x/2) (Chebyshev fit of Numerical Recipes' erfcc, fractional error < 1.2e-7; the
// exponent is formed in Float32, so the relative error grows like 6e-8·x²: 5e-6 at erfc = 1e-30).  Float64: OCML.
template <typename FT> __device__ __forceinline__ FT erfc_rel_dev(FT x);
template <> __device__ __forceinline__ float erfc_rel_dev<float>(float x) {
    using M = Math<float>;
    const float z = __builtin_fabsf(x);
    const float t = M::rcp(M::fma(0.5f, z, 1.0f));
    float p = 0.17087277f;
    p = M::fma(p, t, -0.82215223f);
    p = M::fma(p, t, 1.48851587f);
    p = M::fma(p, t, -1.13520398f);
    p = M::fma(p, t, 0.27886807f);
    p = M::fma(p, t, -0.18628806f);
    p = M::fma(p, t, 0.09678418f);
    p = M::fma(p, t, 0.37409196f);
    p = M::fma(p, t, 1.00002368f);
    p = M::fma(p, t, -1.26551223f);
    const float e = t * M::exp2(M::fma(-z, z, p) * 1.4426950408889634f);
    return x >= 0.0f ? e : 2.0f - e;
}
template <> __device__ __forceinline__ double erfc_rel_dev<double>(double x) { return ::erfc(x); }

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
__device__ __forceinline__ float arg_pow_m34(float y) { const float t = Math<float>::rsqrt(y); return t * Math<float>::sqrt(t); }
template <typename FT> struct ArgIO {
    const FT *T, *p, *w, *q_tot, *q_liq, *q_ice, *N_liq, *N_ice;
    FT *N_act[CMX_ARG_MAX_MODES], *M_act[CMX_ARG_MAX_MODES], *S_max;
    FT *N_tot, *M_tot;   // total_N_activated / total_M_activated: Σ over the modes, left to right like Julia's sum of the tuple (AA:355-433)
    bool want_N, want_M;
};

template <typename FT, int NM> struct ArgOut { FT smax; FT n[NM]; FT m[NM]; };

// ---- one thermodynamic state, in three stages shared by the two kernels: arg_pre (everything in front of the sum over the modes),
// the mode sum (kernel-specific: host-folded mode constants, or mode descriptors streamed from columns), arg_smax (S_max from the sum).
template <typename FT> struct ArgPre {
    FT T, p, inv_T, R_m, L_v, inv_cp, rho_air, l2_TT, dinvT, l2_pvs, inv_pvs, inv_G_liq, ratio, gamma, aw;
    FT l2_A15, zeta, X, E_f, E_g;
};
template <typename FT>
__device__ __forceinline__ ArgPre<FT> arg_pre(const ArgConsts<FT> &c, FT T, FT p, FT w, FT q_tot, FT q_liq, FT q_ice) {
    using M = Math<FT>;
    ArgPre<FT> s;
    s.T = T; s.p = p;
    const FT inv_T = M::rcp_nz(T);                  // temperature, pressure, R_m, cp_m: positive and finite
    // TD.gas_constant_air, cp_m, latent heat, air density, vapour pressures — AA:152-160
    const FT R_m = c.R_d * (FT(1) + (c.Rv_over_Rd - FT(1)) * q_tot - c.Rv_over_Rd * (q_liq + q_ice));
    const FT cp_m = M::fma(c.cpm_qi, q_ice, M::fma(c.cpm_ql, q_liq, M::fma(c.cpm_qt, q_tot, c.cp_d)));
    const FT L_v = M::fma(c.dcp_l, T - c.T_0, c.LH_v0);
    const FT inv_Rm = M::rcp_nz(R_m), inv_cp = M::rcp_nz(cp_m);
    const FT rho_air = p * inv_Rm * inv_T;
    const FT p_v = (q_tot - q_liq - q_ice) * rho_air * c.R_v * T;
    const FT l2_TT = M::log2(T * c.inv_T_tr), dinvT = c.inv_T_tr - inv_T;
    const FT l2_pvs = M::fma(c.psl_a, l2_TT, M::fma(c.psl_b, dinvT, c.ps_c0));
    const FT inv_pvs = M::exp2_fin(-l2_pvs);          // overflow → +Inf is still right (capped by inv_eps_1m below)
    const FT LoRT = L_v * c.inv_R_v * inv_T;
    // 1/G_liq = L/(K T)(L/(R_v T) − 1) + R_v T/(D max(p_vs, ϵ))  (Common.jl:47-63); 1/max(p_vs, ϵ) = min(1/p_vs, 1/ϵ).  Only the
    // reciprocal of G = G_liq/ρ_w enters S_max (αw/G), so G itself is formed only for the sink terms.
    const FT inv_G_liq = M::fma(L_v * c.inv_K * inv_T, LoRT - FT(1), c.Rv_over_D * T * M::min(inv_pvs, c.inv_eps_1m));
    const FT ratio = p_v * inv_pvs;
    const FT alpha = ratio * (LoRT * c.g * inv_cp * inv_T - c.g * inv_Rm * inv_T);                       // AA:164
    const FT gamma = M::fma(ratio * R_m * L_v, LoRT * inv_cp * M::rcp_nz(p), c.R_v * T * inv_pvs);       // AA:165
    const FT aw = alpha * w;
    const FT aw_over_G = aw * c.rho_w * inv_G_liq;
    // A = A_c/T (AA:35-40), ζ = ⅔ A √(αw/G) (AA:168) and X = (αw/G)^1.5/(2π ρw γ) (η_i = X/N_i) assembled in the log2 domain from
    // log2(T/T_tr) (already formed for p_vs), log2(αw/G) and log2 γ: 4 transcendentals instead of sqrt + 3 log2 + rcp
    const FT l2_awG = M::log2(aw_over_G);
    const FT l2_A = c.l2_Ac_Ttr - l2_TT;
    const FT l2_A15 = FT(1.5) * l2_A;
    const FT l2_zeta = c.l2_two_thirds + l2_A + FT(0.5) * l2_awG;
    s.zeta = M::exp2_fin(l2_zeta);                    // w > 0: finite; w ≤ 0: the reference gives NaN (ζ/η = 0/0, √ of a negative)
    const FT l2_X = M::fma(FT(1.5), l2_awG, -(c.l2_two_pi_rho_w + M::log2(gamma)));
    s.X = M::exp2_fin(l2_X);
    // Σ_i (1/Sm_i²)·[f_i (ζ/η_i)^p1 + g_i (Sm_i²/(η_i+3ζ))^p2] — AA:170-183.  Everything that depends only on the mode is
    // folded on the host (ArgModeConsts) or per state from the mode columns; per state three shared powers.
    // the two state-only factors of the sum, one exponential each: A⁻³ (ζ/X)^p1 and A^(3 p2 − 3)
    s.E_f = M::exp2_fin(M::fma(c.p1, l2_zeta - l2_X, FT(-2) * l2_A15));
    s.E_g = M::exp2_fin((FT(2) * c.p2 - FT(2)) * l2_A15);
    s.inv_T = inv_T; s.R_m = R_m; s.L_v = L_v; s.inv_cp = inv_cp; s.rho_air = rho_air; s.l2_TT = l2_TT; s.dinvT = dinvT; s.l2_pvs = l2_pvs;
    s.inv_pvs = inv_pvs; s.inv_G_liq = inv_G_liq; s.ratio = ratio; s.gamma = gamma; s.aw = aw; s.l2_A15 = l2_A15;
    return s;
}
// (η + 3ζ)^(−p2) of one mode
template <typename FT> __device__ __forceinline__ FT arg_pow_p2(const ArgConsts<FT> &c, FT y, bool p2_is_34) {
    using M = Math<FT>;
    if constexpr (sizeof(FT) == 8) {
        // Float64 with ARG2000's own exponent p2 = ¾ (a wave-uniform test): y^(−¾) = t·√t with t = 1/√y — a reciprocal square root and
        // a square root (hardware seed + Newton steps, ≈ 30 instructions) instead of a table-driven log2 and exp2 (≈ 45) per mode.
        // 3ζ + η_k > 0 for w > 0; w = 0 gives ζ/η = 0/0 = NaN in the reference as well, so no 0 / Inf cases to keep
        if (p2_is_34) return arg_pow_m34(y);
    }
    return M::exp2(-c.p2 * M::log2(y));
}
template <typename FT, bool SINKS>
__device__ __forceinline__ FT arg_smax(const ArgConsts<FT> &c, const ArgPre<FT> &s, FT sum1, FT sum2, FT q_liq, FT q_ice, FT N_liq, FT N_ice) {
    using M = Math<FT>;
    const FT T = s.T, p = s.p, inv_T = s.inv_T;
    const FT tmp = M::fma(s.E_g, sum2, s.E_f * sum1);
    const FT S_arg = M::rsqrt_pos(tmp);                                                                       // AA:185
    FT smax;
    if constexpr (SINKS) {   // liquid / ice sink correction — AA:187-197
        const FT L_s = M::fma(c.dcp_i, T - c.T_0, c.LH_s0);
        const FT l2_pvi = M::fma(c.psi_a, s.l2_TT, M::fma(c.psi_b, s.dinvT, c.ps_c0));
        const FT p_vi = M::exp2(l2_pvi), p_vs = M::exp2(s.l2_pvs);
        const FT G = M::rcp(s.inv_G_liq) * c.inv_rho_w;
        const FT r_liq = N_liq < c.eps_ft ? FT(0) : M::exp2(M::log2(s.rho_air * q_liq * M::rcp(N_liq) * c.inv_43pi_rho_w) * FT(1.0 / 3.0));
        const FT K_liq = c.four_pi * c.rho_w * N_liq * r_liq * G * s.gamma;
        const FT gamma_i = M::fma(s.ratio * s.R_m * s.L_v, L_s * c.inv_R_v * s.inv_cp * inv_T * M::rcp(p), c.R_v * T * s.inv_pvs);
        const FT r_ice = N_ice < c.eps_ft ? FT(0) : M::exp2(M::log2(s.rho_air * q_ice * M::rcp(N_ice) * c.inv_43pi_rho_i) * FT(1.0 / 3.0));
        const FT LoRT_s = L_s * c.inv_R_v * inv_T;
        const FT G_ice = M::rcp(M::fma(L_s * c.inv_K * inv_T, LoRT_s - FT(1), c.Rv_over_D * T * M::rcp(M::max(p_vi, c.eps_1m))));
        const FT xi = p_vs * M::rcp(p_vi);
        const FT K_ice = c.four_pi * N_ice * r_ice * G_ice * gamma_i;
        smax = S_arg * (s.aw - K_ice * (xi - FT(1))) * M::rcp(M::fma(M::fma(K_ice, xi, K_liq), S_arg, s.aw));
    } else {
        smax = S_arg;   // N_liq = N_ice = 0: K_liq = K_ice = 0 ⇒ S_max = S_max_ARG·αw/αw
    }
    return smax < FT(0) ? FT(0) : smax;   // AA:199 max(0, S_max) with Julia's NaN rule: a NaN from any input reaches every output below
}

// one thermodynamic state of the shared-distribution kernel.  NM = compile-time mode count (1…8)
template <typename FT, int NM, bool SINKS>
__device__ __forceinline__ ArgOut<FT, NM> arg_point(const ArgConsts<FT> &c, const ArgModeConsts<FT> *__restrict__ cm, FT T, FT p, FT w,
                                                    FT q_tot, FT q_liq, FT q_ice, FT N_liq, FT N_ice, bool want_N, bool want_M) {
    using M = Math<FT>;
    ArgOut<FT, NM> o;
    const ArgPre<FT> s = arg_pre<FT>(c, T, p, w, q_tot, q_liq, q_ice);
    // with the mode-only factors c1_i, c2_i the sum is  A⁻³ (ζ/X)^p1 Σ c1_i + A^(3p2 − 3) Σ c2_i (η_i + 3ζ)^(−p2): per mode one
    // multiply, one FMA, one log2, one exp2 and one accumulating FMA
    FT sum1 = FT(0), sum2 = FT(0);
#pragma unroll
    for (int k = 0; k < NM; ++k) sum1 += cm[k].c1;
    const bool p2_34 = sizeof(FT) == 8 && CMX_ARG_P2_ROOTS && c.p2 == FT(0.75);
#pragma unroll
    for (int k = 0; k < NM; ++k) sum2 = M::fma(cm[k].c2, arg_pow_p2<FT>(c, M::fma(FT(3), s.zeta, s.X * cm[k].inv_N), p2_34), sum2);
    const FT smax = arg_smax<FT, SINKS>(c, s, sum1, sum2, q_liq, q_ice, N_liq, N_ice);
    o.smax = smax;
    const FT dl0 = s.l2_A15 - M::log2(smax);                       // log2(Sm_i / S_max) = l2_sm_c + dl0
#pragma unroll
    for (int k = 0; k < NM; ++k) {
        const FT u = M::fma(cm[k].u_c, dl0, cm[k].uc_sm);       // AA:255   (= ln(sm/smax)/fac, AA:316)
        o.n[k] = want_N ? cm[k].half_N * erfc_dev<FT>(u) : FT(0);                  // N ½ (1 − erf u)      AA:257
        // Float64: one mode's erfc at a time — left alone the NM independent table-driven evaluations are interleaved and all their
        // LDS reads hoisted (422 VGPRs for 5 modes × 2 states: one wave per SIMD).  The asm pins the mode's result here and, with its
        // memory clobber, keeps the next mode's table reads behind it.
        if constexpr (sizeof(FT) == 8 && CMX_ARG_LEAN_ERFC) asm volatile("" : "+v"(o.n[k]) : : "memory");
        o.m[k] = want_M ? cm[k].half_M * erfc_rel_dev<FT>(u - cm[k].fac) : FT(0); // M/2 erfc(u − fac)    AA:319
    }
    return o;
}

// N_ONLY: the number-activation-only request (N_act columns, no M_act — the BASELINE configuration) as a compile-time fact; with
// the two wants as run-time flags the compiler keeps both erfc chains and their selects alive (1212 → 729 VALU per 4 points).
#ifndef CMX_ARG_BS
#define CMX_ARG_BS 128
#endif
constexpr int kArgBS = CMX_ARG_BS;   // lanes per workgroup of the activation kernel: 128 (same-box A/B, f32: 256 → 0.650 ms, 128 → 0.638, 512 → 0.637–0.655; f64 indifferent)
template <typename FT, int NM, bool SINKS, int VEC, bool N_ONLY = false>
__global__ __launch_bounds__(kArgBS) void arg_activation_kernel(const ArgConsts<FT> c, const ArgIO<FT> io, const int64_t nvec) {
    const int64_t i = (int64_t)blockIdx.x * kArgBS + threadIdx.x;
    FT T[VEC], p[VEC], w[VEC], qt[VEC], ql[VEC] = {}, qi[VEC] = {}, Nl[VEC] = {}, Ni[VEC] = {};
    if (i < nvec) {
        load_col<FT, VEC>(io.T, i, T); load_col<FT, VEC>(io.p, i, p); load_col<FT, VEC>(io.w, i, w); load_col<FT, VEC>(io.q_tot, i, qt);
        if (io.q_liq) load_col<FT, VEC>(io.q_liq, i, ql);
        if (io.q_ice) load_col<FT, VEC>(io.q_ice, i, qi);
        if constexpr (SINKS) {
            if (io.N_liq) load_col<FT, VEC>(io.N_liq, i, Nl);
            if (io.N_ice) load_col<FT, VEC>(io.N_ice, i, Ni);
        }
    }
    if constexpr (sizeof(FT) == 8 && CMX_ARG_LEAN_ERFC) lean::erfc_tab_fill();   // published by the barrier inside prepare()
    Math<FT>::prepare();   // Float64: exp2 / log2 tables → LDS while the loads fly (every lane of the workgroup reaches the barrier inside); no-op for Float32
    if (i >= nvec) return;
    FT sm[VEC], na[NM][VEC], ma[NM][VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        const ArgOut<FT, NM> o = arg_point<FT, NM, SINKS>(c, c.m, T[k], p[k], w[k], qt[k], ql[k], qi[k], Nl[k], Ni[k], N_ONLY ? true : io.want_N,
                                                         N_ONLY ? false : io.want_M);
        sm[k] = o.smax;
#pragma unroll
        for (int j = 0; j < NM; ++j) { na[j][k] = o.n[j]; ma[j][k] = o.m[j]; }
    }
    if (io.S_max) store_col<FT, VEC>(io.S_max, i, sm);
#pragma unroll
    for (int j = 0; j < NM; ++j) {
        if ((N_ONLY || io.want_N) && io.N_act[j]) store_col<FT, VEC>(io.N_act[j], i, na[j]);
        if constexpr (!N_ONLY) {
            if (io.want_M && io.M_act[j]) store_col<FT, VEC>(io.M_act[j], i, ma[j]);
        }
    }
    if (io.N_tot) {
        FT t[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            t[k] = na[0][k];
#pragma unroll
            for (int j = 1; j < NM; ++j) t[k] += na[j][k];
        }
        store_col<FT, VEC>(io.N_tot, i, t);
    }
    if constexpr (!N_ONLY) {
        if (io.M_tot) {
            FT t[VEC];
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                t[k] = ma[0][k];
#pragma unroll
                for (int j = 1; j < NM; ++j) t[k] += ma[j][k];
            }
            store_col<FT, VEC>(io.M_tot, i, t);
        }
    }
}

template <typename FT, int NM, bool SINKS>
static void launch_arg(const ArgConsts<FT> &c, const ArgIO<FT> &io0, int64_t n, const void *const *ptrs, int nptrs, hipStream_t s) {
    constexpr int VEC = Math<FT>::VEC;
    const uintptr_t mis0 = reinterpret_cast<uintptr_t>(io0.T) & 15u;
    bool same_mis = (mis0 % sizeof(FT)) == 0;
    for (int k = 0; k < nptrs; ++k)
        if (ptrs[k]) same_mis = same_mis && ((reinterpret_cast<uintptr_t>(ptrs[k]) & 15u) == mis0);
    auto off = [](auto *p, int64_t lo) { return p ? p + lo : p; };
    auto launch_range = [&](auto vec_tag, int64_t lo, int64_t count) {
        constexpr int V = decltype(vec_tag)::value;
        if (count <= 0) return;
        ArgIO<FT> io = io0;
        io.T += lo; io.p += lo; io.w += lo; io.q_tot += lo;
        io.q_liq = off(io.q_liq, lo); io.q_ice = off(io.q_ice, lo); io.N_liq = off(io.N_liq, lo); io.N_ice = off(io.N_ice, lo);
        io.S_max = off(io.S_max, lo); io.N_tot = off(io.N_tot, lo); io.M_tot = off(io.M_tot, lo);
        for (int j = 0; j < CMX_ARG_MAX_MODES; ++j) { io.N_act[j] = off(io.N_act[j], lo); io.M_act[j] = off(io.M_act[j], lo); }
        const int64_t nv = count / V;
        const dim3 grid((unsigned)((nv + kArgBS - 1) / kArgBS));
        if (io.want_N && !io.want_M) hipLaunchKernelGGL((arg_activation_kernel<FT, NM, SINKS, V, true>), grid, dim3(kArgBS), 0, s, c, io, nv);
        else hipLaunchKernelGGL((arg_activation_kernel<FT, NM, SINKS, V, false>), grid, dim3(kArgBS), 0, s, c, io, nv);
    };
    if (same_mis) {
        const int64_t head = std::min<int64_t>(n, mis0 ? (int64_t)((16 - mis0) / sizeof(FT)) : 0);
        const int64_t body = ((n - head) / VEC) * VEC;
        launch_range(std::integral_constant<int, 1>{}, 0, head);
        launch_range(std::integral_constant<int, VEC>{}, head, body);
        launch_range(std::integral_constant<int, 1>{}, head + body, n - head - body);
    } else {
        launch_range(std::integral_constant<int, 1>{}, 0, n);
    }
}

template <typename FT, typename AP, typename AD, typename AI, typename TH>
static int32_t arg_entry(const AP *ap, const AD *ad, const AI *aip, const TH *tps, int64_t n, const FT *T, const FT *p,
                         const FT *w, const FT *q_tot, const FT *q_liq, const FT *q_ice, const FT *N_liq, const FT *N_ice,
                         FT *const *N_act, FT *const *M_act, FT *S_max, void *stream, FT *N_tot = nullptr, FT *M_tot = nullptr) {
    if (!ap || !ad || !aip || !tps || n < 0 || ad->n_modes < 1 || ad->n_modes > CMX_ARG_MAX_MODES) return CMX_ERR_BAD_ARG;
    if (n > kMaxPoints) return CMX_ERR_UNSUPPORTED;      // one launch cannot express the grid (cmx_launch.hpp)
    if (n == 0) return CMX_OK;
    if (!T || !p || !w || !q_tot) return CMX_ERR_BAD_ARG;
    const ArgConsts<FT> c = make_arg_consts<FT>(*ap, *ad, *aip, *tps);
    ArgIO<FT> io{};
    io.T = T; io.p = p; io.w = w; io.q_tot = q_tot; io.q_liq = q_liq; io.q_ice = q_ice; io.N_liq = N_liq; io.N_ice = N_ice;
    io.S_max = S_max; io.want_N = N_act != nullptr || N_tot != nullptr; io.want_M = M_act != nullptr || M_tot != nullptr;
    io.N_tot = N_tot; io.M_tot = M_tot;
    const void *ptrs[8 + 2 * CMX_ARG_MAX_MODES + 3] = {T, p, w, q_tot, q_liq, q_ice, N_liq, N_ice, S_max, N_tot, M_tot};
    int np = 11;
    for (int j = 0; j < ad->n_modes; ++j) {
        io.N_act[j] = N_act ? N_act[j] : nullptr;
        io.M_act[j] = M_act ? M_act[j] : nullptr;
        ptrs[np++] = io.N_act[j];
        ptrs[np++] = io.M_act[j];
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bool sinks = N_liq || N_ice;
#define CMX_ARG_CASE(NM)                                                             \
    case NM:                                                                         \
        if (sinks) launch_arg<FT, NM, true>(c, io, n, ptrs, np, s);                  \
        else launch_arg<FT, NM, false>(c, io, n, ptrs, np, s);                       \
        break;
    switch (ad->n_modes) {
        CMX_ARG_CASE(1) CMX_ARG_CASE(2) CMX_ARG_CASE(3) CMX_ARG_CASE(4)
        CMX_ARG_CASE(5) CMX_ARG_CASE(6) CMX_ARG_CASE(7) CMX_ARG_CASE(8)
        default: return CMX_ERR_BAD_ARG;
    }
#undef CMX_ARG_CASE
    CMX_HIP_TRY(hipGetLastError());
    return CMX_OK;
}

// ---- aerosol that varies in space: mode descriptors are columns (test/gpu_tests.jl:45-79) --------------------------------
// The reference's own GPU test builds one AerosolDistribution per element.  Per state the kernel streams the modes TWICE through
// registers instead of holding every mode's constants (round 3: 17 values per mode alive across the whole point function — 414
// VGPRs and 322 spilled SGPRs for 8 Float64 modes, one wave per SIMD):
//   pass 1  per mode: load (r_dry, σ, N, hygroscopicity), form the mode's two terms of the S_max sum (AA:170-183) in the log2 domain —
//           c1 = Sm_c⁻² f N^p1 as ONE exponential, c2 = Sm_c⁻² g Sm_c^(2 p2) as one — and accumulate; keep (ln σ, log2 Sm_c, N [, ΣM w]);
//   S_max   arg_smax (the same function as the shared-distribution kernel);
//   pass 2  per mode: u, ½ N erfc(u) [, ½ M erfc(u − fac)] → store.
// 3 (4) kept values per mode and point; a lane owns VEC consecutive points of every column (16-byte loads / stores in Float32).
template <typename FT> struct ArgColIO {
    const FT *r_dry[CMX_ARG_MAX_MODES], *stdev[CMX_ARG_MAX_MODES], *N[CMX_ARG_MAX_MODES], *hyg[CMX_ARG_MAX_MODES], *mmix[CMX_ARG_MAX_MODES];
};
template <typename FT> struct ArgColPar { FT l2_f1, f2_l2e, g1, g2; };   // log2 f1, f2·log2 e, g1, g2 of the ARG fit (f_i = f1 exp(f2 ln²σ), g_i = g1 + g2 ln σ)

// lanes per workgroup, per float type (same-box A/B, round 4, ms per 1e8 states of 5 modes: Float32 64 lanes 1.967, 128 lanes 2.11, 256 lanes 2.085;
// Float64 5.715 / 5.43 / 5.36 — profiles/r04_ab_sessions.txt, session 17).  -DCMX_ARGCOL_BS=n forces one size for both (A/B switch).
#ifdef CMX_ARGCOL_BS
template <typename FT> constexpr int kArgColBS = CMX_ARGCOL_BS;
#else
template <typename FT> constexpr int kArgColBS = sizeof(FT) == 4 ? 64 : 256;
#endif
template <typename FT, int NM, bool SINKS, int VEC, bool N_ONLY>
__global__ __launch_bounds__(kArgColBS<FT>) void arg_activation_columns_kernel(const ArgConsts<FT> c, const ArgColPar<FT> par, const ArgIO<FT> io,
                                                                           const ArgColIO<FT> mc, const int64_t nvec) {
    using M = Math<FT>;
    const int64_t i = (int64_t)blockIdx.x * kArgColBS<FT> + threadIdx.x;
    FT T[VEC], p[VEC], w[VEC], qt[VEC], ql[VEC] = {}, qi[VEC] = {}, Nl[VEC] = {}, Ni[VEC] = {};
    if (i < nvec) {
        load_col<FT, VEC>(io.T, i, T); load_col<FT, VEC>(io.p, i, p); load_col<FT, VEC>(io.w, i, w); load_col<FT, VEC>(io.q_tot, i, qt);
        if (io.q_liq) load_col<FT, VEC>(io.q_liq, i, ql);
        if (io.q_ice) load_col<FT, VEC>(io.q_ice, i, qi);
        if constexpr (SINKS) {
            if (io.N_liq) load_col<FT, VEC>(io.N_liq, i, Nl);
            if (io.N_ice) load_col<FT, VEC>(io.N_ice, i, Ni);
        }
    }
    if constexpr (sizeof(FT) == 8 && CMX_ARG_LEAN_ERFC) lean::erfc_tab_fill();   // published by the barrier inside prepare()
    Math<FT>::prepare();   // Float64: exp2 / log2 tables → LDS while the loads fly; no-op for Float32
    if (i >= nvec) return;
    const FT ln2 = FT(0.6931471805599453);
    const bool p2_34 = sizeof(FT) == 8 && CMX_ARG_P2_ROOTS && c.p2 == FT(0.75);
    ArgPre<FT> pre[VEC];
    FT sum1[VEC], sum2[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        pre[v] = arg_pre<FT>(c, T[v], p[v], w[v], qt[v], ql[v], qi[v]);
        sum1[v] = FT(0); sum2[v] = FT(0);
    }
    FT k_ls[NM][VEC], k_l2sm[NM][VEC], k_N[NM][VEC], k_mm[N_ONLY ? 1 : NM][VEC];
    // ---- pass 1: the S_max sum
#pragma unroll
    for (int k = 0; k < NM; ++k) {
        FT r[VEC], sd[VEC], Nk[VEC], hy[VEC];
        load_col<FT, VEC>(mc.r_dry[k], i, r); load_col<FT, VEC>(mc.stdev[k], i, sd); load_col<FT, VEC>(mc.N[k], i, Nk); load_col<FT, VEC>(mc.hyg[k], i, hy);
        if constexpr (!N_ONLY) {
            if (mc.mmix[k]) load_col<FT, VEC>(mc.mmix[k], i, k_mm[k]);
            else {
#pragma unroll
                for (int v = 0; v < VEC; ++v) k_mm[k][v] = FT(0);
            }
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const FT ls = M::log2(sd[v]) * ln2;                                           // ln σ
            // log2 Sm_c = log2(2/√B) − 1.5 log2(3 r_dry) (Sm = Sm_c A^1.5, AA:107-118)
            FT l2sm;
            const FT t3 = FT(3) * r[v];
            if constexpr (sizeof(FT) == 8) l2sm = FT(1) - FT(0.5) * M::log2(hy[v] * (t3 * t3 * t3));      // one table-driven log2; (3r)³ B ≥ 1e-40 is far inside the Float64 range
            else l2sm = M::fma(FT(-1.5), M::log2(t3), M::fma(FT(-0.5), M::log2(hy[v]), FT(1)));          // Float32: (3r)³ would underflow for r < 1e-13
            const FT l2N = M::log2(Nk[v]);
            // c1 = Sm_c⁻² f1 exp(f2 ln²σ) N^p1,  c2 = Sm_c⁻² (g1 + g2 ln σ) Sm_c^(2 p2)
            const FT c1 = M::exp2(M::fma(c.p1, l2N, M::fma(par.f2_l2e * ls, ls, M::fma(FT(-2), l2sm, par.l2_f1))));
            const FT c2 = M::fma(par.g2, ls, par.g1) * M::exp2((FT(2) * c.p2 - FT(2)) * l2sm);
            sum1[v] += c1;
            const FT y = M::fma(pre[v].X, M::rcp(Nk[v]), FT(3) * pre[v].zeta);              // η_k + 3ζ
            sum2[v] = M::fma(c2, arg_pow_p2<FT>(c, y, p2_34), sum2[v]);
            k_ls[k][v] = ls; k_l2sm[k][v] = l2sm; k_N[k][v] = Nk[v];
        }
        // one mode at a time (Float64: keeps the next mode's table reads and loads behind this mode's arithmetic, like arg_point's erfc loop)
        if constexpr (sizeof(FT) == 8) asm volatile("" : "+v"(sum2[0]) : : "memory");
    }
    // ---- S_max
    FT dl0[VEC], sm[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        sm[v] = arg_smax<FT, SINKS>(c, pre[v], sum1[v], sum2[v], ql[v], qi[v], Nl[v], Ni[v]);
        dl0[v] = pre[v].l2_A15 - M::log2(sm[v]);                                         // log2(Sm_k / S_max) = log2 Sm_c + dl0
    }
    if (io.S_max) store_col<FT, VEC>(io.S_max, i, sm);
    // ---- pass 2: activated number (and mass) per mode
#pragma unroll
    for (int k = 0; k < NM; ++k) {
        FT na[VEC], ma[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const FT ls = k_ls[k][v];
            const FT u = FT(0.3267527144895157) * M::rcp(ls) * (k_l2sm[k][v] + dl0[v]);   // 2 ln2/(3√2 ln σ) · log2(Sm_k/S_max)   AA:255
            na[v] = FT(0.5) * k_N[k][v] * erfc_dev<FT>(u);                                  // N ½ (1 − erf u)   AA:257
            if constexpr (sizeof(FT) == 8 && CMX_ARG_LEAN_ERFC) asm volatile("" : "+v"(na[v]) : : "memory");
            if constexpr (!N_ONLY) ma[v] = FT(0.5) * k_mm[k][v] * erfc_rel_dev<FT>(u - FT(2.121320343559643) * ls);   // M/2 erfc(u − 3 ln σ √2/2)   AA:319
        }
        if (io.N_act[k]) store_col<FT, VEC>(io.N_act[k], i, na);
        if constexpr (!N_ONLY) {
            if (io.M_act[k]) store_col<FT, VEC>(io.M_act[k], i, ma);
        }
    }
}

template <typename FT, int NM, bool SINKS>
static void launch_arg_columns(const ArgConsts<FT> &c, const ArgColPar<FT> &par, const ArgIO<FT> &io0, const ArgColIO<FT> &mc0, int64_t n,
                               const void *const *ptrs, int nptrs, hipStream_t s) {
    // Float64 runs one point per lane: the kernel is VALU-bound there and the second point's kept values would cost a wave per SIMD
    constexpr int VEC = sizeof(FT) == 4 ? Math<FT>::VEC : 1;
    const uintptr_t mis0 = reinterpret_cast<uintptr_t>(io0.T) & 15u;
    bool same_mis = (mis0 % sizeof(FT)) == 0;
    for (int k = 0; k < nptrs; ++k)
        if (ptrs[k]) same_mis = same_mis && ((reinterpret_cast<uintptr_t>(ptrs[k]) & 15u) == mis0);
    auto off = [](auto *p, int64_t lo) { return p ? p + lo : p; };
    auto launch_range = [&](auto vec_tag, int64_t lo, int64_t count) {
        constexpr int V = decltype(vec_tag)::value;
        if (count <= 0) return;
        ArgIO<FT> io = io0;
        ArgColIO<FT> mc = mc0;
        io.T += lo; io.p += lo; io.w += lo; io.q_tot += lo;
        io.q_liq = off(io.q_liq, lo); io.q_ice = off(io.q_ice, lo); io.N_liq = off(io.N_liq, lo); io.N_ice = off(io.N_ice, lo);
        io.S_max = off(io.S_max, lo);
        for (int j = 0; j < NM; ++j) {
            io.N_act[j] = off(io.N_act[j], lo); io.M_act[j] = off(io.M_act[j], lo);
            mc.r_dry[j] += lo; mc.stdev[j] += lo; mc.N[j] += lo; mc.hyg[j] += lo; mc.mmix[j] = off(mc.mmix[j], lo);
        }
        const int64_t nv = count / V;
        const dim3 grid((unsigned)((nv + kArgColBS<FT> - 1) / kArgColBS<FT>));
        if (!io.want_M) hipLaunchKernelGGL((arg_activation_columns_kernel<FT, NM, SINKS, V, true>), grid, dim3(kArgColBS<FT>), 0, s, c, par, io, mc, nv);
        else hipLaunchKernelGGL((arg_activation_columns_kernel<FT, NM, SINKS, V, false>), grid, dim3(kArgColBS<FT>), 0, s, c, par, io, mc, nv);
    };
    if (same_mis && VEC > 1) {
        const int64_t head = std::min<int64_t>(n, mis0 ? (int64_t)((16 - mis0) / sizeof(FT)) : 0);
        const int64_t body = ((n - head) / VEC) * VEC;
        launch_range(std::integral_constant<int, 1>{}, 0, head);
        launch_range(std::integral_constant<int, VEC>{}, head, body);
        launch_range(std::integral_constant<int, 1>{}, head + body, n - head - body);
    } else {
        launch_range(std::integral_constant<int, 1>{}, 0, n);
    }
}

template <typename FT, typename AP, typename AI, typename TH>
static int32_t arg_columns_entry(const AP *ap, const AI *aip, const TH *tps, int32_t n_modes, int64_t n, const FT *T, const FT *p,
                                 const FT *w, const FT *q_tot, const FT *q_liq, const FT *q_ice, const FT *N_liq, const FT *N_ice,
                                 const FT *const *r_dry, const FT *const *stdev, const FT *const *N_mode, const FT *const *hyg,
                                 const FT *const *mmix, FT *const *N_act, FT *const *M_act, FT *S_max, void *stream) {
    if (!ap || !aip || !tps || n < 0 || n_modes < 1 || n_modes > CMX_ARG_MAX_MODES) return CMX_ERR_BAD_ARG;
    if (n > kMaxPoints) return CMX_ERR_UNSUPPORTED;      // one launch cannot express the grid (cmx_launch.hpp)
    if (n == 0) return CMX_OK;
    if (!T || !p || !w || !q_tot || !r_dry || !stdev || !N_mode || !hyg) return CMX_ERR_BAD_ARG;
    if (M_act && !mmix) return CMX_ERR_BAD_ARG;
    // thermodynamic / parameter constants from the shared builder with an empty distribution
    struct EmptyDist { int32_t n_modes = 0; struct Mode { FT r_dry, stdev, N, hygroscopicity, molar_mass_mix; } modes[1]; } none;
    const ArgConsts<FT> c = make_arg_consts<FT>(*ap, none, *aip, *tps);
    ArgIO<FT> io{};
    io.T = T; io.p = p; io.w = w; io.q_tot = q_tot; io.q_liq = q_liq; io.q_ice = q_ice; io.N_liq = N_liq; io.N_ice = N_ice;
    io.S_max = S_max; io.want_N = N_act != nullptr; io.want_M = M_act != nullptr;
    ArgColIO<FT> mc{};
    const void *ptrs[9 + 7 * CMX_ARG_MAX_MODES] = {T, p, w, q_tot, q_liq, q_ice, N_liq, N_ice, S_max};
    int np = 9;
    for (int j = 0; j < n_modes; ++j) {
        if (!r_dry[j] || !stdev[j] || !N_mode[j] || !hyg[j]) return CMX_ERR_BAD_ARG;
        mc.r_dry[j] = r_dry[j]; mc.stdev[j] = stdev[j]; mc.N[j] = N_mode[j]; mc.hyg[j] = hyg[j]; mc.mmix[j] = mmix ? mmix[j] : nullptr;
        io.N_act[j] = N_act ? N_act[j] : nullptr;
        io.M_act[j] = M_act ? M_act[j] : nullptr;
        const void *q[7] = {mc.r_dry[j], mc.stdev[j], mc.N[j], mc.hyg[j], mc.mmix[j], io.N_act[j], io.M_act[j]};
        for (const void *x : q) ptrs[np++] = x;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bool sinks = N_liq || N_ice;
    const double l2e = 1.4426950408889634074;
    const ArgColPar<FT> par{(FT)std::log2((double)ap->f1), (FT)((double)ap->f2 * l2e), (FT)ap->g1, (FT)ap->g2};
#define CMX_ARGC_CASE(NM)                                                                     \
    case NM:                                                                                  \
        if (sinks) launch_arg_columns<FT, NM, true>(c, par, io, mc, n, ptrs, np, s);          \
        else launch_arg_columns<FT, NM, false>(c, par, io, mc, n, ptrs, np, s);               \
        break;
    switch (n_modes) {
        CMX_ARGC_CASE(1) CMX_ARGC_CASE(2) CMX_ARGC_CASE(3) CMX_ARGC_CASE(4)
        CMX_ARGC_CASE(5) CMX_ARGC_CASE(6) CMX_ARGC_CASE(7) CMX_ARGC_CASE(8)
        default: return CMX_ERR_BAD_ARG;
    }
#undef CMX_ARGC_CASE
    CMX_HIP_TRY(hipGetLastError());
    return CMX_OK;
}

}  // namespace cmx

extern "C" {

int32_t cmx_arg2000_activation_f32(const cmx_aerosol_activation_params_f32 *ap, const cmx_aerosol_distribution_f32 *ad,
                                   const cmx_air_properties_f32 *aip, const cmx_thermo_f32 *tps, int64_t n, const float *T,
                                   const float *p, const float *w, const float *q_tot, const float *q_liq, const float *q_ice,
                                   const float *N_liq, const float *N_ice, float *const *N_act, float *const *M_act,
                                   float *S_max, void *stream) {
    return cmx::arg_entry<float>(ap, ad, aip, tps, n, T, p, w, q_tot, q_liq, q_ice, N_liq, N_ice, N_act, M_act, S_max, stream);
}
int32_t cmx_arg2000_activation_f64(const cmx_aerosol_activation_params_f64 *ap, const cmx_aerosol_distribution_f64 *ad,
                                   const cmx_air_properties_f64 *aip, const cmx_thermo_f64 *tps, int64_t n, const double *T,
                                   const double *p, const double *w, const double *q_tot, const double *q_liq,
                                   const double *q_ice, const double *N_liq, const double *N_ice, double *const *N_act,
                                   double *const *M_act, double *S_max, void *stream) {
    return cmx::arg_entry<double>(ap, ad, aip, tps, n, T, p, w, q_tot, q_liq, q_ice, N_liq, N_ice, N_act, M_act, S_max, stream);
}

int32_t cmx_arg2000_activation_columns_f32(const cmx_aerosol_activation_params_f32 *ap, const cmx_air_properties_f32 *aip,
                                           const cmx_thermo_f32 *tps, int32_t n_modes, int64_t n, const float *T, const float *p,
                                           const float *w, const float *q_tot, const float *q_liq, const float *q_ice, const float *N_liq,
                                           const float *N_ice, const float *const *r_dry, const float *const *stdev,
                                           const float *const *N_mode, const float *const *hygroscopicity, const float *const *molar_mass,
                                           float *const *N_act, float *const *M_act, float *S_max, void *stream) {
    return cmx::arg_columns_entry<float>(ap, aip, tps, n_modes, n, T, p, w, q_tot, q_liq, q_ice, N_liq, N_ice, r_dry, stdev, N_mode,
                                         hygroscopicity, molar_mass, N_act, M_act, S_max, stream);
}
int32_t cmx_arg2000_activation_columns_f64(const cmx_aerosol_activation_params_f64 *ap, const cmx_air_properties_f64 *aip,
                                           const cmx_thermo_f64 *tps, int32_t n_modes, int64_t n, const double *T, const double *p,
                                           const double *w, const double *q_tot, const double *q_liq, const double *q_ice,
                                           const double *N_liq, const double *N_ice, const double *const *r_dry,
                                           const double *const *stdev, const double *const *N_mode, const double *const *hygroscopicity,
                                           const double *const *molar_mass, double *const *N_act, double *const *M_act, double *S_max,
                                           void *stream) {
    return cmx::arg_columns_entry<double>(ap, aip, tps, n_modes, n, T, p, w, q_tot, q_liq, q_ice, N_liq, N_ice, r_dry, stdev, N_mode,
                                          hygroscopicity, molar_mass, N_act, M_act, S_max, stream);
}

// AA.total_N_activated / total_M_activated — src/AerosolActivation.jl:355-433: Σ over the modes, formed in the activation kernel itself
int32_t cmx_arg2000_total_activated_f32(const cmx_aerosol_activation_params_f32 *ap, const cmx_aerosol_distribution_f32 *ad,
                                        const cmx_air_properties_f32 *aip, const cmx_thermo_f32 *tps, int64_t n, const float *T, const float *p,
                                        const float *w, const float *q_tot, const float *q_liq, const float *q_ice, const float *N_liq,
                                        const float *N_ice, float *N_total, float *M_total, void *stream) {
    if (!N_total && !M_total) return CMX_ERR_BAD_ARG;
    return cmx::arg_entry<float>(ap, ad, aip, tps, n, T, p, w, q_tot, q_liq, q_ice, N_liq, N_ice, nullptr, nullptr, nullptr, stream, N_total, M_total);
}
int32_t cmx_arg2000_total_activated_f64(const cmx_aerosol_activation_params_f64 *ap, const cmx_aerosol_distribution_f64 *ad,
                                        const cmx_air_properties_f64 *aip, const cmx_thermo_f64 *tps, int64_t n, const double *T, const double *p,
                                        const double *w, const double *q_tot, const double *q_liq, const double *q_ice, const double *N_liq,
                                        const double *N_ice, double *N_total, double *M_total, void *stream) {
    if (!N_total && !M_total) return CMX_ERR_BAD_ARG;
    return cmx::arg_entry<double>(ap, ad, aip, tps, n, T, p, w, q_tot, q_liq, q_ice, N_liq, N_ice, nullptr, nullptr, nullptr, stream, N_total, M_total);
}

}  // extern "C"
