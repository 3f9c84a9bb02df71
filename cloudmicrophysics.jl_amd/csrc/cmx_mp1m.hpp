// cmx_mp1m.hpp — one-moment (Marshall–Palmer) scheme: host-folded constants and the per-point functions shared by the
// tendencies, LinearizedAverage, source-term, layout and column kernels (cmx_mp1m_kernels.hip, cmx_mp1m_column.hip).
//
// Reference (src = /root/reference/src): BulkMicrophysicsTendencies.jl:141-252 (`_microphysics_source_terms`,
// `_aggregate_tendencies`), :269-465 (`_linearize`, `_linearized_implicit_step`), Microphysics1M.jl (CM1),
// MicrophysicsNonEq.jl:32-58,104-224 (NonEq), Common.jl:47-102, 157-173.  13 option-dispatched processes, 18 source terms,
// 4 tendencies.
//
// Per point the two saturation pressures, the three Marshall–Palmer slope parameters λ⁻¹ and the rain v0 are evaluated ONCE
// (the reference recomputes them per process); every power of a λ⁻¹ comes from its one log2; parameter-only factors (Γ terms,
// a0·χa·χv·E…, r0 powers, products of two parameters) are folded on the host in double.  The reference's
// Microphysics1MOptions arrive as a flags word: a compile-time constant for the default option set (one straight-line basic
// block), an SGPR otherwise (disabled processes skipped by wave-uniform branches).
//
// Round 3 restructuring of the point function (VALU instructions per Float32 point 327 → 253): the logistic integrals
// without their three selects (max of the two branches), the limited saturation excess as one max / med3, the rain–snow kernel
// polynomial on host-folded coefficients and the already-formed squares of the slope parameters, the melt gates as max(T − T_freeze, 0),
// ρ R_v T inverted as a product of the two reciprocals the point needs anyway.
#pragma once
#include <algorithm>
#include <cmath>
#include <type_traits>

#include "../../include/cmx.h"
#include "cmx_math.hpp"

namespace cmx {

// Host-folded constants, ordered by the phase of the point function that reads them (Float64: only a phase's constants are live,
// cmx_math.hpp consts_after).
template <typename FT> struct Mp1mConsts {
    uint32_t flags;
    // ---- phase 1: thermodynamics and cloud formation
    FT eps_1m, inv_T_tr, ps_c0, psl_a, psl_b, psi_a, psi_b;
    FT T_0, LH_v0, LH_s0, LH_f0, dcp_l, dcp_i, dcp_f, R_v, inv_R_v, T_freeze;
    FT cp_d, cpm_qt, cpm_ql, cpm_qi;
    FT tau_l, tau_i, inv_tau_i;
    FT td_b10, td_l2a, td_c3, td_fourpiD, l2_eps;      // TemperatureDependent cloud-ice formation (NonEq:32-50)
    // ---- phase 2: supersaturations and G functions
    FT inv_K, Rv_over_D, inv_eps;
    // ---- phase 3: size distributions.  log2 λ⁻¹ = max(floor, a·log2(ρ q) + b [− a·log2 n0 for snow]); lamp_* = the same for the
    // root the default-exponent instantiation exponentiates (rain λ⁻¹^¼, snow λ⁻¹^⅛)
    FT lam_a_rai, lam_b_rai, lam_floor_rai, lam_a_icl, lam_b_icl, lam_floor_icl, lam_a_sno, lam_b_sno, lam_floor_sno;
    FT lamp_a_rai, lamp_b_rai, lamp_floor_rai, lamp_a_sno, lamp_b_sno, lamp_floor_sno;
    FT sno_l2_mu, sno_nu, rho_w;
    // ---- phase 4: autoconversion.  Kessler-type logistic integral (Common.jl:157-173) in the log2 domain: y2 = x·ka_y2,
    // rate = max(log2(ka_omemk + ka_emk·2^y2), y2 − ka_kl2e)·ka_out
    FT ka_qthr, ka_y2, ka_emk, ka_omemk, ka_kl2e, ka_out, ka_inv_tau;
    FT ks_qthr, ks_y2, ks_emk, ks_omemk, ks_kl2e, ks_out, ks_inv_tau;
    FT nd_coeff;                                     // PrescribedNd: 1/(τ (Nc/1e8)^α)
    FT r_is, r_is2_over_me, four_pi_n0_icl;          // WithSupersaturation
    // ---- phase 5: accretion.  rate = q_clo · k · √(ρw/ρ − 1) [rain] · n0 [snow] · λ⁻¹^e
    FT cv_l, acc_k_lcl_rai, acc_k_icl_rai, acc_e_rai, acc_k_lcl_sno, acc_k_icl_sno, acc_e_sno;
    FT sink_k, sink_e;                               // accretion_rain_sink
    // ---- phase 6: accretion_snow_rain.  v = vt_k · [√(ρw/ρ − 1)] · λ⁻¹^vt_e;  Σ/2 = λi² + q1·λi λj + q2·λj²
    FT vt_k_rai, vt_e_rai, vt_k_sno, vt_e_sno, coeff_disp;
    FT rs_k_rai, rs_dp1_rai, rs_q1_rai, rs_q2_rai, rs_k_sno, rs_dp1_sno, rs_q1_sno, rs_q2_sno;
    // ---- phase 7: ventilated vapour exchange and melting.  4π n0 F = ven_a + ven_b · [(ρw/ρ − 1)^¼] · λ⁻¹^e
    FT ven_a_rai, ven_b_rai, ven_e_rai, ven_a_sno, ven_b_sno, ven_e_sno;
    FT K_therm, mi_k;
};

// default slope-parameter exponents — rain: fall speed ½, accretion 3½, ice–rain sink 6½, snow–rain kernel 4 (δ = 3), ventilation ¾
// (all multiples of ¼); snow: ¼, 3¼, 3 (δ = 2), ⅝ (multiples of ⅛).  The eleven powers of the two λ⁻¹ are then products of ONE exp2
// each (r = λ⁻¹^¼, s = λ⁻¹^⅛) instead of eleven exp2 — in Float64 ≈ 180 instructions of a point.  mp1m_default_exponents() decides on
// the host; any other parameter set takes the run-time-flags kernels with the general exp2(e·log2 λ⁻¹) forms.
constexpr uint32_t kRuntimeFlags = 0xffffffffu;
constexpr uint32_t kDefExpBit = 0x40000000u;   // internal (not an ABI flag): set in the compile-time FLAGS of the default instantiation
static_assert((CMX_1M_DEFAULT_OPTIONS & kDefExpBit) == 0, "internal bit collides with an option flag");
#ifndef CMX_1M_DEFEXP
#define CMX_1M_DEFEXP 1      // A/B switch
#endif
template <typename CT> inline bool mp1m_default_exponents(const CT &c) {
    return CMX_1M_DEFEXP && c.vt_e_rai == 0.5 && c.acc_e_rai == 3.5 && c.sink_e == 6.5 && c.rs_dp1_rai == 4 && c.ven_e_rai == 0.75 && c.vt_e_sno == 0.25 &&
           c.acc_e_sno == 3.25 && c.rs_dp1_sno == 3 && c.ven_e_sno == 0.625 &&
           c.ka_qthr >= c.eps_1m && c.ks_qthr >= c.eps_1m;   // … and regular autoconversion thresholds (logistic_rate)
}

template <typename FT, typename MP, typename TH>
static Mp1mConsts<FT> make_mp1m_consts(const MP &mp, const TH &tp, uint32_t flags, double eps) {
    Mp1mConsts<FT> c{};
    const double l2e = 1.4426950408889634074, ln2 = 0.69314718055994530942, pi = 3.14159265358979323846;
    c.flags = flags;
    const double Rv = tp.R_v, T0 = tp.T_0;
    const double dcp_l = (double)tp.cp_v - (double)tp.cp_l, dcp_i = (double)tp.cp_v - (double)tp.cp_i;
    c.T_0 = (FT)T0; c.T_freeze = (FT)tp.T_freeze; c.LH_v0 = (FT)tp.LH_v0; c.LH_s0 = (FT)tp.LH_s0;
    c.LH_f0 = (FT)((double)tp.LH_s0 - (double)tp.LH_v0);
    c.dcp_l = (FT)dcp_l; c.dcp_i = (FT)dcp_i; c.dcp_f = (FT)((double)tp.cp_l - (double)tp.cp_i);
    c.R_v = (FT)Rv; c.inv_R_v = (FT)(1.0 / Rv); c.cv_l = (FT)tp.cv_l;
    c.ps_c0 = (FT)std::log2((double)tp.press_triple);
    c.psl_a = (FT)(dcp_l / Rv); c.psl_b = (FT)(((double)tp.LH_v0 - dcp_l * T0) / Rv * l2e);
    c.psi_a = (FT)(dcp_i / Rv); c.psi_b = (FT)(((double)tp.LH_s0 - dcp_i * T0) / Rv * l2e);
    c.inv_T_tr = (FT)(1.0 / (double)tp.T_triple);
    c.cp_d = (FT)tp.cp_d; c.cpm_qt = (FT)((double)tp.cp_v - (double)tp.cp_d);
    c.cpm_ql = (FT)((double)tp.cp_l - (double)tp.cp_v); c.cpm_qi = (FT)((double)tp.cp_i - (double)tp.cp_v);
    const auto &pp = mp.process_params;
    c.tau_l = (FT)pp.cloud_liquid_formation_tau_relax; c.tau_i = (FT)pp.cloud_ice_formation_tau_relax;
    c.inv_tau_i = (FT)(1.0 / (double)pp.cloud_ice_formation_tau_relax);
    const double K_safe = std::fmax((double)mp.air_properties.K_therm, eps), D_safe = std::fmax((double)mp.air_properties.D_vapor, eps);
    const double nu_air = mp.air_properties.nu_air;
    c.inv_K = (FT)(1.0 / K_safe); c.Rv_over_D = (FT)(Rv / D_safe); c.eps_1m = (FT)eps; c.l2_eps = (FT)std::log2(eps);
    c.inv_eps = (FT)(1.0 / eps);
    c.K_therm = (FT)mp.air_properties.K_therm;
    {   // τ_relax (NonEq:32-50): N = (−b T_c/10)⁹/a, r = ∛(3 q/(4π N ρᵢ)) ∨ 1e-6, τ_dep⁻¹ = 4π D_vapor N r
        const auto &fr = pp.cloud_ice_formation_frostenberg;
        c.td_b10 = (FT)((double)fr.b / 10.0); c.td_l2a = (FT)((double)fr.log_a * l2e);
        c.td_c3 = (FT)(3.0 / (4.0 * pi * (double)mp.cloud_ice.rho_i)); c.td_fourpiD = (FT)(4.0 * pi * (double)mp.air_properties.D_vapor);
    }
    // CM1.lambda_inverse :126-152 — λ⁻¹ = (ρ q r0^d / (χm m0 n0 Γ(d+1)))^(1/(d+1)) floored at r0·1e-5, d = me + Δm
    auto slope = [&](const auto &m, double n0, double root, FT &aa, FT &bb, FT &fl, FT *pa, FT *pb, FT *pf) {
        const double d = (double)m.me + (double)m.delta_m;
        const double denom_wo_n0 = (double)m.chi_m * (double)m.m0 * (double)m.gamma_coeff;
        const double cc = std::log2(std::pow((double)m.r0, d) / denom_wo_n0) - (n0 > 0 ? std::log2(std::fmax(n0, eps)) : 0.0);
        const double ee = 1.0 / (d + 1.0), floor_ = std::log2((double)m.r0 * 1e-5);
        aa = (FT)ee; bb = (FT)(cc * ee); fl = (FT)floor_;
        if (pa) { *pa = (FT)(ee * root); *pb = (FT)(cc * ee * root); *pf = (FT)(floor_ * root); }
    };
    slope(mp.rain.mass, (double)mp.rain.n0, 0.25, c.lam_a_rai, c.lam_b_rai, c.lam_floor_rai, &c.lamp_a_rai, &c.lamp_b_rai, &c.lamp_floor_rai);
    slope(mp.snow.mass, 0.0, 0.125, c.lam_a_sno, c.lam_b_sno, c.lam_floor_sno, &c.lamp_a_sno, &c.lamp_b_sno, &c.lamp_floor_sno);
    slope(mp.cloud_ice.mass, (double)mp.cloud_ice.n0, 1.0, c.lam_a_icl, c.lam_b_icl, c.lam_floor_icl, nullptr, nullptr, nullptr);
    c.sno_l2_mu = (FT)std::log2((double)mp.snow.mu); c.sno_nu = (FT)mp.snow.nu;
    const double n0_rai = mp.rain.n0, n0_icl = mp.cloud_ice.n0;
    const auto &vr = mp.vel_rain;
    const auto &vs = mp.vel_snow;
    // get_v0 :101-104: v0 = √(8/3/C_drag · g r0 (ρw/ρ − 1)) = v0c · √(ρw/ρ − 1);  snow: a parameter
    const double v0c = std::sqrt(8.0 / 3.0 / (double)vr.C_drag * (double)vr.grav * (double)vr.r0), v0s = vs.v0;
    c.rho_w = (FT)vr.rho_w;
    // terminal_velocity :223-238: χv v0 (λ⁻¹/r0)^(ve+Δv) Γ_term/Γ_coeff
    auto vt = [&](const auto &v, const auto &m, double v0, FT &kk, FT &ee) {
        const double p = (double)v.ve + (double)v.delta_v;
        kk = (FT)((double)v.chi_v * (double)v.gamma_term / (double)m.gamma_coeff * std::pow((double)m.r0, -p) * v0);
        ee = (FT)p;
    };
    vt(vr, mp.rain.mass, v0c, c.vt_k_rai, c.vt_e_rai);
    vt(vs, mp.snow.mass, v0s, c.vt_k_sno, c.vt_e_sno);
    // autoconversion
    auto logistic = [&](const auto &a, FT &qthr, FT &y2, FT &emk, FT &omemk, FT &kl2e, FT &out, FT &inv_tau) {
        const double x0 = std::fmax((double)a.q_threshold, (double)Math<FT>::eps_1m()), k = a.k;
        qthr = (FT)a.q_threshold; y2 = (FT)(k / x0 * l2e); emk = (FT)std::exp(-k); omemk = (FT)(-std::expm1(-k)); kl2e = (FT)(k * l2e);
        out = (FT)(ln2 * x0 / k / (double)a.tau); inv_tau = (FT)(1.0 / (double)a.tau);
    };
    logistic(pp.rain_autoconversion, c.ka_qthr, c.ka_y2, c.ka_emk, c.ka_omemk, c.ka_kl2e, c.ka_out, c.ka_inv_tau);
    logistic(pp.snow_autoconversion, c.ks_qthr, c.ks_y2, c.ks_emk, c.ks_omemk, c.ks_kl2e, c.ks_out, c.ks_inv_tau);
    c.nd_coeff = (FT)(1.0 / ((double)pp.rain_autoconversion_nd.tau *
                             std::pow((double)pp.rain_autoconversion_nd.Nc / 1e8, (double)pp.rain_autoconversion_nd.alpha)));
    c.r_is = (FT)pp.r_ice_snow;
    c.r_is2_over_me = (FT)((double)pp.r_ice_snow * (double)pp.r_ice_snow / ((double)mp.cloud_ice.mass.me + (double)mp.cloud_ice.mass.delta_m));
    c.four_pi_n0_icl = (FT)(4.0 * pi * n0_icl);
    // accretion :491-514: q_clo E n0 a0 v0 χa χv λ⁻¹ Γ_accr / (r0/λ⁻¹)^p,  p = ae+ve+Δa+Δv
    auto acc = [&](const auto &m, const auto &a, const auto &v, double E, double n0v0, FT &kk, FT &ee) {
        const double p = (double)a.ae + (double)v.ve + (double)a.delta_a + (double)v.delta_v;
        kk = (FT)(E * (double)a.a0 * (double)a.chi_a * (double)v.chi_v * (double)v.gamma_accr * std::pow((double)m.r0, -p) * n0v0);
        ee = (FT)(1.0 + p);
    };
    FT tmp;
    acc(mp.rain.mass, mp.rain.area, vr, pp.e_lcl_rai, n0_rai * v0c, c.acc_k_lcl_rai, c.acc_e_rai);
    acc(mp.rain.mass, mp.rain.area, vr, pp.e_icl_rai, n0_rai * v0c, c.acc_k_icl_rai, tmp);
    acc(mp.snow.mass, mp.snow.area, vs, pp.e_lcl_sno, v0s, c.acc_k_lcl_sno, c.acc_e_sno);
    acc(mp.snow.mass, mp.snow.area, vs, pp.e_icl_sno, v0s, c.acc_k_icl_sno, tmp);
    {   // accretion_rain_sink :535-561
        const auto &m = mp.rain.mass;
        const auto &a = mp.rain.area;
        const double P = (double)m.me + (double)a.ae + (double)vr.ve + (double)m.delta_m + (double)a.delta_a + (double)vr.delta_v;
        c.sink_k = (FT)((double)pp.e_icl_rai * n0_rai * n0_icl * (double)m.m0 * (double)a.a0 * (double)m.chi_m * (double)a.chi_a *
                        (double)vr.chi_v * (double)vr.gamma_accr_rain_sink * std::pow((double)m.r0, -P) * v0c);
        c.sink_e = (FT)(1.0 + P);
    }
    // accretion_snow_rain :604-644 with type_j: π m0 χm E Γ_coeff / r0^δ · n0_rai n0_sno |Δv| / ρ ·
    //   (2 λi⁻³ λj⁻^(δ+1) + 2(δ+1) λi⁻² λj⁻^(δ+2) + (δ+2)(δ+1) λi⁻¹ λj⁻^(δ+3)) = … · λi⁻¹ λj⁻^(δ+1) · 2 (λi⁻² + (δ+1) λi⁻¹λj⁻¹ + ½(δ+2)(δ+1) λj⁻²)
    auto rs = [&](const auto &mj, FT &kk, FT &dp1, FT &q1, FT &q2) {
        const double d = (double)mj.me + (double)mj.delta_m;
        kk = (FT)(2.0 * pi * (double)mj.m0 * (double)mj.chi_m * (double)pp.e_rai_sno * (double)mj.gamma_coeff * std::pow((double)mj.r0, -d) * n0_rai);
        dp1 = (FT)(d + 1.0); q1 = (FT)(d + 1.0); q2 = (FT)(0.5 * (d + 2.0) * (d + 1.0));
    };
    rs(mp.rain.mass, c.rs_k_rai, c.rs_dp1_rai, c.rs_q1_rai, c.rs_q2_rai);
    rs(mp.snow.mass, c.rs_k_sno, c.rs_dp1_sno, c.rs_q1_sno, c.rs_q2_sno);
    c.coeff_disp = (FT)pp.coeff_disp;
    // ventilation factor (CM1:948-956): a + b ∛Sc Γ_vent √(2 χv/ν) · √v0 · λ⁻¹^(1/2 + (ve+Δv)/2) / r0^((ve+Δv)/2), times 4π n0 [rain]
    const double cbrt_Sc = std::cbrt(nu_air / D_safe);
    auto vent = [&](const auto &ve_, const auto &v, const auto &m, double pre, double sqrt_v0, FT &aa, FT &bb, FT &ee) {
        const double h = ((double)v.ve + (double)v.delta_v) / 2.0;
        aa = (FT)(pre * (double)ve_.a);
        bb = (FT)(pre * (double)ve_.b * cbrt_Sc * (double)v.gamma_vent * std::sqrt(2.0 * (double)v.chi_v / nu_air) * std::pow((double)m.r0, -h) * sqrt_v0);
        ee = (FT)(0.5 + h);
    };
    vent(mp.rain.vent, vr, mp.rain.mass, 4.0 * pi * n0_rai, std::sqrt(v0c), c.ven_a_rai, c.ven_b_rai, c.ven_e_rai);
    vent(mp.snow.vent, vs, mp.snow.mass, 4.0 * pi, std::sqrt(v0s), c.ven_a_sno, c.ven_b_sno, c.ven_e_sno);
    c.mi_k = (FT)(4.0 * pi * n0_icl * (double)mp.air_properties.K_therm);
    return c;
}

// The source terms of one point, before the warm / cold routing of the three temperature-routed accretion terms (BMT:171-198): of each
// warm / cold pair one member is exactly 0, so the unsplit term + is_warm carries the same information.
template <typename FT> struct Mp1mSrc {
    FT vap_lcl, vap_icl, acnv_lcl_rai, acnv_icl_sno, accr_lcl_rai, accr_icl_rai, freeze_icl_rai, accr_icl_sno, vap_rai, vap_sno, melt_icl,
        melt_sno;
    FT S_lcl_sno, S_rai_sno, S_sno_rai, alpha;   // α = warm_accretion_melt_factor (0 at and below T_freeze)
    typename Math<FT>::Mask is_warm;             // T ≥ T_freeze (FT: the VALUE type — a point, or the packed pair f32x2 with a lane mask here; cmx_math.hpp)
    FT qsat_l, qsat_i;                           // q_sat over liquid / ice (LinearizedAverage)
};

// CO.logistic_function_integral (Common.jl:157-173) times 1/τ.  With t = −log(1−e^{−k})/k,
//   (log1pexp(k(x/x0 − 1 + t))/k − t)·x0  =  log((1 − e^{−k}) + e^{−k} e^{y})·x0/k,   y = k x/x0,
// and beyond y = 60 the same quantity is (y − k) + log1p(e^{k−y} − e^{−y}) = y − k to 1e-25.  Since (1 − e^{−k}) > 0 the logarithm is
// ≥ y − k everywhere and the two meet there: the reference's branch is the MAX of the two forms, with the exponent capped where the
// reference switches (so 2^y2 cannot overflow).  In the log2 domain, constants folded on the host: 2 transcendentals + 6 instructions.
// The reference forms the left side — a difference of two terms of size t·x0 — in FT arithmetic, so its own absolute accuracy is
// eps(FT)·t·x0 (the oracle reports 2 t x0 as the operand scale of this term); the right side has absolute error eps(FT)·x0/k from the
// rounding of its argument near 1, the same class.  x arrives clamped to ≥ 0; x0 < ϵ (a parameter-only case: the reference returns x)
// takes a wave-uniform branch of the run-time-flags instantiations.
template <typename FT, bool REGULAR, typename S> __device__ __forceinline__ FT logistic_rate(FT x, S qthr, S y2c, S emk, S omemk, S kl2e, S out, S inv_tau, S eps) {
    using M = Math<FT>;      // FT: the value type, S: its scalar type (the constants)
    // REGULAR: the host has checked x0 ≥ ϵ (mp1m_default_exponents) — no branch in the default Float32 instantiation, whose four points
    // per lane then stay one basic block.  (The Float64 default instantiation keeps the never-taken scalar branch: without it the
    // scheduler hoists the autoconversion constants across the phase boundary and spills 38 SGPR pairs to VGPR lanes, 933 → 1000
    // instructions per point.)
    if constexpr (!REGULAR)
        if (qthr < eps) return x < eps ? FT(0) : FT(x * inv_tau);
    const FT y2 = x * y2c;
    const FT lg2 = M::log2_pn(M::fma(emk, M::exp2_fin(M::min(y2, FT(60.0 * 1.4426950408889634))), omemk));   // y2 = x·y2c, x ≥ 0 finite; the argument lies in [1 − e^{−k}, 1 + e^{60−k}]
    return x < eps ? FT(0) : FT(M::max(lg2, y2 - kl2e) * out);
}

template <typename FT, uint32_t FLAGS = kRuntimeFlags, typename C>
__device__ __forceinline__ Mp1mSrc<FT> mp1m_point(const C &c0, FT rho, FT T, FT q_tot, FT q_lcl, FT q_icl, FT q_rai, FT q_sno) {
    using M = Math<FT>;
    const C *c = &c0;   // Float64: re-derived at the phase boundaries (consts_after, cmx_math.hpp) so only a phase's constants are live
    Mp1mSrc<FT> o{};
    const uint32_t fl = FLAGS == kRuntimeFlags ? c->flags : FLAGS;
    constexpr bool DEFEXP = FLAGS != kRuntimeFlags && (FLAGS & kDefExpBit) != 0;
    using B = typename M::Mask;
    const typename M::Scalar eps = c->eps_1m;   // ϵ_numerics(FT) = cbrt(floatmin(FT))  Utilities.jl:318
    // clamp_to_nonneg — BMT:147-152 (T is not clamped)
    rho = max0(rho); q_tot = max0(q_tot); q_lcl = max0(q_lcl);
    q_icl = max0(q_icl); q_rai = max0(q_rai); q_sno = max0(q_sno);
    // T: a temperature (positive, finite).  ρ > 0 (include/cmx.h): Float32 — the hardware reciprocal — takes a ρ clamped to 0 like the reference; the
    // Float64 entries poison every output of a point with ρ ≤ 0 (cmx_math.hpp bad_density), so the finite-argument form serves
    const FT inv_rho = M::rcp_nz(rho), inv_T = M::rcp_nz(T);
    const B has_lcl = q_lcl > eps, has_icl = q_icl > eps, has_rai = q_rai > eps, has_sno = q_sno > eps;

    // ---- thermodynamics, once -------------------------------------------------------------------------------
    const FT l2_TT = M::log2(T * c->inv_T_tr), dinvT = c->inv_T_tr - inv_T;
    const FT psat_l = M::exp2_fin(M::fma(c->psl_a, l2_TT, M::fma(c->psl_b, dinvT, c->ps_c0)));
    const FT psat_i = M::exp2_fin(M::fma(c->psi_a, l2_TT, M::fma(c->psi_b, dinvT, c->ps_c0)));
    const FT dT0 = T - c->T_0;
    const FT L_v = M::fma(c->dcp_l, dT0, c->LH_v0), L_s = M::fma(c->dcp_i, dT0, c->LH_s0), L_f = M::fma(c->dcp_f, dT0, c->LH_f0);
    const FT q_liq = q_lcl + q_rai, q_ice = q_icl + q_sno;
    const FT q_vap = M::max(FT(0), (q_tot - q_liq) - q_ice);                     // TDI.q_vap :60
    const FT inv_RT = c->inv_R_v * inv_T;
    const FT rho_RvT = rho * (c->R_v * T);
    const FT inv_rho_RvT = inv_rho * inv_RT;                                      // 1/(ρ R_v T) from the two reciprocals at hand
    const FT cp_air = M::fma(c->cpm_qi, q_ice, M::fma(c->cpm_ql, q_liq, M::fma(c->cpm_qt, q_tot, c->cp_d)));
    const FT inv_cp = M::rcp_nz(cp_air);
    const FT dTf = T - c->T_freeze;
    const B above_freezing = T > c->T_freeze;
    o.qsat_l = psat_l * inv_rho_RvT; o.qsat_i = psat_i * inv_rho_RvT;
    // (L/(R_v T) − 1)/T: the factor of dq_sat/dT (NonEq dqcld_dT) and of the conduction term of the G functions (Common.jl:47-102)
    const FT u_v = M::fma(L_v, inv_RT, FT(-1)) * inv_T, u_s = M::fma(L_s, inv_RT, FT(-1)) * inv_T;
    if (fl & CMX_1M_CLOUD_LIQUID_FORMATION) {   // NonEq:117-140: S < 0 ? −min(−S, q)/(τΓ) : S/(τΓ)  =  max(S, −q)/(τΓ) for q ≥ 0
        const FT inv_ts = M::rcp_nz(c->tau_l * M::fma(L_v * inv_cp, o.qsat_l * u_v, FT(1)));
        o.vap_lcl = M::max(q_vap - o.qsat_l, -q_lcl) * inv_ts;
    }
    if (fl & (CMX_1M_CLOUD_ICE_FORMATION_CONST | CMX_1M_CLOUD_ICE_FORMATION_TDEP)) {
        // INP_limiter :56-58 suppresses deposition (a positive tendency) above freezing: the excess is capped at 0 there
        const FT cap = above_freezing ? FT(0) : FT(__builtin_inf());
        const FT ex = q_vap - o.qsat_i;
        const FT lim = clamp_ordered(ex, -q_icl, cap);
        const FT inv_G = M::rcp_nz(M::fma(L_s * inv_cp, o.qsat_i * u_s, FT(1)));
        if (fl & CMX_1M_CLOUD_ICE_FORMATION_CONST) {   // NonEq:168-193
            o.vap_icl = lim * (c->inv_tau_i * inv_G);
        } else {   // TemperatureDependent — NonEq:194-224 with τ_dep = τ_relax(…) :32-50 (Frostenberg 2023 INP number, spherical crystals)
            const FT Tc = M::min(dTf, FT(0));
            const FT l2_N = M::fma(FT(9), M::log2(-(c->td_b10 * Tc)), -c->td_l2a);   // log2 of exp(INP_concentration_mean)
            const FT N = M::exp2(l2_N);
            const FT r = N > eps ? FT(M::exp2((M::log2(c->td_c3 * q_icl) - M::max(l2_N, c->l2_eps)) * FT(1.0 / 3.0))) : FT(0);
            const FT inv_tau_dep = c->td_fourpiD * N * M::max(r, FT(1e-6));
            o.vap_icl = lim * ((ex < FT(0) ? FT(c->inv_tau_i) : inv_tau_dep) * inv_G);
        }
    }
    c = &consts_after(*c, o.vap_icl);
    // ---- supersaturations and G functions ---------------------------------------------------------------------
    const FT inv_ps_l = M::rcp_nz(psat_l), inv_ps_i = M::rcp_nz(psat_i);      // 2^finite of a temperature: positive normal numbers above T ≈ 10 K
    const FT pv = q_vap * rho_RvT;
    const FT S_l = M::fma(pv, inv_ps_l, FT(-1));                                  // TDI.supersaturation_over_liquid
    const FT S_i = M::fma(pv, inv_ps_i, FT(-1));                                  // …over_ice
    // 1/max(p_sat, ϵ) = min(1/p_sat, 1/ϵ): the reciprocal is shared with the supersaturation
    const FT RvDT = c->Rv_over_D * T;
    const FT G_l = M::rcp_nz(M::fma(L_v * c->inv_K, u_v, RvDT * M::min(inv_ps_l, c->inv_eps)));   // Common.jl:47-63
    const FT G_i = M::rcp_nz(M::fma(L_s * c->inv_K, u_s, RvDT * M::min(inv_ps_i, c->inv_eps)));   // :83-102
    const FT SG_i = S_i * G_i;

    c = &consts_after(*c, SG_i);
    // ---- size_distr_parameters — CM1:375-388 ------------------------------------------------------------------
    const FT l2_rq_rai = log2_floored(rho * q_rai), l2_rq_sno = log2_floored(rho * q_sno), l2_rq_icl = log2_floored(rho * q_icl);   // q = 0: ordinary
    // snow: n0 = μ (ρ max(q, ϵ))^ν if q > ϵ else 0 (get_n0 :83-86); λ⁻¹ uses max(n0, ϵ).  For q ≤ ϵ every snow term is gated to 0 below
    // (through n0 = 0 or has_sno), so the slope parameter only has to stay finite there: no select on log2 n0
    const FT l2_n0_sno = M::fma(c->sno_nu, l2_rq_sno, c->sno_l2_mu);
    const FT n0_sno = has_sno ? FT(M::exp2_fin(l2_n0_sno)) : FT(0);      // finite where q_sno > ϵ; the other side of the select is discarded
    const FT l2_rqn_sno = l2_rq_sno - M::max(l2_n0_sno, c->l2_eps);
    const FT li_icl = M::exp2_fin(M::max(c->lam_floor_icl, M::fma(l2_rq_icl, c->lam_a_icl, c->lam_b_icl)));
    // powers of the two slope parameters (see kDefExpBit): rain r = λ⁻¹^¼, snow s = λ⁻¹^⅛
    FT l2_li_rai = FT(0), l2_li_sno = FT(0), li_rai, li_sno, li2_rai, li2_sno;
    FT pr_half = FT(0), pr_075 = FT(0), pr_3h = FT(0), pr_r12 = FT(0), pw_rs = FT(0), ps_q = FT(0), ps_58 = FT(0), pw_sr = FT(0), ps_3q = FT(0);
    if constexpr (DEFEXP) {
        const FT r = M::exp2_fin(M::max(c->lamp_floor_rai, M::fma(l2_rq_rai, c->lamp_a_rai, c->lamp_b_rai)));
        const FT r2 = r * r, r4 = r2 * r2, r8 = r4 * r4, r12 = r8 * r4;
        li_rai = r4; li2_rai = r8; pr_half = r2; pr_075 = r2 * r; pr_3h = r12 * r2; pr_r12 = r12;
        const FT s = M::exp2_fin(M::max(c->lamp_floor_sno, M::fma(l2_rqn_sno, c->lamp_a_sno, c->lamp_b_sno)));
        const FT s2 = s * s, s4 = s2 * s2, s8 = s4 * s4, s16 = s8 * s8, s24 = s16 * s8;
        li_sno = s8; li2_sno = s16; ps_q = s2; ps_58 = s4 * s; ps_3q = s24 * s2;
        pw_rs = s8 * (r8 * r8);      // λ_sno⁻¹ λ_rai⁻⁴
        pw_sr = r4 * s24;            // λ_rai⁻¹ λ_sno⁻³
    } else {
        l2_li_rai = M::max(c->lam_floor_rai, M::fma(l2_rq_rai, c->lam_a_rai, c->lam_b_rai));
        l2_li_sno = M::max(c->lam_floor_sno, M::fma(l2_rqn_sno, c->lam_a_sno, c->lam_b_sno));
        li_rai = M::exp2_fin(l2_li_rai); li_sno = M::exp2_fin(l2_li_sno); li2_rai = li_rai * li_rai; li2_sno = li_sno * li_sno;
    }
    // get_v0 :101-104: v0_rai = v0c · sq (v0c folded into the constants)
    // (Float64: floored at the smallest normal number instead of 0 — air denser than water gets √ = 1.5e-154 — so that this root, its root
    // below and the one of the snow–rain kernel are the positive-argument forms)
    FT sq_arg = M::fma(c->rho_w, inv_rho, FT(-1));
    if constexpr (M::IS_F64) sq_arg = M::max(sq_arg, FT(2.2250738585072014e-308)); else sq_arg = max0(sq_arg);
    const FT sq = M::sqrt_pos(sq_arg);

    c = &consts_after(*c, sq);
    // ---- autoconversion — CM1:354-364, 414-446 ------------------------------------------------------------------
    if (fl & CMX_1M_RAIN_ACNV_KESSLER)
        o.acnv_lcl_rai = logistic_rate<FT, DEFEXP && !M::IS_F64>(q_lcl, c->ka_qthr, c->ka_y2, c->ka_emk, c->ka_omemk, c->ka_kl2e, c->ka_out, c->ka_inv_tau, eps);
    else if (fl & CMX_1M_RAIN_ACNV_PRESCRIBED_ND)
        o.acnv_lcl_rai = q_lcl * c->nd_coeff;
    if (fl & CMX_1M_SNOW_ACNV_NO_SUPERSAT) {
        o.acnv_icl_sno = logistic_rate<FT, DEFEXP && !M::IS_F64>(q_icl, c->ks_qthr, c->ks_y2, c->ks_emk, c->ks_omemk, c->ks_kl2e, c->ks_out, c->ks_inv_tau, eps);
    } else if (fl & CMX_1M_SNOW_ACNV_WITH_SUPERSAT) {
        const FT x = c->r_is * M::rcp_nz(li_icl);                 // ≥ 2^floor > 0
        const FT rate = c->four_pi_n0_icl * SG_i * inv_rho * M::exp2_fin(x * FT(-1.4426950408889634)) *
                        M::fma(x + FT(1), li_icl * li_icl, c->r_is2_over_me);
        o.acnv_icl_sno = m_and(has_icl, S_i > FT(0), T < c->T_freeze) ? rate : FT(0);
    }

    c = &consts_after(*c, o.acnv_icl_sno);
    // ---- accretion — CM1:491-897, routed by temperature as in BMT:171-198 ----------------------------------------
    o.is_warm = T >= c->T_freeze;
    const FT w_melt = M::rcp_nz(L_f) * max0(dTf);            // (T − T_freeze)/L_f above freezing, 0 at and below
    o.alpha = c->cv_l * w_melt;                           // warm_accretion_melt_factor :458-465
    const FT A_rai = sq * (DEFEXP ? pr_3h : M::exp2_fin(c->acc_e_rai * l2_li_rai));
    const FT A_sno = n0_sno * (DEFEXP ? ps_3q : M::exp2_fin(c->acc_e_sno * l2_li_sno));
    if (fl & CMX_1M_ACCR_LCL_RAI) o.accr_lcl_rai = m_and(has_lcl, has_rai) ? FT((q_lcl * c->acc_k_lcl_rai) * A_rai) : FT(0);
    if (fl & CMX_1M_ACCR_LCL_SNO) o.S_lcl_sno = m_and(has_lcl, has_sno) ? FT((q_lcl * c->acc_k_lcl_sno) * A_sno) : FT(0);
    if (fl & CMX_1M_ACCR_ICL_RAI) {
        const B both = m_and(has_icl, has_rai);
        o.accr_icl_rai = both ? FT((q_icl * c->acc_k_icl_rai) * A_rai) : FT(0);
        // λ_rai⁻^6½ = λ_rai⁻^3½ · λ_rai⁻³
        const FT p = DEFEXP ? A_rai * pr_r12 : sq * M::exp2_fin(c->sink_e * l2_li_rai);
        o.freeze_icl_rai = both ? FT((c->sink_k * inv_rho) * (li_icl * p)) : FT(0);
    }
    if (fl & CMX_1M_ACCR_ICL_SNO) o.accr_icl_sno = m_and(has_icl, has_sno) ? FT((q_icl * c->acc_k_icl_sno) * A_sno) : FT(0);
    c = &consts_after(*c, A_sno);
    const FT nir_sno = n0_sno * inv_rho;
    if (fl & CMX_1M_ACCR_RAI_SNO) {   // CM1:604-644, 815-867 (the fall speeds of absent species are not needed: the term is gated on both)
        const FT v_rai = (c->vt_k_rai * sq) * (DEFEXP ? pr_half : M::exp2_fin(c->vt_e_rai * l2_li_rai));
        const FT v_sno = c->vt_k_sno * (DEFEXP ? ps_q : M::exp2_fin(c->vt_e_sno * l2_li_sno));
        const FT dv = v_sno - v_rai;
        const FT dv_eff = M::sqrt_pos(M::fma(dv, dv, c->coeff_disp * M::fma(v_sno, v_sno, v_rai * v_rai)));   // v_sno > 0: the slope parameter is floored
        const FT pre = nir_sno * dv_eff;
        const B both = m_and(has_rai, has_sno);
        const FT X = li_sno * li_rai;
        if constexpr (!DEFEXP) {
            pw_rs = M::exp2_fin(M::fma(c->rs_dp1_rai, l2_li_rai, l2_li_sno));
            pw_sr = M::exp2_fin(M::fma(c->rs_dp1_sno, l2_li_sno, l2_li_rai));
        }
        // i = snow, j = rain (δ of rain);  i = rain, j = snow (δ of snow)
        const FT poly_rs = M::fma(c->rs_q2_rai, li2_rai, M::fma(c->rs_q1_rai, X, li2_sno));
        const FT poly_sr = M::fma(c->rs_q2_sno, li2_sno, M::fma(c->rs_q1_sno, X, li2_rai));
        o.S_rai_sno = both ? FT((pre * c->rs_k_rai) * (pw_rs * poly_rs)) : FT(0);
        o.S_sno_rai = both ? FT((pre * c->rs_k_sno) * (pw_sr * poly_sr)) : FT(0);
    }

    c = &consts_after(*c, o.S_sno_rai);
    // ---- ventilated vapour exchange and melting — CM1:917-1139 ---------------------------------------------------
    if (fl & CMX_1M_RAIN_EVAPORATION) {   // min(0, S G 4π n0/ρ λ⁻² F) with the factor of S ≥ 0: the min moves onto S
        const FT F4 = M::fma(c->ven_b_rai * M::sqrt_pos(sq), DEFEXP ? pr_075 : M::exp2_fin(c->ven_e_rai * l2_li_rai), c->ven_a_rai);
        o.vap_rai = has_rai ? FT((inv_rho * li2_rai) * (F4 * (M::min(S_l, FT(0)) * G_l))) : FT(0);
    }
    const FT F4_sno = M::fma(c->ven_b_sno, DEFEXP ? ps_58 : M::exp2_fin(c->ven_e_sno * l2_li_sno), c->ven_a_sno);
    const FT mp_sno = (nir_sno * li2_sno) * F4_sno;                                // 4π n0/ρ λ⁻² F
    if (fl & (CMX_1M_SNOW_SUBLIMATION_ONLY | CMX_1M_SNOW_DEP_AND_SUBL)) {
        const FT rate = has_sno ? FT(mp_sno * SG_i) : FT(0);
        o.vap_sno = (fl & CMX_1M_SNOW_DEP_AND_SUBL) ? rate : M::min(FT(0), rate);
    }
    if (fl & CMX_1M_CLOUD_ICE_MELT) o.melt_icl = m_and(has_icl, above_freezing) ? FT((c->mi_k * inv_rho) * (w_melt * (li_icl * li_icl))) : FT(0);
    if (fl & CMX_1M_SNOW_MELT) o.melt_sno = m_and(has_sno, above_freezing) ? FT(mp_sno * (c->K_therm * w_melt)) : FT(0);
    return o;
}

// The 18 source terms of `_microphysics_source_terms` in the order of cmx_mp1m_source_column
template <typename FT> __device__ __forceinline__ void mp1m_expand(const Mp1mSrc<FT> &p, FT (&s)[CMX_MP1M_NSRC]) {
    s[CMX_1M_S_PHASE_CHANGE_VAP_LCL] = p.vap_lcl; s[CMX_1M_S_PHASE_CHANGE_VAP_ICL] = p.vap_icl;
    s[CMX_1M_S_ACNV_LCL_RAI] = p.acnv_lcl_rai; s[CMX_1M_S_ACNV_ICL_SNO] = p.acnv_icl_sno;
    s[CMX_1M_S_ACCR_LCL_RAI] = p.accr_lcl_rai;
    s[CMX_1M_S_ACCR_LCL_SNO_COLD] = p.is_warm ? FT(0) : p.S_lcl_sno;
    s[CMX_1M_S_ACCR_LCL_SNO_WARM] = p.is_warm ? p.S_lcl_sno : FT(0);
    s[CMX_1M_S_ACCR_MELT_LCL_SNO] = p.alpha * p.S_lcl_sno;
    s[CMX_1M_S_ACCR_ICL_RAI] = p.accr_icl_rai; s[CMX_1M_S_ACCR_FREEZE_ICL_RAI] = p.freeze_icl_rai; s[CMX_1M_S_ACCR_ICL_SNO] = p.accr_icl_sno;
    s[CMX_1M_S_ACCR_RAI_SNO_COLD] = p.is_warm ? FT(0) : p.S_rai_sno;
    s[CMX_1M_S_ACCR_RAI_SNO_WARM] = p.is_warm ? p.S_sno_rai : FT(0);
    s[CMX_1M_S_ACCR_MELT_RAI_SNO] = p.is_warm ? p.alpha * p.S_rai_sno : FT(0);
    s[CMX_1M_S_PHASE_CHANGE_VAP_RAI] = p.vap_rai; s[CMX_1M_S_PHASE_CHANGE_VAP_SNO] = p.vap_sno;
    s[CMX_1M_S_MELT_ICL_LCL] = p.melt_icl; s[CMX_1M_S_MELT_SNO_RAI] = p.melt_sno;
}

// _aggregate_tendencies — BMT:227-252, formed from the UNSPLIT accretion terms: the same set of non-zero terms with one select per
// tendency instead of six selects and ten additions of zeros (different association of the additions: agreement to rounding).
template <typename FT> __device__ __forceinline__ void mp1m_aggregate(const Mp1mSrc<FT> &p, FT &dl, FT &di, FT &dr, FT &ds) {
    dl = (((p.vap_lcl - p.acnv_lcl_rai) - p.accr_lcl_rai) - p.S_lcl_sno) + p.melt_icl;
    di = (((p.vap_icl - p.acnv_icl_sno) - p.accr_icl_rai) - p.accr_icl_sno) - p.melt_icl;
    // warm: liquid collected by snow is shed as rain (+ melt), rain collects snow;  cold: snow collects rain
    const FT melted = p.alpha * (p.S_lcl_sno + p.S_rai_sno);                 // α = 0 at and below T_freeze
    const FT to_rai = p.is_warm ? (p.S_lcl_sno + p.S_sno_rai) + melted : -p.S_rai_sno;
    const FT to_sno = p.is_warm ? -(p.S_sno_rai + melted) : p.S_lcl_sno + p.S_rai_sno;
    dr = ((((p.acnv_lcl_rai + p.accr_lcl_rai) - p.freeze_icl_rai) + to_rai) + p.vap_rai) + p.melt_sno;
    ds = (((((p.acnv_icl_sno + p.accr_icl_rai) + p.freeze_icl_rai) + p.accr_icl_sno) + to_sno) + p.vap_sno) - p.melt_sno;
}

// A lane of the Float32 vector kernels evaluates 4 points.  An instruction reads at most one SGPR, so every `a·x + b` with both a and b
// kernel constants costs a v_mov_b32 of b per POINT (the compiler rematerialises it rather than keep a VGPR live: 12 of the 266
// instructions of a point).  CMX_1M_HOIST (default on) parks those additive constants in VGPRs once per lane: 70 → 76 VGPRs, same-box
// A/B with CMX_MAX0_INT 0.782–0.793 → 0.778–0.779 ms per 1e8 points.
#ifndef CMX_1M_HOIST
#define CMX_1M_HOIST 1
#endif
template <typename FT> __device__ __forceinline__ void mp1m_hoist_consts(Mp1mConsts<FT> &c) {
#if CMX_1M_HOIST
    c.ps_c0 = keep(c.ps_c0); c.LH_v0 = keep(c.LH_v0); c.LH_s0 = keep(c.LH_s0); c.LH_f0 = keep(c.LH_f0); c.cp_d = keep(c.cp_d);
    c.lam_b_icl = keep(c.lam_b_icl); c.lamp_b_rai = keep(c.lamp_b_rai); c.lamp_b_sno = keep(c.lamp_b_sno); c.sno_l2_mu = keep(c.sno_l2_mu);
    c.ka_omemk = keep(c.ka_omemk); c.ks_omemk = keep(c.ks_omemk); c.ven_a_rai = keep(c.ven_a_rai); c.ven_a_sno = keep(c.ven_a_sno);
#endif
}

// bulk_microphysics_tendencies(Instantaneous(), Microphysics1Moment(), …) of one point — BMT:505-514.  NaN in → NaN out (cmx_math.hpp any_nan)
template <typename FT, uint32_t FLAGS = kRuntimeFlags, typename C>
__device__ __forceinline__ void mp1m_tendencies_point(const C &c, FT rho, FT T, FT q_tot, FT q_lcl, FT q_icl, FT q_rai, FT q_sno, FT &dl, FT &di,
                                                      FT &dr, FT &ds) {
    const Mp1mSrc<FT> p = mp1m_point<FT, FLAGS>(c, rho, T, q_tot, q_lcl, q_icl, q_rai, q_sno);
    mp1m_aggregate<FT>(p, dl, di, dr, ds);
    if constexpr (lanes_of<FT>::value == 1) {
        if ((int)any_nan(rho, q_tot, q_lcl, q_icl, q_rai, q_sno, T) | (int)bad_density(rho)) dl = di = dr = ds = Math<FT>::nan();
    } else {
        const typename Math<FT>::Mask poisoned = nan_mask(rho, q_tot, q_lcl, q_icl, q_rai, q_sno, T);
        dl = poisoned ? Math<FT>::nan() : dl; di = poisoned ? Math<FT>::nan() : di; dr = poisoned ? Math<FT>::nan() : dr; ds = poisoned ? Math<FT>::nan() : ds;
    }
}

// bulk_microphysics_tendencies(LinearizedAverage(), Microphysics1Moment(), …, Δt, nsub) of one point — BMT:572-632: nsub linearized
// implicit substeps (BMT:381-465) of the donor-based linearization dq/dt ≈ M q + e (BMT:269-379), temperature updated from the latent
// heating of each substep.  The entries of M are sums of source terms divided by the donor: each sum is formed first and multiplied
// by the donor's reciprocal once.  `args(dep)` returns the step constants (Float64: read after the point function, like a phase of it).
template <typename FT> struct Mp1mLinArgs { FT q_min, dt, dt_sub, inv_dt_sub, inv_dt, Lv_over_cp, Ls_over_cp; int32_t nsub; };

template <typename FT, uint32_t FLAGS = kRuntimeFlags, typename C, typename AF>
__device__ __forceinline__ void mp1m_linearized_point(const C &c, AF args, int nsub, FT rho, FT T0, FT q_tot, FT ql0, FT qi0, FT qr0, FT qs0, FT &dl_avg,
                                                      FT &di_avg, FT &dr_avg, FT &ds_avg) {
    using M = Math<FT>;
    FT T = T0, ql = ql0, qi = qi0, qr = qr0, qs = qs0;
    for (int k = 0; k < nsub; ++k) {
        const Mp1mSrc<FT> p = mp1m_point<FT, FLAGS>(c, rho, T, q_tot, ql, qi, qr, qs);
        const auto &a = args(p.qsat_i);
        // _linearize — BMT:269-379
        const FT il = M::rcp(M::max(a.q_min, ql)), ii = M::rcp(M::max(a.q_min, qi)), ir = M::rcp(M::max(a.q_min, qr)),
                 is = M::rcp(M::max(a.q_min, qs));
        const FT e1 = M::max(p.vap_lcl, FT(0)), e2 = M::max(p.vap_icl, FT(0)), e4 = M::max(p.vap_sno, FT(0));
        const FT lcl_to_rai = p.acnv_lcl_rai + p.accr_lcl_rai, icl_to_sno = (p.acnv_icl_sno + p.accr_icl_rai) + p.accr_icl_sno;
        const FT warm_lcl_sno = p.is_warm ? p.S_lcl_sno : FT(0), cold_lcl_sno = p.is_warm ? FT(0) : p.S_lcl_sno;
        const FT cold_rai_sno = p.is_warm ? FT(0) : p.S_rai_sno;
        const FT M11 = ((M::min(p.vap_lcl, FT(0)) - lcl_to_rai) - p.S_lcl_sno) * il;
        const FT M31 = (lcl_to_rai + warm_lcl_sno) * il, M41 = cold_lcl_sno * il;
        const FT M12 = p.melt_icl * ii, M42 = icl_to_sno * ii;
        const FT M22 = ((M::min(p.vap_icl, FT(0)) - p.melt_icl) - icl_to_sno) * ii;
        const FT M43 = (p.freeze_icl_rai + cold_rai_sno) * ir;
        const FT M33 = p.vap_rai * ir - M43;                                  // vap_rai ≤ 0: evaporation is a sink of rain
        const FT sno_to_rai = M::fma(p.alpha, p.S_lcl_sno, p.is_warm ? M::fma(p.alpha, p.S_rai_sno, p.S_sno_rai) : FT(0)) + p.melt_sno;
        const FT M34 = sno_to_rai * is;
        const FT M44 = M::min(p.vap_sno, FT(0)) * is - M34;
        // _linearized_implicit_step — BMT:381-465
        const FT q_sat_min = M::min(p.qsat_l, p.qsat_i);
        const FT q_v = (((q_tot - ql) - qi) - qr) - qs;
        const FT alpha = M::min(FT(1), M::max(FT(0), q_v - q_sat_min) * a.inv_dt_sub * M::rcp_nz(M::max((e1 + e2) + e4, FT(M::eps()))));
        const FT a11 = a.inv_dt_sub - M11, a22 = a.inv_dt_sub - M22, a33 = a.inv_dt_sub - M33, a44 = a.inv_dt_sub - M44;
        const FT b1 = M::fma(alpha, e1, a.inv_dt_sub * ql), b2 = M::fma(alpha, e2, a.inv_dt_sub * qi), b3 = a.inv_dt_sub * qr,
                 b4 = M::fma(alpha, e4, a.inv_dt_sub * qs);
        const FT inv_det12 = M::rcp_nz1(a11 * a22);                          // a_kk ≥ 1/Δt_sub > 0
        const FT ql_new = M::fma(b1, a22, M12 * b2) * inv_det12, qi_new = a11 * b2 * inv_det12;
        const FT r3 = M::fma(M31, ql_new, b3);
        const FT r4 = M::fma(M41, ql_new, M::fma(M42, qi_new, b4));
        const FT inv_det = M::rcp_nz1(M::fma(-M34, M43, a33 * a44));          // a33 ≥ 1/Δt + M43, a44 ≥ 1/Δt + M34: positive
        const FT qr_new = M::fma(r3, a44, M34 * r4) * inv_det, qs_new = M::fma(a33, r4, r3 * M43) * inv_det;
        const FT dl = (ql_new - ql) * a.inv_dt_sub, di = (qi_new - qi) * a.inv_dt_sub, dr = (qr_new - qr) * a.inv_dt_sub,
                 ds = (qs_new - qs) * a.inv_dt_sub;
        // BMT:606-617 (the state advances by rate·Δt_sub exactly as the reference writes it)
        ql += dl * a.dt_sub; qi += di * a.dt_sub; qr += dr * a.dt_sub; qs += ds * a.dt_sub;
        T += (a.Lv_over_cp * (dl + dr) + a.Ls_over_cp * (di + ds)) * a.dt_sub;
    }
    const auto &a = args(T);
    typename M::Mask bad = nan_mask(rho, q_tot, ql0, qi0, qr0, qs0, T0);                 // NaN in → NaN out (cmx_math.hpp any_nan)
    if constexpr (M::IS_F64) bad = (bool)((int)bad | (int)bad_density(rho));
    const FT poison = bad ? M::nan() : FT(0);
    dl_avg = (ql - ql0) * a.inv_dt + poison; di_avg = (qi - qi0) * a.inv_dt + poison;
    dr_avg = (qr - qr0) * a.inv_dt + poison; ds_avg = (qs - qs0) * a.inv_dt + poison;
}

template <typename FT> inline Mp1mLinArgs<FT> make_mp1m_lin_args(FT q_min, FT dt, int32_t nsub, FT LH_v0, FT LH_s0, FT cp_d) {
    Mp1mLinArgs<FT> a{};
    a.q_min = q_min; a.dt = dt; a.nsub = nsub;
    a.dt_sub = dt / (FT)nsub;                       // Δt / FT(nsub), BMT:598
    a.inv_dt_sub = FT(1) / a.dt_sub; a.inv_dt = FT(1) / dt;
    a.Lv_over_cp = LH_v0 / cp_d; a.Ls_over_cp = LH_s0 / cp_d;
    return a;
}

// Both constant structs of the LinearizedAverage kernels travel as ONE by-value kernel argument, so that the phase-local reads of the
// Float64 instantiations (front_consts: the FIRST kernel argument) address members of one struct — no hand-computed offsets into the
// kernel-argument segment.
template <typename FT> struct Mp1mLinKernArgs { Mp1mConsts<FT> c; Mp1mLinArgs<FT> a; };

// Microphysics1MOptions as flags (include/cmx.h §5): at most one variant of a process
inline int32_t check_flags_1m(uint32_t flags) {
    if ((flags & CMX_1M_CLOUD_ICE_FORMATION_CONST) && (flags & CMX_1M_CLOUD_ICE_FORMATION_TDEP)) return CMX_ERR_BAD_ARG;
    if ((flags & CMX_1M_RAIN_ACNV_KESSLER) && (flags & CMX_1M_RAIN_ACNV_PRESCRIBED_ND)) return CMX_ERR_BAD_ARG;
    if ((flags & CMX_1M_SNOW_ACNV_NO_SUPERSAT) && (flags & CMX_1M_SNOW_ACNV_WITH_SUPERSAT)) return CMX_ERR_BAD_ARG;
    if ((flags & CMX_1M_SNOW_SUBLIMATION_ONLY) && (flags & CMX_1M_SNOW_DEP_AND_SUBL)) return CMX_ERR_BAD_ARG;
    if (flags >> 17) return CMX_ERR_BAD_ARG;
    return CMX_OK;
}

template <typename FT> struct Mp1mIn { const FT *rho, *T, *q_tot, *q_lcl, *q_icl, *q_rai, *q_sno; };
template <typename FT> struct Mp1mOut { FT *dq_lcl, *dq_icl, *dq_rai, *dq_sno; };
template <typename FT> struct Mp1mSrcOut { FT *col[CMX_MP1M_NSRC]; };

}  // namespace cmx
