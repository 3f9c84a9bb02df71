// cmx_mp1m_kernels.hip — one-moment (Marshall–Palmer) bulk scheme, fused per point, for gfx950; C-ABI entry
// points of include/cmx.h §(5).
//
// Reference (src = /root/reference/src): BulkMicrophysicsTendencies.jl:141-252 (`_microphysics_source_terms`,
// `_aggregate_tendencies`), Microphysics1M.jl (CM1), MicrophysicsNonEq.jl:104-193 (NonEq), Common.jl:47-102,
// 157-173.  13 option-dispatched processes, 18 source terms, 4 tendencies.
//
// HBM-bound pointwise map: 7 state columns in, 4 tendency columns out = 44 B/point (f32).  Same launch shape as
// the SB2006 kernel (one 16-byte vector per lane, one short-lived 256-lane workgroup per tile, non-temporal
// accesses).  Per point: the two saturation pressures, the three Marshall–Palmer slope parameters λ⁻¹ and the
// rain v0 are evaluated ONCE (the reference recomputes them per process through `size_distr_parameters` at
// best once, p_sat six times); every power of λ⁻¹ comes from its one log2; parameter-only factors (Γ terms,
// a0·χa·χv·E…, r0 powers) are folded on the host in double.  The reference's Microphysics1MOptions arrive as a
// flags word in an SGPR: disabled processes are skipped by wave-uniform branches.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <type_traits>

#include "cmx_launch.hpp"
#include "cmx_layout.hpp"
#include "cmx_math.hpp"

namespace cmx {

template <typename FT> struct Mp1mConsts {
    uint32_t flags;
    // thermodynamics
    FT T_0, T_freeze, LH_v0, LH_s0, LH_f0, dcp_l, dcp_i, dcp_f, R_v, inv_R_v, cv_l;
    FT ps_c0, psl_a, psl_b, psi_a, psi_b, inv_T_tr;
    FT cp_d, cpm_qt, cpm_ql, cpm_qi;
    FT tau_l, tau_i;
    FT inv_K, Rv_over_D, eps_1m, l2_eps, K_therm;
    // Marshall–Palmer slopes: log2 λ⁻¹ = (log2(ρ q) + c [− log2 n0]) · e, floored at log2(r0·1e-5)
    FT lam_c_rai, lam_e_rai, lam_floor_rai;
    FT lam_c_sno, lam_e_sno, lam_floor_sno, sno_l2_mu, sno_nu;
    FT lam_c_icl, lam_e_icl, lam_floor_icl;
    FT n0_rai, n0_icl, v0c_rai, rho_w, v0_sno;
    // terminal velocities (accretion_snow_rain): v = vt_c · v0 · exp2(vt_e · log2 λ⁻¹)
    FT vt_c_rai, vt_e_rai, vt_c_sno, vt_e_sno;
    // autoconversion
    FT ka_qthr, ka_k, ka_inv_tau, ka_emk;           // Kessler: threshold, k, 1/τ, e^{-k}
    FT ka_k_over_x0, ka_x0_over_k, ks_k_over_x0, ks_x0_over_k;   // k/max(x0, ϵ), max(x0, ϵ)/k
    FT sqrt_v0_sno, inv_eps;
    FT nd_coeff;                                     // PrescribedNd: 1/(τ (Nc/1e8)^α)
    FT ks_qthr, ks_k, ks_inv_tau, ks_emk;           // snow NoSupersaturation
    FT r_is, inv_me_dm_icl;                          // WithSupersaturation
    // accretion: rate = q_clo · n0 · v0 · c · exp2(e · log2 λ⁻¹)
    FT acc_c_lcl_rai, acc_c_icl_rai, acc_e_rai;
    FT acc_c_lcl_sno, acc_c_icl_sno, acc_e_sno;
    FT sink_c, sink_e;                               // accretion_rain_sink
    FT rs_c_rai, rs_d_rai, rs_c_sno, rs_d_sno, coeff_disp;   // accretion_snow_rain with type_j = rain / snow
    // ventilation: F = a + b' √v0 exp2(e · log2 λ⁻¹)
    FT vent_a_rai, vent_b_rai, vent_e_rai, vent_a_sno, vent_b_sno, vent_e_sno;
    FT four_pi;
    FT fourpi_n0_rai, fourpi_n0_icl, vent_bs_sno;   // host-folded products of two constants (a product of two kernel arguments is a VALU multiply per point)
};

template <typename FT, typename MP, typename TH>
static Mp1mConsts<FT> make_mp1m_consts(const MP &mp, const TH &tp, uint32_t flags, double eps) {
    Mp1mConsts<FT> c{};
    const double l2e = 1.4426950408889634074, pi = 3.14159265358979323846;
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
    const double K_safe = std::fmax((double)mp.air_properties.K_therm, eps), D_safe = std::fmax((double)mp.air_properties.D_vapor, eps);
    const double nu_air = mp.air_properties.nu_air;
    c.inv_K = (FT)(1.0 / K_safe); c.Rv_over_D = (FT)(Rv / D_safe); c.eps_1m = (FT)eps; c.l2_eps = (FT)std::log2(eps);
    c.K_therm = (FT)mp.air_properties.K_therm;
    auto slope = [&](const auto &m, double n0, FT &cc, FT &ee, FT &fl) {   // CM1.lambda_inverse :126-152
        const double d = (double)m.me + (double)m.delta_m;
        const double denom_wo_n0 = (double)m.chi_m * (double)m.m0 * (double)m.gamma_coeff;
        cc = (FT)(std::log2(std::pow((double)m.r0, d) / denom_wo_n0) - (n0 > 0 ? std::log2(std::fmax(n0, eps)) : 0.0));
        ee = (FT)(1.0 / (d + 1.0));
        fl = (FT)std::log2((double)m.r0 * 1e-5);
    };
    slope(mp.rain.mass, (double)mp.rain.n0, c.lam_c_rai, c.lam_e_rai, c.lam_floor_rai);
    slope(mp.snow.mass, 0.0, c.lam_c_sno, c.lam_e_sno, c.lam_floor_sno);
    slope(mp.cloud_ice.mass, (double)mp.cloud_ice.n0, c.lam_c_icl, c.lam_e_icl, c.lam_floor_icl);
    c.sno_l2_mu = (FT)std::log2((double)mp.snow.mu); c.sno_nu = (FT)mp.snow.nu;
    c.n0_rai = (FT)mp.rain.n0; c.n0_icl = (FT)mp.cloud_ice.n0;
    const auto &vr = mp.vel_rain;
    const auto &vs = mp.vel_snow;
    c.v0c_rai = (FT)std::sqrt(8.0 / 3.0 / (double)vr.C_drag * (double)vr.grav * (double)vr.r0);   // get_v0 :101-104
    c.rho_w = (FT)vr.rho_w; c.v0_sno = (FT)vs.v0;
    // terminal_velocity :223-238: χv v0 (λ⁻¹/r0)^(ve+Δv) Γ_term/Γ_coeff
    auto vt = [&](const auto &v, const auto &m, FT &cc, FT &ee) {
        const double p = (double)v.ve + (double)v.delta_v;
        cc = (FT)((double)v.chi_v * (double)v.gamma_term / (double)m.gamma_coeff * std::pow((double)m.r0, -p));
        ee = (FT)p;
    };
    vt(vr, mp.rain.mass, c.vt_c_rai, c.vt_e_rai);
    vt(vs, mp.snow.mass, c.vt_c_sno, c.vt_e_sno);
    // autoconversion
    c.ka_qthr = (FT)pp.rain_autoconversion.q_threshold; c.ka_k = (FT)pp.rain_autoconversion.k;
    c.ka_inv_tau = (FT)(1.0 / (double)pp.rain_autoconversion.tau); c.ka_emk = (FT)std::exp(-(double)pp.rain_autoconversion.k);
    c.nd_coeff = (FT)(1.0 / ((double)pp.rain_autoconversion_nd.tau *
                             std::pow((double)pp.rain_autoconversion_nd.Nc / 1e8, (double)pp.rain_autoconversion_nd.alpha)));
    c.ks_qthr = (FT)pp.snow_autoconversion.q_threshold; c.ks_k = (FT)pp.snow_autoconversion.k;
    {
        const double e1 = (double)Math<FT>::eps_1m();
        const double xa = std::fmax((double)pp.rain_autoconversion.q_threshold, e1), xs = std::fmax((double)pp.snow_autoconversion.q_threshold, e1);
        c.ka_k_over_x0 = (FT)((double)pp.rain_autoconversion.k / xa); c.ka_x0_over_k = (FT)(xa / (double)pp.rain_autoconversion.k);
        c.ks_k_over_x0 = (FT)((double)pp.snow_autoconversion.k / xs); c.ks_x0_over_k = (FT)(xs / (double)pp.snow_autoconversion.k);
        c.sqrt_v0_sno = (FT)std::sqrt((double)vs.v0); c.inv_eps = (FT)(1.0 / e1);
    }
    c.ks_inv_tau = (FT)(1.0 / (double)pp.snow_autoconversion.tau); c.ks_emk = (FT)std::exp(-(double)pp.snow_autoconversion.k);
    c.r_is = (FT)pp.r_ice_snow;
    c.inv_me_dm_icl = (FT)(1.0 / ((double)mp.cloud_ice.mass.me + (double)mp.cloud_ice.mass.delta_m));
    // accretion :491-514: q_clo E n0 a0 v0 χa χv λ⁻¹ Γ_accr / (r0/λ⁻¹)^p,  p = ae+ve+Δa+Δv
    auto acc = [&](const auto &m, const auto &a, const auto &v, double E, FT &cc, FT &ee) {
        const double p = (double)a.ae + (double)v.ve + (double)a.delta_a + (double)v.delta_v;
        cc = (FT)(E * (double)a.a0 * (double)a.chi_a * (double)v.chi_v * (double)v.gamma_accr * std::pow((double)m.r0, -p));
        ee = (FT)(1.0 + p);
    };
    FT tmp;
    acc(mp.rain.mass, mp.rain.area, vr, pp.e_lcl_rai, c.acc_c_lcl_rai, c.acc_e_rai);
    acc(mp.rain.mass, mp.rain.area, vr, pp.e_icl_rai, c.acc_c_icl_rai, tmp);
    acc(mp.snow.mass, mp.snow.area, vs, pp.e_lcl_sno, c.acc_c_lcl_sno, c.acc_e_sno);
    acc(mp.snow.mass, mp.snow.area, vs, pp.e_icl_sno, c.acc_c_icl_sno, tmp);
    {   // accretion_rain_sink :535-561
        const auto &m = mp.rain.mass;
        const auto &a = mp.rain.area;
        const double P = (double)m.me + (double)a.ae + (double)vr.ve + (double)m.delta_m + (double)a.delta_a + (double)vr.delta_v;
        c.sink_c = (FT)((double)pp.e_icl_rai * (double)mp.rain.n0 * (double)mp.cloud_ice.n0 * (double)m.m0 * (double)a.a0 *
                        (double)m.chi_m * (double)a.chi_a * (double)vr.chi_v * (double)vr.gamma_accr_rain_sink * std::pow((double)m.r0, -P));
        c.sink_e = (FT)(1.0 + P);
    }
    // accretion_snow_rain :604-644 with type_j: π m0 χm E Γ_coeff / r0^δ
    auto rs = [&](const auto &mj, FT &cc, FT &dd) {
        const double d = (double)mj.me + (double)mj.delta_m;
        cc = (FT)(pi * (double)mj.m0 * (double)mj.chi_m * (double)pp.e_rai_sno * (double)mj.gamma_coeff * std::pow((double)mj.r0, -d));
        dd = (FT)d;
    };
    rs(mp.rain.mass, c.rs_c_rai, c.rs_d_rai);
    rs(mp.snow.mass, c.rs_c_sno, c.rs_d_sno);
    c.coeff_disp = (FT)pp.coeff_disp;
    // ventilation factor (CM1:948-956): a + b ∛Sc Γ_vent √(2 χv/ν) · √v0 · λ⁻¹^(1/2 + (ve+Δv)/2) / r0^((ve+Δv)/2)
    const double cbrt_Sc = std::cbrt(nu_air / D_safe);
    auto vent = [&](const auto &ve_, const auto &v, const auto &m, FT &aa, FT &bb, FT &ee) {
        const double h = ((double)v.ve + (double)v.delta_v) / 2.0;
        aa = (FT)ve_.a;
        bb = (FT)((double)ve_.b * cbrt_Sc * (double)v.gamma_vent * std::sqrt(2.0 * (double)v.chi_v / nu_air) * std::pow((double)m.r0, -h));
        ee = (FT)(0.5 + h);
    };
    vent(mp.rain.vent, vr, mp.rain.mass, c.vent_a_rai, c.vent_b_rai, c.vent_e_rai);
    vent(mp.snow.vent, vs, mp.snow.mass, c.vent_a_sno, c.vent_b_sno, c.vent_e_sno);
    c.four_pi = (FT)(4.0 * pi);
    c.fourpi_n0_rai = (FT)(4.0 * pi * (double)mp.rain.n0); c.fourpi_n0_icl = (FT)(4.0 * pi * (double)mp.cloud_ice.n0);
    c.vent_bs_sno = (FT)((double)c.vent_b_sno * std::sqrt((double)vs.v0));
    return c;
}

template <typename FT> struct Mp1mSrc {
    FT s[CMX_MP1M_NSRC]; FT qsat_l, qsat_i;   // + q_sat over liquid / ice (LinearizedAverage)
    // the temperature-routed accretion terms before the warm / cold split, for the direct aggregation of the tendencies kernel
    FT S_lcl_sno, S_rai_sno, S_sno_rai, alpha; bool is_warm;
};

// CO.logistic_function_integral (Common.jl:157-173): with t = −log(1−e^{−k})/k,
//   (log1pexp(k(x/x0 − 1 + t))/k − t)·x0  =  log(1 + e^{−k}(e^{y} − 1))·x0/k  =  log((1 − e^{−k}) + e^{−k} e^{y})·x0/k,   y = k x/x0.
// The reference forms the left side — a difference of two terms of size t·x0 — in FT arithmetic, so its own absolute accuracy is
// eps(FT)·t·x0 (the oracle reports 2 t x0 as the operand scale of this term); the right side has absolute error eps(FT)·x0/k from the
// rounding of its argument near 1, the same class, for two transcendentals and four other instructions.  (Round 1 used the
// compensated expm1 / log1p pair: 25 instructions per integral; LEGACY_LOGISTIC=1 restores it for A/B runs.)
#ifndef CMX_LEGACY_LOGISTIC
#define CMX_LEGACY_LOGISTIC 0
#endif
template <typename FT> __device__ __forceinline__ FT logistic_integral(FT x, FT x0, FT k, FT emk, FT k_over_x0, FT x0_over_k, FT eps) {
    using M = Math<FT>;
    // x arrives clamped to ≥ 0 (mp1m_point); the reference's max(x, ϵ) only matters below ϵ, where the result is the 0 of the last line
    const FT xs = x;
    // beyond y = k x/x0 = 60 the same quantity is (y − k) + log1p(e^{k−y} − e^{−y}) = y − k to 1e-25 (and e^{y}
    // cannot overflow below it); k/max(x0, ϵ) and its inverse are parameter-only (host-folded)
    const FT y = xs * k_over_x0;
#if CMX_LEGACY_LOGISTIC
    const FT lg = M::log1p(emk * M::expm1(M::min(y, FT(60))));
#else
    const FT ey = M::exp2(M::min(y, FT(60)) * FT(1.4426950408889634));
    const FT lg = M::log2(M::fma(emk, ey, FT(1) - emk)) * FT(0.6931471805599453);
#endif
    const FT r = (y > FT(60) ? y - k : lg) * x0_over_k;
    return x < eps ? FT(0) : (x0 < eps ? x : r);
}

// FLAGS: the Microphysics1MOptions bits as a compile-time constant (the default option set gets its own instantiation:
// one straight-line basic block, the unselected variants removed), or kRuntimeFlags to read them from the constants.
constexpr uint32_t kRuntimeFlags = 0xffffffffu;
// Internal (not an ABI flag): set in the compile-time FLAGS of the default instantiation when the slope-parameter exponents are the
// default set — rain: fall speed ½, accretion 3½, ice-rain sink 6½, snow–rain kernel 4, ventilation ¾ (all multiples of ¼); snow:
// ¼, 3¼, 3, ⅝ (multiples of ⅛).  The eleven powers of the two λ⁻¹ are then products of ONE exp2 each (r = λ⁻¹^¼, s = λ⁻¹^⅛: 9 + 7
// multiplies) instead of eleven exp2 — in Float64 ≈ 180 of the ≈ 990 instructions of a point.  mp1m_default_exponents() decides on the
// host; any other parameter set takes the run-time-flags kernels with the general exp2(e·log2 λ⁻¹) forms.
constexpr uint32_t kDefExpBit = 0x40000000u;
static_assert((CMX_1M_DEFAULT_OPTIONS & kDefExpBit) == 0, "internal bit collides with an option flag");
#ifndef CMX_1M_DEFEXP
#define CMX_1M_DEFEXP 1      // A/B switch
#endif
template <typename CT> inline bool mp1m_default_exponents(const CT &c) {
    return CMX_1M_DEFEXP && c.vt_e_rai == 0.5 && c.acc_e_rai == 3.5 && c.sink_e == 6.5 && c.rs_d_rai == 3 && c.vent_e_rai == 0.75 && c.vt_e_sno == 0.25 &&
           c.acc_e_sno == 3.25 && c.rs_d_sno == 2 && c.vent_e_sno == 0.625;
}
template <typename FT, uint32_t FLAGS = kRuntimeFlags, typename C>
__device__ __forceinline__ Mp1mSrc<FT> mp1m_point(const C &c0, FT rho, FT T, FT q_tot, FT q_lcl, FT q_icl,
                                                  FT q_rai, FT q_sno) {
    using M = Math<FT>;
    const C *c = &c0;   // Float64: re-derived at the phase boundaries (consts_after, cmx_math.hpp) so only a phase's constants are live
    Mp1mSrc<FT> o;
#pragma unroll
    for (int k = 0; k < CMX_MP1M_NSRC; ++k) o.s[k] = FT(0);
    const uint32_t fl = FLAGS == kRuntimeFlags ? c->flags : FLAGS;
    constexpr bool DEFEXP = FLAGS != kRuntimeFlags && (FLAGS & kDefExpBit) != 0;
    const FT eps = c->eps_1m;   // ϵ_numerics(FT) = cbrt(floatmin(FT))  Utilities.jl:318
    // clamp_to_nonneg — BMT:147-152 (T is not clamped)
    rho = max0(rho); q_tot = max0(q_tot); q_lcl = max0(q_lcl);
    q_icl = max0(q_icl); q_rai = max0(q_rai); q_sno = max0(q_sno);
    const FT inv_rho = M::rcp(rho), inv_T = M::rcp(T);
    const bool has_lcl = q_lcl > eps, has_icl = q_icl > eps, has_rai = q_rai > eps, has_sno = q_sno > eps;

    // ---- thermodynamics, once -------------------------------------------------------------------------------
    const FT l2_TT = M::log2(T * c->inv_T_tr), dinvT = c->inv_T_tr - inv_T;
    const FT psat_l = M::exp2(M::fma(c->psl_a, l2_TT, M::fma(c->psl_b, dinvT, c->ps_c0)));
    const FT psat_i = M::exp2(M::fma(c->psi_a, l2_TT, M::fma(c->psi_b, dinvT, c->ps_c0)));
    const FT dT0 = T - c->T_0;
    const FT L_v = M::fma(c->dcp_l, dT0, c->LH_v0), L_s = M::fma(c->dcp_i, dT0, c->LH_s0), L_f = M::fma(c->dcp_f, dT0, c->LH_f0);
    const FT q_liq = q_lcl + q_rai, q_ice = q_icl + q_sno;
    const FT q_vap = M::max(FT(0), (q_tot - q_liq) - q_ice);                     // TDI.q_vap :60
    const FT rho_RvT = rho * (c->R_v * T);
    const FT inv_rho_RvT = M::rcp(rho_RvT);
    const FT cp_air = M::fma(c->cpm_qi, q_ice, M::fma(c->cpm_ql, q_liq, M::fma(c->cpm_qt, q_tot, c->cp_d)));
    const FT inv_cp = M::rcp(cp_air);
    const FT inv_RT = c->inv_R_v * inv_T;
    const bool above_freezing = T > c->T_freeze;
    const FT dTf = T - c->T_freeze;
    o.qsat_l = psat_l * inv_rho_RvT; o.qsat_i = psat_i * inv_rho_RvT;
    if (fl & CMX_1M_CLOUD_LIQUID_FORMATION) {   // NonEq:117-140
        const FT q_sat = psat_l * inv_rho_RvT;
        const FT dq_dT = q_sat * (L_v * inv_RT * inv_T - inv_T);
        const FT inv_ts = M::rcp(c->tau_l * M::fma(L_v * inv_cp, dq_dT, FT(1)));
        const FT ex = q_vap - q_sat;
        o.s[CMX_1M_S_PHASE_CHANGE_VAP_LCL] = (ex < FT(0) ? -M::min(-ex, q_lcl) : ex) * inv_ts;
    }
    if (fl & CMX_1M_CLOUD_ICE_FORMATION_CONST) {   // NonEq:168-193 + INP_limiter :56-58
        const FT q_sat = psat_i * inv_rho_RvT;
        const FT dq_dT = q_sat * (L_s * inv_RT * inv_T - inv_T);
        const FT inv_ts = M::rcp(c->tau_i * M::fma(L_s * inv_cp, dq_dT, FT(1)));
        const FT ex = q_vap - q_sat;
        const FT tend = (ex < FT(0) ? -M::min(-ex, q_icl) : ex) * inv_ts;
        o.s[CMX_1M_S_PHASE_CHANGE_VAP_ICL] = (above_freezing && tend > FT(0)) ? FT(0) : tend;
    }
    c = &consts_after(*c, o.qsat_i);
    const FT inv_ps_l = M::rcp(psat_l), inv_ps_i = M::rcp(psat_i);
    const FT S_l = M::fma(q_vap * rho_RvT, inv_ps_l, FT(-1));                      // TDI.supersaturation_over_liquid
    const FT S_i = M::fma(q_vap * rho_RvT, inv_ps_i, FT(-1));                      // …over_ice
    const FT LoRT_v = L_v * inv_RT, LoRT_s = L_s * inv_RT;
    // 1/max(p_sat, ϵ) = min(1/p_sat, 1/ϵ): the reciprocal is shared with the supersaturation
    const FT G_l = M::rcp(M::fma(L_v * c->inv_K * inv_T, LoRT_v - FT(1), c->Rv_over_D * T * M::min(inv_ps_l, c->inv_eps)));   // Common.jl:47-63
    const FT G_i = M::rcp(M::fma(L_s * c->inv_K * inv_T, LoRT_s - FT(1), c->Rv_over_D * T * M::min(inv_ps_i, c->inv_eps)));   // :83-102

    c = &consts_after(*c, G_i);
    // ---- size_distr_parameters — CM1:375-388 ------------------------------------------------------------------
    const FT l2_rq_rai = M::log2(rho * q_rai), l2_rq_sno = M::log2(rho * q_sno), l2_rq_icl = M::log2(rho * q_icl);
    const FT l2_li_rai = M::max(c->lam_floor_rai, (l2_rq_rai + c->lam_c_rai) * c->lam_e_rai);
    const FT l2_li_icl = M::max(c->lam_floor_icl, (l2_rq_icl + c->lam_c_icl) * c->lam_e_icl);
    // snow: n0 = μ (ρ max(q, ϵ))^ν if q > ϵ else 0 (get_n0 :83-86); λ⁻¹ uses max(n0, ϵ)
    // (for q_sno > ϵ, ρ·max(q_sno, ϵ) = ρ q_sno: its log2 is l2_rq_sno)
    const FT l2_n0_sno = has_sno ? M::fma(c->sno_nu, l2_rq_sno, c->sno_l2_mu) : c->l2_eps;
    const FT n0_sno = has_sno ? M::exp2(l2_n0_sno) : FT(0);
    const FT l2_li_sno = M::max(c->lam_floor_sno, (l2_rq_sno + c->lam_c_sno - M::max(l2_n0_sno, c->l2_eps)) * c->lam_e_sno);
    // powers of the two slope parameters (see kDefExpBit): rain r = λ⁻¹^¼, snow s = λ⁻¹^⅛
    FT li_rai, li_sno, pr_half = FT(0), pr_075 = FT(0), pr_3h = FT(0), pr_4 = FT(0), pr_6h = FT(0), ps_q = FT(0), ps_58 = FT(0), ps_3 = FT(0), ps_3q = FT(0);
    if constexpr (DEFEXP) {
        const FT r = M::exp2(FT(0.25) * l2_li_rai), r2 = r * r, r4 = r2 * r2, r8 = r4 * r4, r16 = r8 * r8;
        li_rai = r4; pr_half = r2; pr_075 = r2 * r; pr_4 = r16; pr_3h = (r8 * r4) * r2; pr_6h = (r16 * r8) * r2;
        const FT s = M::exp2(FT(0.125) * l2_li_sno), s2 = s * s, s4 = s2 * s2, s8 = s4 * s4, s16 = s8 * s8, s24 = s16 * s8;
        li_sno = s8; ps_q = s2; ps_58 = s4 * s; ps_3 = s24; ps_3q = s24 * s2;
    } else {
        li_rai = M::exp2(l2_li_rai); li_sno = M::exp2(l2_li_sno);
    }
    const FT li_icl = M::exp2(l2_li_icl);
    const FT v0_rai = c->v0c_rai * M::sqrt(M::max(c->rho_w * inv_rho - FT(1), FT(0)));   // get_v0 :101-104
    const FT v0_sno = c->v0_sno;

    c = &consts_after(*c, li_icl);
    // ---- autoconversion — CM1:354-364, 414-446 ------------------------------------------------------------------
    if (fl & CMX_1M_RAIN_ACNV_KESSLER)
        o.s[CMX_1M_S_ACNV_LCL_RAI] = logistic_integral<FT>(q_lcl, c->ka_qthr, c->ka_k, c->ka_emk, c->ka_k_over_x0, c->ka_x0_over_k, eps) * c->ka_inv_tau;
    else if (fl & CMX_1M_RAIN_ACNV_PRESCRIBED_ND)
        o.s[CMX_1M_S_ACNV_LCL_RAI] = q_lcl * c->nd_coeff;
    if (fl & CMX_1M_SNOW_ACNV_NO_SUPERSAT) {
        o.s[CMX_1M_S_ACNV_ICL_SNO] = logistic_integral<FT>(q_icl, c->ks_qthr, c->ks_k, c->ks_emk, c->ks_k_over_x0, c->ks_x0_over_k, eps) * c->ks_inv_tau;
    } else if (fl & CMX_1M_SNOW_ACNV_WITH_SUPERSAT) {
        const FT x = c->r_is * M::rcp(li_icl);
        const FT rate = c->four_pi * S_i * G_i * c->n0_icl * inv_rho * M::exp2(x * FT(-1.4426950408889634)) *
                        M::fma(c->r_is * c->r_is, c->inv_me_dm_icl, (x + FT(1)) * (li_icl * li_icl));
        o.s[CMX_1M_S_ACNV_ICL_SNO] = (has_icl && S_i > FT(0) && T < c->T_freeze) ? rate : FT(0);
    }

    c = &consts_after(*c, o.s[CMX_1M_S_ACNV_ICL_SNO]);
    // ---- accretion — CM1:491-897, routed by temperature as in BMT:171-198 ----------------------------------------
    const bool is_warm = T >= c->T_freeze;
    const FT alpha = (T <= c->T_freeze) ? FT(0) : c->cv_l * M::rcp(L_f) * dTf;    // warm_accretion_melt_factor :458-465
    const FT acc_rai = c->n0_rai * v0_rai * (DEFEXP ? pr_3h : M::exp2(c->acc_e_rai * l2_li_rai));
    const FT acc_sno = n0_sno * v0_sno * (DEFEXP ? ps_3q : M::exp2(c->acc_e_sno * l2_li_sno));
    if (fl & CMX_1M_ACCR_LCL_RAI) o.s[CMX_1M_S_ACCR_LCL_RAI] = (has_lcl && has_rai) ? q_lcl * c->acc_c_lcl_rai * acc_rai : FT(0);
    o.S_lcl_sno = o.S_rai_sno = o.S_sno_rai = FT(0); o.alpha = alpha; o.is_warm = is_warm;
    if (fl & CMX_1M_ACCR_LCL_SNO) {
        const FT S = (has_lcl && has_sno) ? q_lcl * c->acc_c_lcl_sno * acc_sno : FT(0);
        o.S_lcl_sno = S;
        o.s[CMX_1M_S_ACCR_LCL_SNO_COLD] = is_warm ? FT(0) : S;
        o.s[CMX_1M_S_ACCR_LCL_SNO_WARM] = is_warm ? S : FT(0);
        o.s[CMX_1M_S_ACCR_MELT_LCL_SNO] = alpha * S;
    }
    if (fl & CMX_1M_ACCR_ICL_RAI) {
        const bool both = has_icl && has_rai;
        o.s[CMX_1M_S_ACCR_ICL_RAI] = both ? q_icl * c->acc_c_icl_rai * acc_rai : FT(0);
        o.s[CMX_1M_S_ACCR_FREEZE_ICL_RAI] = both ? c->sink_c * inv_rho * v0_rai * li_icl * (DEFEXP ? pr_6h : M::exp2(c->sink_e * l2_li_rai)) : FT(0);
    }
    if (fl & CMX_1M_ACCR_ICL_SNO) o.s[CMX_1M_S_ACCR_ICL_SNO] = (has_icl && has_sno) ? q_icl * c->acc_c_icl_sno * acc_sno : FT(0);
    c = &consts_after(*c, acc_sno);
    if (fl & CMX_1M_ACCR_RAI_SNO) {   // CM1:604-644, 815-867
        const FT v_rai = has_rai ? c->vt_c_rai * v0_rai * (DEFEXP ? pr_half : M::exp2(c->vt_e_rai * l2_li_rai)) : FT(0);
        const FT v_sno = has_sno ? c->vt_c_sno * v0_sno * (DEFEXP ? ps_q : M::exp2(c->vt_e_sno * l2_li_sno)) : FT(0);
        const FT dv = v_sno - v_rai;
        const FT dv_eff = M::sqrt(M::fma(dv, dv, c->coeff_disp * M::fma(v_sno, v_sno, v_rai * v_rai)));
        const FT pre = inv_rho * c->n0_rai * n0_sno * dv_eff;
        const bool both = has_rai && has_sno;
        // Σ = 2 λi³ λj^(δ+1) + 2(δ+1) λi² λj^(δ+2) + (δ+2)(δ+1) λi λj^(δ+3) = λi λj^(δ+1) (2λi² + 2(δ+1)λiλj + (δ+2)(δ+1)λj²)
        // pw = λi⁻¹ λj⁻¹^(δ+1): δ = 3 (rain), 2 (snow) in the default set
        auto kernel = [&](FT cj, FT d, FT li, FT l2_li, FT lj, FT l2_lj, FT pw_default) {
            const FT poly = M::fma(FT(2) * li, li, M::fma(FT(2) * (d + FT(1)) * li, lj, (d + FT(2)) * (d + FT(1)) * (lj * lj)));
            return pre * cj * (DEFEXP ? pw_default : M::exp2(l2_li + (d + FT(1)) * l2_lj)) * poly;
        };
        const FT S_rai_sno = both ? kernel(c->rs_c_rai, c->rs_d_rai, li_sno, l2_li_sno, li_rai, l2_li_rai, li_sno * pr_4) : FT(0);   // i = snow, j = rain
        const FT S_sno_rai = both ? kernel(c->rs_c_sno, c->rs_d_sno, li_rai, l2_li_rai, li_sno, l2_li_sno, li_rai * ps_3) : FT(0);   // i = rain, j = snow
        o.S_rai_sno = S_rai_sno; o.S_sno_rai = S_sno_rai;
        o.s[CMX_1M_S_ACCR_RAI_SNO_COLD] = is_warm ? FT(0) : S_rai_sno;
        o.s[CMX_1M_S_ACCR_RAI_SNO_WARM] = is_warm ? S_sno_rai : FT(0);
        o.s[CMX_1M_S_ACCR_MELT_RAI_SNO] = is_warm ? alpha * S_rai_sno : FT(0);
    }

    c = &consts_after(*c, o.S_sno_rai);
    // ---- ventilated vapour exchange and melting — CM1:917-1139 ---------------------------------------------------
    const FT F_rai = M::fma(c->vent_b_rai * M::sqrt(v0_rai), DEFEXP ? pr_075 : M::exp2(c->vent_e_rai * l2_li_rai), c->vent_a_rai);
    const FT F_sno = M::fma(c->vent_bs_sno, DEFEXP ? ps_58 : M::exp2(c->vent_e_sno * l2_li_sno), c->vent_a_sno);
    const FT mp_rai = c->fourpi_n0_rai * inv_rho * (li_rai * li_rai) * F_rai;      // 4π n0/ρ λ⁻² F
    const FT mp_sno = c->four_pi * n0_sno * inv_rho * (li_sno * li_sno) * F_sno;
    if (fl & CMX_1M_RAIN_EVAPORATION)
        o.s[CMX_1M_S_PHASE_CHANGE_VAP_RAI] = M::min(FT(0), (has_rai && S_l < FT(0)) ? mp_rai * S_l * G_l : FT(0));
    if (fl & (CMX_1M_SNOW_SUBLIMATION_ONLY | CMX_1M_SNOW_DEP_AND_SUBL)) {
        const FT rate = has_sno ? mp_sno * S_i * G_i : FT(0);
        o.s[CMX_1M_S_PHASE_CHANGE_VAP_SNO] = (fl & CMX_1M_SNOW_DEP_AND_SUBL) ? rate : M::min(FT(0), rate);
    }
    const FT melt_f = c->K_therm * M::rcp(L_f) * dTf;
    if (fl & CMX_1M_CLOUD_ICE_MELT)
        o.s[CMX_1M_S_MELT_ICL_LCL] = (has_icl && above_freezing) ? c->fourpi_n0_icl * inv_rho * melt_f * (li_icl * li_icl) : FT(0);
    if (fl & CMX_1M_SNOW_MELT) o.s[CMX_1M_S_MELT_SNO_RAI] = (has_sno && above_freezing) ? mp_sno * melt_f : FT(0);
    return o;
}

template <typename FT> struct Mp1mIn { const FT *rho, *T, *q_tot, *q_lcl, *q_icl, *q_rai, *q_sno; };
template <typename FT> struct Mp1mOut { FT *dq_lcl, *dq_icl, *dq_rai, *dq_sno; };
template <typename FT> struct Mp1mSrcOut { FT *col[CMX_MP1M_NSRC]; };

// _aggregate_tendencies — BMT:227-252 (same order of additions)
template <typename FT> __device__ __forceinline__ void mp1m_aggregate(const FT *s, FT &dl, FT &di, FT &dr, FT &ds) {
    dl = ((((s[CMX_1M_S_PHASE_CHANGE_VAP_LCL] - s[CMX_1M_S_ACNV_LCL_RAI]) - s[CMX_1M_S_ACCR_LCL_RAI]) -
              s[CMX_1M_S_ACCR_LCL_SNO_COLD]) - s[CMX_1M_S_ACCR_LCL_SNO_WARM]) + s[CMX_1M_S_MELT_ICL_LCL];
    di = (((s[CMX_1M_S_PHASE_CHANGE_VAP_ICL] - s[CMX_1M_S_ACNV_ICL_SNO]) - s[CMX_1M_S_ACCR_ICL_RAI]) -
             s[CMX_1M_S_ACCR_ICL_SNO]) - s[CMX_1M_S_MELT_ICL_LCL];
    dr = ((((((((s[CMX_1M_S_ACNV_LCL_RAI] + s[CMX_1M_S_ACCR_LCL_RAI]) + s[CMX_1M_S_ACCR_LCL_SNO_WARM]) +
                  s[CMX_1M_S_ACCR_MELT_LCL_SNO]) - s[CMX_1M_S_ACCR_FREEZE_ICL_RAI]) - s[CMX_1M_S_ACCR_RAI_SNO_COLD]) +
               s[CMX_1M_S_ACCR_RAI_SNO_WARM]) + s[CMX_1M_S_ACCR_MELT_RAI_SNO]) + s[CMX_1M_S_PHASE_CHANGE_VAP_RAI]) +
            s[CMX_1M_S_MELT_SNO_RAI];
    ds = (((((((((s[CMX_1M_S_ACNV_ICL_SNO] + s[CMX_1M_S_ACCR_LCL_SNO_COLD]) - s[CMX_1M_S_ACCR_MELT_LCL_SNO]) +
                   s[CMX_1M_S_ACCR_ICL_RAI]) + s[CMX_1M_S_ACCR_FREEZE_ICL_RAI]) + s[CMX_1M_S_ACCR_ICL_SNO]) +
                s[CMX_1M_S_ACCR_RAI_SNO_COLD]) - s[CMX_1M_S_ACCR_RAI_SNO_WARM]) - s[CMX_1M_S_ACCR_MELT_RAI_SNO]) +
             s[CMX_1M_S_PHASE_CHANGE_VAP_SNO]) - s[CMX_1M_S_MELT_SNO_RAI];
}

// The same sums for the tendencies kernels, formed from the UNSPLIT accretion terms: of each warm / cold pair one member is exactly 0
// (BMT:171-198 routes by T ≥ T_freeze), so Σ is the same set of non-zero terms with one select per tendency instead of six
// selects and ten additions of zeros (different association of the additions: agreement with mp1m_aggregate to rounding).
template <typename FT> __device__ __forceinline__ void mp1m_aggregate_direct(const Mp1mSrc<FT> &p, FT &dl, FT &di, FT &dr, FT &ds) {
    const FT *s = p.s;
    dl = (((s[CMX_1M_S_PHASE_CHANGE_VAP_LCL] - s[CMX_1M_S_ACNV_LCL_RAI]) - s[CMX_1M_S_ACCR_LCL_RAI]) - p.S_lcl_sno) + s[CMX_1M_S_MELT_ICL_LCL];
    di = (((s[CMX_1M_S_PHASE_CHANGE_VAP_ICL] - s[CMX_1M_S_ACNV_ICL_SNO]) - s[CMX_1M_S_ACCR_ICL_RAI]) - s[CMX_1M_S_ACCR_ICL_SNO]) -
         s[CMX_1M_S_MELT_ICL_LCL];
    // warm: liquid collected by snow is shed as rain (+ melt), rain collects snow;  cold: snow collects rain
    const FT melted = p.alpha * (p.S_lcl_sno + p.S_rai_sno);                 // α = 0 at and below T_freeze
    const FT to_rai = p.is_warm ? (p.S_lcl_sno + p.S_sno_rai) + melted : -p.S_rai_sno;
    const FT to_sno = p.is_warm ? -(p.S_sno_rai + melted) : p.S_lcl_sno + p.S_rai_sno;
    dr = ((((s[CMX_1M_S_ACNV_LCL_RAI] + s[CMX_1M_S_ACCR_LCL_RAI]) - s[CMX_1M_S_ACCR_FREEZE_ICL_RAI]) + to_rai) +
          s[CMX_1M_S_PHASE_CHANGE_VAP_RAI]) + s[CMX_1M_S_MELT_SNO_RAI];
    ds = (((((s[CMX_1M_S_ACNV_ICL_SNO] + s[CMX_1M_S_ACCR_ICL_RAI]) + s[CMX_1M_S_ACCR_FREEZE_ICL_RAI]) + s[CMX_1M_S_ACCR_ICL_SNO]) + to_sno) +
          s[CMX_1M_S_PHASE_CHANGE_VAP_SNO]) - s[CMX_1M_S_MELT_SNO_RAI];
}

// bulk_microphysics_tendencies(Instantaneous(), Microphysics1Moment(), …) over columns — BMT:505-514
template <typename FT, int VEC, uint32_t FLAGS = kRuntimeFlags>
__global__ __launch_bounds__(kBlock) void mp1m_tendencies_kernel(const Mp1mConsts<FT> c, const Mp1mIn<FT> in,
                                                                 const Mp1mOut<FT> out, const int64_t nvec) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    FT rho[VEC], T[VEC], q_tot[VEC], q_lcl[VEC], q_icl[VEC], q_rai[VEC], q_sno[VEC];
    if (i < nvec) {
        load_col<FT, VEC>(in.rho, i, rho); load_col<FT, VEC>(in.T, i, T); load_col<FT, VEC>(in.q_tot, i, q_tot);
        load_col<FT, VEC>(in.q_lcl, i, q_lcl); load_col<FT, VEC>(in.q_icl, i, q_icl); load_col<FT, VEC>(in.q_rai, i, q_rai);
        load_col<FT, VEC>(in.q_sno, i, q_sno);
    }
    Math<FT>::prepare();   // Float64: exp2 / log2 tables → LDS while the loads fly (every lane of the workgroup reaches the barrier inside); no-op for Float32
    if (i >= nvec) return;
    FT dl[VEC], di[VEC], dr[VEC], ds[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        const Mp1mSrc<FT> p = mp1m_point<FT, FLAGS>(front_consts<FT, FLAGS == kRuntimeFlags>(c), rho[k], T[k], q_tot[k], q_lcl[k], q_icl[k], q_rai[k], q_sno[k]);
        mp1m_aggregate_direct<FT>(p, dl[k], di[k], dr[k], ds[k]);
        if (any_nan(rho[k], q_tot[k], q_lcl[k], q_icl[k], q_rai[k], q_sno[k], T[k])) dl[k] = di[k] = dr[k] = ds[k] = Math<FT>::nan();
    }
    store_col<FT, VEC>(out.dq_lcl, i, dl); store_col<FT, VEC>(out.dq_icl, i, di);
    store_col<FT, VEC>(out.dq_rai, i, dr); store_col<FT, VEC>(out.dq_sno, i, ds);
}

// _microphysics_source_terms over columns (KAT / diagnostics harness): one point per lane
// ---------------------------------------------------------------------------------------------------------------------
// bulk_microphysics_tendencies(LinearizedAverage(), Microphysics1Moment(), …, Δt, nsub) — BMT:572-632: nsub linearized
// implicit substeps (BMT:381-465) of the donor-based linearization dq/dt ≈ M q + e (BMT:269-379), temperature updated
// from the latent heating of each substep.  One point per lane (the substep loop carries five state variables); the
// source terms come from the same mp1m_point as the Instantaneous mode.
template <typename FT> struct Mp1mLinArgs { FT q_min, dt, dt_sub, inv_dt_sub, inv_dt, Lv_over_cp, Ls_over_cp; int32_t nsub; };

template <typename FT, uint32_t FLAGS = kRuntimeFlags>
__global__ __launch_bounds__(kBlock) void mp1m_linearized_kernel(const Mp1mConsts<FT> c, const Mp1mLinArgs<FT> a0, const Mp1mIn<FT> in,
                                                                 const Mp1mOut<FT> out, const int64_t n) {
    Math<FT>::prepare();   // Float64: exp2 / log2 tables → LDS (no-op for Float32)
    using M = Math<FT>;
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const FT rho = in.rho[i], q_tot = in.q_tot[i];
    const FT ql0 = in.q_lcl[i], qi0 = in.q_icl[i], qr0 = in.q_rai[i], qs0 = in.q_sno[i];
    FT T = in.T[i], ql = ql0, qi = qi0, qr = qr0, qs = qs0;
    constexpr size_t kArgsOffset = (sizeof(Mp1mConsts<FT>) + alignof(Mp1mLinArgs<FT>) - 1) / alignof(Mp1mLinArgs<FT>) * alignof(Mp1mLinArgs<FT>);
    const int nsub = a0.nsub;
    for (int k = 0; k < nsub; ++k) {
        const Mp1mSrc<FT> p = mp1m_point<FT, FLAGS>(front_consts<FT, FLAGS == kRuntimeFlags>(c), rho, T, q_tot, ql, qi, qr, qs);
        const FT *S = p.s;
        // Float64: the step's own constants (second kernel argument) are read after the point function, like a phase of it
        const auto &a = consts_after(kernarg_at<FT, kArgsOffset>(a0), p.qsat_i);
        // _linearize — BMT:269-379
        const FT il = M::rcp(M::max(a.q_min, ql)), ii = M::rcp(M::max(a.q_min, qi)), ir = M::rcp(M::max(a.q_min, qr)),
                 is = M::rcp(M::max(a.q_min, qs));
        FT M11 = FT(0), M12, M22 = FT(0), M31, M33, M34, M41, M42, M43, M44 = FT(0), e1 = FT(0), e2 = FT(0), e4 = FT(0);
        {
            const FT s1 = S[CMX_1M_S_PHASE_CHANGE_VAP_LCL], s2 = S[CMX_1M_S_PHASE_CHANGE_VAP_ICL], s4 = S[CMX_1M_S_PHASE_CHANGE_VAP_SNO];
            e1 = s1 >= FT(0) ? s1 : FT(0); M11 = s1 >= FT(0) ? FT(0) : s1 * il;
            e2 = s2 >= FT(0) ? s2 : FT(0); M22 = s2 >= FT(0) ? FT(0) : s2 * ii;
            e4 = s4 >= FT(0) ? s4 : FT(0); M44 = s4 >= FT(0) ? FT(0) : s4 * is;
        }
        FT D;
        D = S[CMX_1M_S_MELT_ICL_LCL] * ii;            M22 -= D; M12 = D;
        D = S[CMX_1M_S_ACNV_LCL_RAI] * il;            M11 -= D; M31 = D;
        D = S[CMX_1M_S_ACNV_ICL_SNO] * ii;            M22 -= D; M42 = D;
        D = S[CMX_1M_S_ACCR_LCL_RAI] * il;            M11 -= D; M31 += D;
        {
            const FT Dc = S[CMX_1M_S_ACCR_LCL_SNO_COLD] * il, Dw = S[CMX_1M_S_ACCR_LCL_SNO_WARM] * il;
            M11 -= Dc + Dw; M31 += Dw; M41 = Dc;
        }
        D = S[CMX_1M_S_ACCR_MELT_LCL_SNO] * is;       M44 -= D; M34 = D;
        D = S[CMX_1M_S_ACCR_ICL_RAI] * ii;            M22 -= D; M42 += D;
        D = S[CMX_1M_S_ACCR_ICL_SNO] * ii;            M22 -= D; M42 += D;
        D = S[CMX_1M_S_ACCR_FREEZE_ICL_RAI] * ir;     M33 = -D; M43 = D;
        D = S[CMX_1M_S_ACCR_RAI_SNO_WARM] * is;       M44 -= D; M34 += D;
        D = S[CMX_1M_S_ACCR_MELT_RAI_SNO] * is;       M44 -= D; M34 += D;
        D = S[CMX_1M_S_ACCR_RAI_SNO_COLD] * ir;       M33 -= D; M43 += D;
        D = (-S[CMX_1M_S_PHASE_CHANGE_VAP_RAI]) * ir; M33 -= D;
        D = S[CMX_1M_S_MELT_SNO_RAI] * is;            M44 -= D; M34 += D;
        // _linearized_implicit_step — BMT:381-465
        const FT q_sat_min = M::min(p.qsat_l, p.qsat_i);
        const FT q_v = (((q_tot - ql) - qi) - qr) - qs;
        const FT alpha = M::min(FT(1), M::max(FT(0), q_v - q_sat_min) * a.inv_dt_sub * M::rcp(M::max((e1 + e2) + e4, M::eps())));
        const FT a11 = a.inv_dt_sub - M11, a12 = -M12, a22 = a.inv_dt_sub - M22, a31 = -M31, a33 = a.inv_dt_sub - M33, a34 = -M34;
        const FT a41 = -M41, a42 = -M42, a43 = -M43, a44 = a.inv_dt_sub - M44;
        const FT b1 = M::fma(alpha, e1, a.inv_dt_sub * ql), b2 = M::fma(alpha, e2, a.inv_dt_sub * qi), b3 = a.inv_dt_sub * qr,
                 b4 = M::fma(alpha, e4, a.inv_dt_sub * qs);
        const FT inv_det12 = M::rcp(a11 * a22);
        const FT ql_new = (b1 * a22 - a12 * b2) * inv_det12, qi_new = a11 * b2 * inv_det12;
        const FT r3 = M::fma(-a31, ql_new, b3);
        const FT r4 = M::fma(-a41, ql_new, M::fma(-a42, qi_new, b4));
        const FT inv_det = M::rcp(M::fma(-a34, a43, a33 * a44));
        const FT qr_new = (r3 * a44 - a34 * r4) * inv_det, qs_new = (a33 * r4 - r3 * a43) * inv_det;
        const FT dl = (ql_new - ql) * a.inv_dt_sub, di = (qi_new - qi) * a.inv_dt_sub, dr = (qr_new - qr) * a.inv_dt_sub,
                 ds = (qs_new - qs) * a.inv_dt_sub;
        // BMT:606-617 (the state advances by rate·Δt_sub exactly as the reference writes it)
        ql += dl * a.dt_sub; qi += di * a.dt_sub; qr += dr * a.dt_sub; qs += ds * a.dt_sub;
        T += (a.Lv_over_cp * (dl + dr) + a.Ls_over_cp * (di + ds)) * a.dt_sub;
    }
    const FT poison = any_nan(rho, q_tot, ql0, qi0, qr0, qs0, in.T[i]) ? M::nan() : FT(0);   // NaN in → NaN out (cmx_math.hpp any_nan)
    out.dq_lcl[i] = (ql - ql0) * a0.inv_dt + poison; out.dq_icl[i] = (qi - qi0) * a0.inv_dt + poison;
    out.dq_rai[i] = (qr - qr0) * a0.inv_dt + poison; out.dq_sno[i] = (qs - qs0) * a0.inv_dt + poison;
}

template <typename FT>
__global__ __launch_bounds__(kBlock) void mp1m_sources_kernel(const Mp1mConsts<FT> c, const Mp1mIn<FT> in,
                                                              const Mp1mSrcOut<FT> out, const int64_t n) {
    Math<FT>::prepare();   // Float64: exp2 / log2 tables → LDS (no-op for Float32)
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const Mp1mSrc<FT> p = mp1m_point<FT>(front_consts<FT>(c), in.rho[i], in.T[i], in.q_tot[i], in.q_lcl[i], in.q_icl[i], in.q_rai[i], in.q_sno[i]);
#pragma unroll
    for (int k = 0; k < CMX_MP1M_NSRC; ++k)
        if (out.col[k]) out.col[k][i] = p.s[k];
}

// ---- terminal velocities over (ρ, q) columns — CM1:223-270 --------------------------------------------------------
template <typename FT> struct Vel1mConsts {
    FT eps_1m, l2_eps, lam_c_rai, lam_e_rai, lam_floor_rai, lam_c_sno, lam_e_sno, lam_floor_sno, sno_l2_mu, sno_nu;
    FT v0c_rai, rho_w, v0_sno, vt_c_rai, vt_e_rai, vt_c_sno, vt_e_sno;
    FT ch_rho0_l2e, ch_a[3], ch_a3_pow, ch_b[3], ch_b_rho, ch_c1000[3], l2_1000;
    // cloud liquid, Stokes (NonEq:250-265): v = st_pref (ρw/ρ − 1) D², D³ = st_D3 ρ q
    FT st_pref, st_rho_w, st_D3;
    // cloud ice, Chen-2022 small ice reduced at ρᵢ(cloud ice) (NonEq:267-281, Common.jl:304-325): D³ = ci_D3 ρ q
    FT ci_D3, ci_A, ci_B, ci_C, ci_E, ci_F, ci_c2;
    // snow, Chen-2022 large ice reduced at ρᵢ(snow), mass-weighted over the Marshall–Palmer PSD (CM1:272-297):
    // ϕ^κ Γ(b+4)/3! folded into the amplitudes
    FT sn_A, sn_a1, sn_b1, sn_a2, sn_H, sn_b2, sn_c2;
};
template <typename FT> struct Vel1mIO {
    const FT *rho, *q_rai, *q_sno; FT *vt_rai, *vt_sno, *vt_chen;
    const FT *q_lcl, *q_icl; FT *w_lcl, *w_icl, *w_sno_chen;
};

template <typename FT>
__global__ __launch_bounds__(kBlock) void mp1m_velocity_kernel(const Vel1mConsts<FT> c, const Vel1mIO<FT> io, const int64_t n) {
    Math<FT>::prepare();   // Float64: exp2 / log2 tables → LDS (no-op for Float32)
    using M = Math<FT>;
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const FT rho = io.rho[i], eps = c.eps_1m;
    const FT rp = M::max(FT(0), rho);
    if (io.vt_rai || io.vt_chen) {
        const FT q = io.q_rai[i];
        const FT l2_li = M::max(c.lam_floor_rai, (M::log2(rp * M::max(FT(0), q)) + c.lam_c_rai) * c.lam_e_rai);
        if (io.vt_rai) {
            const FT v0 = c.v0c_rai * M::sqrt(M::max(c.rho_w * M::rcp(rho) - FT(1), FT(0)));
            io.vt_rai[i] = q > eps ? c.vt_c_rai * v0 * M::exp2(c.vt_e_rai * l2_li) : FT(0);
        }
        if (io.vt_chen) {   // Chen 2022 rain, mass-weighted (k = 3), diameter slope = 2 λ⁻¹ — CM1:251-270, Common.jl:290-302,414-422
            const FT l2_lam_inv = l2_li + FT(1);
            const FT lam = M::exp2(-l2_lam_inv);
            const FT l2_q = c.ch_rho0_l2e * rp, l2_rho = M::log2(rp);
            FT w = FT(0);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const FT bi = M::fma(-c.ch_b_rho, rp, c.ch_b[k]);
                const FT l2_mag = l2_q + bi * c.l2_1000 + (k == 2 ? c.ch_a3_pow * l2_rho : FT(0));
                const FT l2_den = M::log2(lam + c.ch_c1000[k]);
                const FT e3 = M::exp2(l2_mag - FT(4) * l2_lam_inv - (bi + FT(4)) * l2_den);
                // Γ(b+4)/3! = (b+3)(b+2)(b+1)·Γ(b+1)/6 with Γ on its polynomial range (cmx_math.hpp)
                w = M::fma(c.ch_a[k] * e3, M::tgamma(bi + FT(1)) * (bi + FT(3)) * (bi + FT(2)) * (bi + FT(1)) * FT(1.0 / 6.0), w);
            }
            io.vt_chen[i] = q > eps ? M::max(FT(0), w) : FT(0);
        }
    }
    if (io.vt_sno) {
        const FT q = io.q_sno[i];
        const bool has = q > eps;
        const FT l2_rq = M::log2(rp * M::max(FT(0), q));
        const FT l2_n0 = has ? M::fma(c.sno_nu, l2_rq, c.sno_l2_mu) : c.l2_eps;
        const FT l2_li = M::max(c.lam_floor_sno, (l2_rq + c.lam_c_sno - M::max(l2_n0, c.l2_eps)) * c.lam_e_sno);
        io.vt_sno[i] = has ? c.vt_c_sno * c.v0_sno * M::exp2(c.vt_e_sno * l2_li) : FT(0);
    }
    if (io.w_lcl) {   // CMNonEq.terminal_velocity(::CloudLiquid, ::StokesRegimeVelType, ρ, q): Stokes at the mean-volume diameter
        const FT q = io.q_lcl[i];
        const FT D2 = M::exp2(FT(2.0 / 3.0) * M::log2(c.st_D3 * rho * M::max(FT(0), q)));
        io.w_lcl[i] = q > eps ? c.st_pref * (c.st_rho_w * M::rcp(rho) - FT(1)) * D2 : FT(0);
    }
    if (io.w_icl) {   // CMNonEq.terminal_velocity(::CloudIce, ::Chen2022VelTypeSmallIce, ρ, q): Σ aₖ D^bₖ e^{−cₖD} at that diameter
        const FT q = io.q_icl[i];
        const FT l2_D = FT(1.0 / 3.0) * M::log2(c.ci_D3 * rho * M::max(FT(0), q));
        const FT D = M::exp2(l2_D);
        const FT b = M::fma(rp, c.ci_C, c.ci_B);
        const FT common = M::exp2(c.ci_A * M::log2(rp) + b * (c.l2_1000 + l2_D));           // ρₐ^As · (1000 D)^b
        const FT w = common * M::fma(c.ci_F, M::exp2(-c.ci_c2 * D * FT(1.4426950408889634)), c.ci_E);
        io.w_icl[i] = q > eps ? M::max(FT(0), w) : FT(0);
    }
    if (io.w_sno_chen) {   // CM1.terminal_velocity(::Snow, ::Chen2022VelTypeLargeIce, ρ, q): mass-weighted (k = 3), λ_D⁻¹ = 2 λ⁻¹
        const FT q = io.q_sno[i];
        const bool has = q > eps;
        const FT l2_rq = M::log2(rp * M::max(FT(0), q));
        const FT l2_n0 = has ? M::fma(c.sno_nu, l2_rq, c.sno_l2_mu) : c.l2_eps;
        const FT l2_li = M::max(c.lam_floor_sno, (l2_rq + c.lam_c_sno - M::max(l2_n0, c.l2_eps)) * c.lam_e_sno);
        const FT l2_ld = l2_li + FT(1), lam = M::exp2(-l2_ld);
        const FT l2_ra = c.sn_A * M::log2(rp);
        // aₖ e^{−4 ln λ_D⁻¹ − (bₖ+4) ln(λ_D + cₖ)}: term 1 has c = 0 → λ_D^{−b₁}·… collapses to one power
        const FT t1 = c.sn_a1 * M::exp2(l2_ra + c.sn_b1 * l2_ld);
        const FT t2 = c.sn_a2 * M::exp2(l2_ra + c.sn_H * rp * FT(1.4426950408889634) - FT(4) * l2_ld - (c.sn_b2 + FT(4)) * M::log2(lam + c.sn_c2));
        io.w_sno_chen[i] = has ? M::max(FT(0), t1 + t2) : FT(0);
    }
}

// ---- host side -----------------------------------------------------------------------------------------------------
static int32_t check_flags_1m(uint32_t flags) {
    if (flags & CMX_1M_CLOUD_ICE_FORMATION_TDEP) return CMX_ERR_UNSUPPORTED;
    if ((flags & CMX_1M_RAIN_ACNV_KESSLER) && (flags & CMX_1M_RAIN_ACNV_PRESCRIBED_ND)) return CMX_ERR_BAD_ARG;
    if ((flags & CMX_1M_SNOW_ACNV_NO_SUPERSAT) && (flags & CMX_1M_SNOW_ACNV_WITH_SUPERSAT)) return CMX_ERR_BAD_ARG;
    if ((flags & CMX_1M_SNOW_SUBLIMATION_ONLY) && (flags & CMX_1M_SNOW_DEP_AND_SUBL)) return CMX_ERR_BAD_ARG;
    if (flags >> 17) return CMX_ERR_BAD_ARG;
    return CMX_OK;
}

template <typename FT, typename MP, typename TH>
static int32_t tendencies_1m_entry(const MP *mp, const TH *tps, uint32_t flags, int64_t n, const FT *rho, const FT *T,
                                   const FT *q_tot, const FT *q_lcl, const FT *q_icl, const FT *q_rai, const FT *q_sno,
                                   FT *dq_lcl, FT *dq_icl, FT *dq_rai, FT *dq_sno, void *stream) {
    if (!mp || !tps || n < 0) return CMX_ERR_BAD_ARG;
    if (n > kMaxPoints) return CMX_ERR_UNSUPPORTED;      // one launch cannot express the grid (cmx_launch.hpp)
    if (const int32_t st = check_flags_1m(flags)) return st;
    if (n == 0) return CMX_OK;
    if (!rho || !T || !q_tot || !q_lcl || !q_icl || !q_rai || !q_sno || !dq_lcl || !dq_icl || !dq_rai || !dq_sno) return CMX_ERR_BAD_ARG;
    const Mp1mConsts<FT> c = make_mp1m_consts<FT>(*mp, *tps, flags, (double)Math<FT>::eps_1m());
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    constexpr int VEC = (sizeof(FT) == 8 && CMX_F64_ONE_POINT_PER_LANE) ? 1 : Math<FT>::VEC;
    const void *ptrs[] = {rho, T, q_tot, q_lcl, q_icl, q_rai, q_sno, dq_lcl, dq_icl, dq_rai, dq_sno};
    const uintptr_t mis0 = reinterpret_cast<uintptr_t>(rho) & 15u;
    bool same_mis = (mis0 % sizeof(FT)) == 0;
    for (const void *p : ptrs) same_mis = same_mis && ((reinterpret_cast<uintptr_t>(p) & 15u) == mis0);
    auto launch_range = [&](auto vec_tag, int64_t lo, int64_t count) {
        constexpr int V = decltype(vec_tag)::value;
        if (count <= 0) return;
        Mp1mIn<FT> in{rho + lo, T + lo, q_tot + lo, q_lcl + lo, q_icl + lo, q_rai + lo, q_sno + lo};
        Mp1mOut<FT> out{dq_lcl + lo, dq_icl + lo, dq_rai + lo, dq_sno + lo};
        const int64_t nv = count / V;
        const dim3 grid((unsigned)((nv + kBlock - 1) / kBlock));
        if (flags == CMX_1M_DEFAULT_OPTIONS && mp1m_default_exponents(c))
            hipLaunchKernelGGL((mp1m_tendencies_kernel<FT, V, CMX_1M_DEFAULT_OPTIONS | kDefExpBit>), grid, dim3(kBlock), 0, s, c, in, out, nv);
        else
            hipLaunchKernelGGL((mp1m_tendencies_kernel<FT, V>), grid, dim3(kBlock), 0, s, c, in, out, nv);
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
    CMX_HIP_TRY(hipGetLastError());
    return CMX_OK;
}

// host-model layouts: the Instantaneous tendencies as a policy of the generic adapter kernel (cmx_layout.hpp)
template <typename FT, uint32_t FLAGS> struct Mp1mLayoutPolicy {
    static constexpr int NIN = 7, NOUT = 4, NAOS = 4;   // rho, T, q_tot, q_lcl, q_icl, q_rai, q_sno → (dq_lcl_dt, dq_icl_dt, dq_rai_dt, dq_sno_dt)
    using Consts = Mp1mConsts<FT>;
    template <typename C> static __device__ __forceinline__ void point(const C &c, const FT (&x)[NIN], FT (&y)[NOUT]) {
        const Mp1mSrc<FT> p = mp1m_point<FT, FLAGS>(c, x[0], x[1], x[2], x[3], x[4], x[5], x[6]);
        mp1m_aggregate_direct<FT>(p, y[0], y[1], y[2], y[3]);
        if (any_nan(x[0], x[2], x[3], x[4], x[5], x[6], x[1])) y[0] = y[1] = y[2] = y[3] = Math<FT>::nan();
    }
};
template <typename FT, typename MP, typename TH>
static int32_t fields_1m_entry(const MP *mp, const TH *tps, uint32_t flags, int64_t n_seg, int64_t seg_len, const FT *const *in,
                               const int64_t *in_stride, FT *const *out, const int64_t *out_stride, FT *aos, void *stream) {
    if (!mp || !tps) return CMX_ERR_BAD_ARG;
    if (const int32_t st = check_flags_1m(flags)) return st;
    const Mp1mConsts<FT> c = make_mp1m_consts<FT>(*mp, *tps, flags, (double)Math<FT>::eps_1m());
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (flags == CMX_1M_DEFAULT_OPTIONS && mp1m_default_exponents(c))
        return launch_layout<FT, Mp1mLayoutPolicy<FT, CMX_1M_DEFAULT_OPTIONS | kDefExpBit>>(c, n_seg, seg_len, in, in_stride, out, out_stride, aos, s);
    return launch_layout<FT, Mp1mLayoutPolicy<FT, kRuntimeFlags>>(c, n_seg, seg_len, in, in_stride, out, out_stride, aos, s);
}

template <typename FT, typename MP, typename TH>
static int32_t linearized_1m_entry(const MP *mp, const TH *tps, uint32_t flags, FT q_min, FT dt, int32_t nsub, int64_t n, const FT *rho,
                                   const FT *T, const FT *q_tot, const FT *q_lcl, const FT *q_icl, const FT *q_rai, const FT *q_sno,
                                   FT *dq_lcl, FT *dq_icl, FT *dq_rai, FT *dq_sno, void *stream) {
    if (!mp || !tps || n < 0 || nsub < 1 || !(dt > FT(0)) || !(q_min >= FT(0))) return CMX_ERR_BAD_ARG;
    if (n > kMaxPoints) return CMX_ERR_UNSUPPORTED;      // one launch cannot express the grid (cmx_launch.hpp)
    if (const int32_t st = check_flags_1m(flags)) return st;
    if (n == 0) return CMX_OK;
    if (!rho || !T || !q_tot || !q_lcl || !q_icl || !q_rai || !q_sno || !dq_lcl || !dq_icl || !dq_rai || !dq_sno) return CMX_ERR_BAD_ARG;
    const Mp1mConsts<FT> c = make_mp1m_consts<FT>(*mp, *tps, flags, (double)Math<FT>::eps_1m());
    Mp1mLinArgs<FT> a{};
    a.q_min = q_min; a.dt = dt; a.nsub = nsub;
    a.dt_sub = dt / (FT)nsub;                       // Δt / FT(nsub), BMT:598
    a.inv_dt_sub = FT(1) / a.dt_sub; a.inv_dt = FT(1) / dt;
    a.Lv_over_cp = (FT)tps->LH_v0 / (FT)tps->cp_d; a.Ls_over_cp = (FT)tps->LH_s0 / (FT)tps->cp_d;
    Mp1mIn<FT> in{rho, T, q_tot, q_lcl, q_icl, q_rai, q_sno};
    Mp1mOut<FT> out{dq_lcl, dq_icl, dq_rai, dq_sno};
    if (flags == CMX_1M_DEFAULT_OPTIONS && mp1m_default_exponents(c))
        hipLaunchKernelGGL((mp1m_linearized_kernel<FT, CMX_1M_DEFAULT_OPTIONS | kDefExpBit>), dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                           reinterpret_cast<hipStream_t>(stream), c, a, in, out, n);
    else
        hipLaunchKernelGGL((mp1m_linearized_kernel<FT>), dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                           reinterpret_cast<hipStream_t>(stream), c, a, in, out, n);
    CMX_HIP_TRY(hipGetLastError());
    return CMX_OK;
}

template <typename FT, typename MP, typename TH>
static int32_t sources_1m_entry(const MP *mp, const TH *tps, uint32_t flags, int64_t n, const FT *rho, const FT *T,
                                const FT *q_tot, const FT *q_lcl, const FT *q_icl, const FT *q_rai, const FT *q_sno,
                                FT *const out[CMX_MP1M_NSRC], void *stream) {
    if (!mp || !tps || !out || n < 0) return CMX_ERR_BAD_ARG;
    if (n > kMaxPoints) return CMX_ERR_UNSUPPORTED;      // one launch cannot express the grid (cmx_launch.hpp)
    if (const int32_t st = check_flags_1m(flags)) return st;
    if (n == 0) return CMX_OK;
    if (!rho || !T || !q_tot || !q_lcl || !q_icl || !q_rai || !q_sno) return CMX_ERR_BAD_ARG;
    const Mp1mConsts<FT> c = make_mp1m_consts<FT>(*mp, *tps, flags, (double)Math<FT>::eps_1m());
    Mp1mIn<FT> in{rho, T, q_tot, q_lcl, q_icl, q_rai, q_sno};
    Mp1mSrcOut<FT> o;
    for (int k = 0; k < CMX_MP1M_NSRC; ++k) o.col[k] = out[k];
    hipLaunchKernelGGL((mp1m_sources_kernel<FT>), dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       reinterpret_cast<hipStream_t>(stream), c, in, o, n);
    CMX_HIP_TRY(hipGetLastError());
    return CMX_OK;
}

template <typename FT, typename MP, typename CH>
static Vel1mConsts<FT> make_vel1m_consts(const MP &mpr, const CH *chen) {
    const MP *mp = &mpr;
    // reuse the folding of the tendencies kernel (thermo part unused): a neutral thermo struct keeps it well-defined
    cmx_thermo_f64 tp{461.5, 287.0, 1004.5, 1859.0, 4181.0, 2070.0, 2.5008e6, 2.8344e6, 273.16, 273.16, 611.657, 273.15, 4181.0};
    const Mp1mConsts<FT> m = make_mp1m_consts<FT>(*mp, tp, 0u, (double)Math<FT>::eps_1m());
    Vel1mConsts<FT> c{};
    c.eps_1m = m.eps_1m; c.l2_eps = m.l2_eps; c.lam_c_rai = m.lam_c_rai; c.lam_e_rai = m.lam_e_rai; c.lam_floor_rai = m.lam_floor_rai;
    c.lam_c_sno = m.lam_c_sno; c.lam_e_sno = m.lam_e_sno; c.lam_floor_sno = m.lam_floor_sno; c.sno_l2_mu = m.sno_l2_mu; c.sno_nu = m.sno_nu;
    c.v0c_rai = m.v0c_rai; c.rho_w = m.rho_w; c.v0_sno = m.v0_sno; c.vt_c_rai = m.vt_c_rai; c.vt_e_rai = m.vt_e_rai;
    c.vt_c_sno = m.vt_c_sno; c.vt_e_sno = m.vt_e_sno;
    if (chen) {
        c.ch_rho0_l2e = (FT)((double)chen->rho_0 * 1.4426950408889634074);
        for (int k = 0; k < 3; ++k) { c.ch_a[k] = (FT)chen->a[k]; c.ch_b[k] = (FT)chen->b[k]; c.ch_c1000[k] = (FT)((double)chen->c[k] * 1000.0); }
        c.ch_a3_pow = (FT)chen->a3_pow; c.ch_b_rho = (FT)chen->b_rho; c.l2_1000 = (FT)std::log2(1000.0);
    }
    return c;
}

template <typename FT, typename MP, typename CH>
static int32_t velocity_1m_entry(const MP *mp, const CH *chen, int64_t n, const FT *rho, const FT *q_rai, const FT *q_sno,
                                 FT *vt_rai, FT *vt_sno, FT *vt_chen, void *stream) {
    if (!mp || n < 0) return CMX_ERR_BAD_ARG;
    if (n > kMaxPoints) return CMX_ERR_UNSUPPORTED;      // one launch cannot express the grid (cmx_launch.hpp)
    if (n == 0) return CMX_OK;
    if (!rho || ((vt_rai || vt_chen) && !q_rai) || (vt_sno && !q_sno) || (vt_chen && !chen)) return CMX_ERR_BAD_ARG;
    if (vt_chen && !chen_rain_gamma_domain_ok(*chen)) return CMX_ERR_UNSUPPORTED;   // polynomial Γ domain (cmx_math.hpp)
    const Vel1mConsts<FT> c = make_vel1m_consts<FT>(*mp, chen);
    Vel1mIO<FT> io{rho, q_rai, q_sno, vt_rai, vt_sno, vt_chen, nullptr, nullptr, nullptr, nullptr, nullptr};
    hipLaunchKernelGGL((mp1m_velocity_kernel<FT>), dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       reinterpret_cast<hipStream_t>(stream), c, io, n);
    CMX_HIP_TRY(hipGetLastError());
    return CMX_OK;
}

// The four bulk sedimentation velocities a host model precomputes (ClimaAtmos set_sedimentation_precomputed_quantities;
// test/gpu_clima_core_test.jl:36-45, KA kernel test/gpu_tests.jl:608-630)
template <typename FT, typename MP, typename ST, typename CH, typename CI>
static int32_t sedimentation_entry(const MP *mp, const ST *stokes, const CH *chen_rain, const CI *chen_ice, int64_t n, const FT *rho,
                                   const FT *q_lcl, const FT *q_icl, const FT *q_rai, const FT *q_sno, FT *w_lcl, FT *w_icl, FT *w_rai,
                                   FT *w_sno, void *stream) {
    if (!mp || n < 0) return CMX_ERR_BAD_ARG;
    if (n > kMaxPoints) return CMX_ERR_UNSUPPORTED;      // one launch cannot express the grid (cmx_launch.hpp)
    if (n == 0) return CMX_OK;
    if (!rho || (w_lcl && (!q_lcl || !stokes)) || (w_icl && (!q_icl || !chen_ice)) || (w_rai && (!q_rai || !chen_rain)) ||
        (w_sno && (!q_sno || !chen_ice)))
        return CMX_ERR_BAD_ARG;
    if (w_rai && !chen_rain_gamma_domain_ok(*chen_rain)) return CMX_ERR_UNSUPPORTED;   // polynomial Γ domain (cmx_math.hpp)
    Vel1mConsts<FT> c = make_vel1m_consts<FT>(*mp, chen_rain);
    const double pi = 3.14159265358979323846;
    c.l2_1000 = (FT)std::log2(1000.0);
    if (stokes) {
        c.st_pref = (FT)((double)stokes->grav / (18.0 * (double)stokes->nu_air));
        c.st_rho_w = (FT)stokes->rho_w;
        c.st_D3 = (FT)(6.0 / pi / ((double)mp->cloud_liquid.N_0 * (double)mp->cloud_liquid.rho_w));
    }
    if (chen_ice) {
        {   // small ice reduced at the cloud-ice apparent density — Common.jl:304-325
            const auto &t = chen_ice->small_ice;
            const double ri = (double)mp->cloud_ice.rho_i, l = std::log(ri), sq = std::sqrt(ri);
            c.ci_D3 = (FT)(6.0 / pi / ((double)mp->cloud_ice.N_0 * ri));
            c.ci_A = (FT)((double)t.A[1] * l * l - (double)t.A[2] * l + (double)t.A[0]);
            c.ci_B = (FT)(1.0 / ((double)t.B[0] + (double)t.B[1] * l + (double)t.B[2] / sq));
            c.ci_C = (FT)((double)t.C[0] + (double)t.C[1] * std::exp((double)t.C[2] * ri) + (double)t.C[3] * sq);
            c.ci_E = (FT)((double)t.E[0] - (double)t.E[1] * l * l + (double)t.E[2] * sq);
            c.ci_F = (FT)(-std::exp((double)t.F[0] - (double)t.F[1] * l * l + (double)t.F[2] * l));
            c.ci_c2 = (FT)(1000.0 / ((double)t.G[0] + (double)t.G[1] / l - (double)t.G[2] * l / ri));
        }
        {   // large ice reduced at the snow apparent density — Common.jl:327-350; ϕ^κ Γ(b+4)/3! folded in (CM1:287-295)
            const auto &t = chen_ice->large_ice;
            const double ri = (double)mp->snow.rho_i, l = std::log(ri), sq = std::sqrt(ri);
            const double Al = (double)t.A[0] + (double)t.A[1] * l + (double)t.A[2] / (ri * sq);
            const double Bl = std::exp((double)t.B[0] + (double)t.B[1] * l * l + (double)t.B[2] * l);
            const double Cl = std::exp((double)t.C[0] + (double)t.C[1] / l + (double)t.C[2] / ri);
            const double El = (double)t.E[0] + (double)t.E[1] * l * sq + (double)t.E[2] * sq;
            const double Fl = (double)t.F[0] + (double)t.F[1] * l - std::exp(std::log(-(double)t.F[2]) - ri);
            const double Gl = 1.0 / ((double)t.G[0] + (double)t.G[1] * l * sq + (double)t.G[2] / sq);
            const double Hl = (double)t.H[0] + (double)t.H[1] * ri * ri * sq + std::exp(std::log(-(double)t.H[2]) - ri);
            const double pk = std::pow((double)mp->snow.phi, (double)mp->snow.kappa);
            c.sn_A = (FT)Al; c.sn_b1 = (FT)Cl; c.sn_b2 = (FT)Fl; c.sn_H = (FT)Hl; c.sn_c2 = (FT)(1000.0 * Gl);
            c.sn_a1 = (FT)(pk * Bl * std::pow(1000.0, Cl) * std::tgamma(Cl + 4.0) / 6.0);
            c.sn_a2 = (FT)(pk * El * std::pow(1000.0, Fl) * std::tgamma(Fl + 4.0) / 6.0);
        }
    }
    Vel1mIO<FT> io{rho, q_rai, q_sno, nullptr, nullptr, w_rai, q_lcl, q_icl, w_lcl, w_icl, w_sno};
    hipLaunchKernelGGL((mp1m_velocity_kernel<FT>), dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       reinterpret_cast<hipStream_t>(stream), c, io, n);
    CMX_HIP_TRY(hipGetLastError());
    return CMX_OK;
}

}  // namespace cmx

extern "C" {

int32_t cmx_sedimentation_velocities_f32(const cmx_microphysics_1m_f32 *mp, const cmx_stokes_vel_f32 *stokes,
                                         const cmx_chen2022_rain_vel_f32 *chen_rain, const cmx_chen2022_ice_vel_f32 *chen_ice, int64_t n,
                                         const float *rho, const float *q_lcl, const float *q_icl, const float *q_rai, const float *q_sno,
                                         float *w_lcl, float *w_icl, float *w_rai, float *w_sno, void *stream) {
    return cmx::sedimentation_entry<float>(mp, stokes, chen_rain, chen_ice, n, rho, q_lcl, q_icl, q_rai, q_sno, w_lcl, w_icl, w_rai, w_sno,
                                           stream);
}
int32_t cmx_sedimentation_velocities_f64(const cmx_microphysics_1m_f64 *mp, const cmx_stokes_vel_f64 *stokes,
                                         const cmx_chen2022_rain_vel_f64 *chen_rain, const cmx_chen2022_ice_vel_f64 *chen_ice, int64_t n,
                                         const double *rho, const double *q_lcl, const double *q_icl, const double *q_rai,
                                         const double *q_sno, double *w_lcl, double *w_icl, double *w_rai, double *w_sno, void *stream) {
    return cmx::sedimentation_entry<double>(mp, stokes, chen_rain, chen_ice, n, rho, q_lcl, q_icl, q_rai, q_sno, w_lcl, w_icl, w_rai, w_sno,
                                            stream);
}

int32_t cmx_mp1m_linearized_average_f32(const cmx_microphysics_1m_f32 *mp, const cmx_thermo_f32 *tps, uint32_t flags, float q_min, float dt,
                                        int32_t nsub, int64_t n, const float *rho, const float *T, const float *q_tot, const float *q_lcl,
                                        const float *q_icl, const float *q_rai, const float *q_sno, float *dq_lcl_dt, float *dq_icl_dt,
                                        float *dq_rai_dt, float *dq_sno_dt, void *stream) {
    return cmx::linearized_1m_entry<float>(mp, tps, flags, q_min, dt, nsub, n, rho, T, q_tot, q_lcl, q_icl, q_rai, q_sno, dq_lcl_dt,
                                           dq_icl_dt, dq_rai_dt, dq_sno_dt, stream);
}
int32_t cmx_mp1m_linearized_average_f64(const cmx_microphysics_1m_f64 *mp, const cmx_thermo_f64 *tps, uint32_t flags, double q_min,
                                        double dt, int32_t nsub, int64_t n, const double *rho, const double *T, const double *q_tot,
                                        const double *q_lcl, const double *q_icl, const double *q_rai, const double *q_sno,
                                        double *dq_lcl_dt, double *dq_icl_dt, double *dq_rai_dt, double *dq_sno_dt, void *stream) {
    return cmx::linearized_1m_entry<double>(mp, tps, flags, q_min, dt, nsub, n, rho, T, q_tot, q_lcl, q_icl, q_rai, q_sno, dq_lcl_dt,
                                            dq_icl_dt, dq_rai_dt, dq_sno_dt, stream);
}

int32_t cmx_mp1m_tendencies_f32(const cmx_microphysics_1m_f32 *mp, const cmx_thermo_f32 *tps, uint32_t flags, int64_t n,
                                const float *rho, const float *T, const float *q_tot, const float *q_lcl, const float *q_icl,
                                const float *q_rai, const float *q_sno, float *dq_lcl_dt, float *dq_icl_dt, float *dq_rai_dt,
                                float *dq_sno_dt, void *stream) {
    return cmx::tendencies_1m_entry<float>(mp, tps, flags, n, rho, T, q_tot, q_lcl, q_icl, q_rai, q_sno, dq_lcl_dt, dq_icl_dt,
                                           dq_rai_dt, dq_sno_dt, stream);
}
int32_t cmx_mp1m_tendencies_f64(const cmx_microphysics_1m_f64 *mp, const cmx_thermo_f64 *tps, uint32_t flags, int64_t n,
                                const double *rho, const double *T, const double *q_tot, const double *q_lcl,
                                const double *q_icl, const double *q_rai, const double *q_sno, double *dq_lcl_dt,
                                double *dq_icl_dt, double *dq_rai_dt, double *dq_sno_dt, void *stream) {
    return cmx::tendencies_1m_entry<double>(mp, tps, flags, n, rho, T, q_tot, q_lcl, q_icl, q_rai, q_sno, dq_lcl_dt, dq_icl_dt,
                                            dq_rai_dt, dq_sno_dt, stream);
}
int32_t cmx_mp1m_tendencies_fields_f32(const cmx_microphysics_1m_f32 *mp, const cmx_thermo_f32 *tps, uint32_t flags, int64_t n_seg, int64_t seg_len,
                                       const float *const *in, const int64_t *in_seg_stride, float *const *out, const int64_t *out_seg_stride,
                                       float *out_aos, void *stream) {
    return cmx::fields_1m_entry<float>(mp, tps, flags, n_seg, seg_len, in, in_seg_stride, out, out_seg_stride, out_aos, stream);
}
int32_t cmx_mp1m_tendencies_fields_f64(const cmx_microphysics_1m_f64 *mp, const cmx_thermo_f64 *tps, uint32_t flags, int64_t n_seg, int64_t seg_len,
                                       const double *const *in, const int64_t *in_seg_stride, double *const *out, const int64_t *out_seg_stride,
                                       double *out_aos, void *stream) {
    return cmx::fields_1m_entry<double>(mp, tps, flags, n_seg, seg_len, in, in_seg_stride, out, out_seg_stride, out_aos, stream);
}
int32_t cmx_mp1m_source_terms_f32(const cmx_microphysics_1m_f32 *mp, const cmx_thermo_f32 *tps, uint32_t flags, int64_t n,
                                  const float *rho, const float *T, const float *q_tot, const float *q_lcl, const float *q_icl,
                                  const float *q_rai, const float *q_sno, float *const out[CMX_MP1M_NSRC], void *stream) {
    return cmx::sources_1m_entry<float>(mp, tps, flags, n, rho, T, q_tot, q_lcl, q_icl, q_rai, q_sno, out, stream);
}
int32_t cmx_mp1m_source_terms_f64(const cmx_microphysics_1m_f64 *mp, const cmx_thermo_f64 *tps, uint32_t flags, int64_t n,
                                  const double *rho, const double *T, const double *q_tot, const double *q_lcl,
                                  const double *q_icl, const double *q_rai, const double *q_sno,
                                  double *const out[CMX_MP1M_NSRC], void *stream) {
    return cmx::sources_1m_entry<double>(mp, tps, flags, n, rho, T, q_tot, q_lcl, q_icl, q_rai, q_sno, out, stream);
}
int32_t cmx_mp1m_terminal_velocity_f32(const cmx_microphysics_1m_f32 *mp, const cmx_chen2022_rain_vel_f32 *chen, int64_t n,
                                       const float *rho, const float *q_rai, const float *q_sno, float *vt_rai_blk1m,
                                       float *vt_sno_blk1m, float *vt_rai_chen, void *stream) {
    return cmx::velocity_1m_entry<float>(mp, chen, n, rho, q_rai, q_sno, vt_rai_blk1m, vt_sno_blk1m, vt_rai_chen, stream);
}
int32_t cmx_mp1m_terminal_velocity_f64(const cmx_microphysics_1m_f64 *mp, const cmx_chen2022_rain_vel_f64 *chen, int64_t n,
                                       const double *rho, const double *q_rai, const double *q_sno, double *vt_rai_blk1m,
                                       double *vt_sno_blk1m, double *vt_rai_chen, void *stream) {
    return cmx::velocity_1m_entry<double>(mp, chen, n, rho, q_rai, q_sno, vt_rai_blk1m, vt_sno_blk1m, vt_rai_chen, stream);
}

}  // extern "C"
