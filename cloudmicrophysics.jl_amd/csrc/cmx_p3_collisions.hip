// cmx_p3_collisions.hip — P3 liquid–ice collisions for gfx950: bulk_liquid_ice_collision_sources and the ten
// ∫liquid_ice_collisions integrals (src/P3_processes.jl:96-655); C-ABI entry points of include/cmx.h §(8).
//
// Work decomposition — one grid POINT per 8-lane group (8 points per wave64, 32 per 256-lane workgroup), one OUTER
// quadrature node (ice diameter Dᵢ) per lane:
//   * the reference evaluates, per outer node, two inner integrals over the liquid diameters (cloud: 3 sums by
//     quadrature; rain: N and M in closed form, the rime-volume sum by quadrature).  The inner nodes and everything
//     at them that does not depend on Dᵢ — D, v_l(D), w·n(D)[·m(D)] — are the same for every outer node of a point:
//     the group computes them once (lane j ↔ inner node j) into LDS and every outer node then re-reads them as LDS
//     broadcasts: 1 log + 5 exp per (outer, inner) pair become ≈20 plain VALU operations;
//   * the closed-form rain integral needs ∫ D^{z−1} e^{−αD} dD on [D_lo, D*] and [D*, D_hi] for 4 (α, z₀) families ×
//     6 consecutive z (cross-section monomials i = 0..2 × moments p = 0, 3).  One unregularised lower incomplete gamma
//     function per (family, x) plus the three-term recurrence in z (downward from the series at the top z, upward from
//     the continued fraction at the bottom z — both stable) replaces the reference's 96 regularised gamma_inc calls
//     per outer node by 4; the end-point values (x = α D_lo, α D_hi) do not depend on the outer node either and are
//     evaluated once per point by 8 lanes of the group;
//   * the ten outer sums are reduced over the group with log2(group) xor-shuffle steps.
// The per-point set-up (P3 state, the two Halley solves for the integration bounds, PSD parameters) is evaluated
// redundantly by the 8 lanes of a group — that is what bounds the group width from above; DESIGN.md §4.6 has the
// instruction budget and the measurement that led to 8 (the end-point incomplete gammas and the six
// quantile solves each use up to 8 lanes of the group: 8 is also the smallest width that keeps them one pass).
//
// COMPUTE-bound (FP64 / FP32 vector rate); HBM traffic is 11 input + ≤17 output columns per point.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "cmx_p3.hpp"
#include "cmx_sb2006.hpp"

namespace cmx {

// lanes per grid point (template parameter GROUP of the kernel): 8 up to quadrature order 47, 16 above.  Measured, 1e6 f64 states:
// GaussLegendre(16): 8 → 32.3 ms, 16 → 36.2, 32 → 69.4;  (40): 8 → 134, 16 → 160;  (64): 8 → 441, 16 → 307 (the per-wave LDS caches of 8
// states no longer leave room for two waves per SIMD).  Lanes 0..7 of a group hold the end-point gammas and the quantile solves.
static inline int collision_group(int nq) { return nq >= 48 ? 16 : 8; }

template <typename FT> struct P3ColConsts {
    // rain Chen-2022 curve (table B1; Common.jl:290-302): v_l(D) = Σ_j a_j exp(e_j + b_j logD − c_j D)
    FT r_a[3], r_b[3], r_c[3], r_rho0, r_brho, r_a3pow;
    // cloud PSD (CM2:172-236)
    FT nu_c, mu_c, lg_z1, lg_z2, log_mu_c, z1, logN0_shift, log_km_mu, nu_cD, mu_cD, inv_mu_cD, log_zq_lo, log_zq_hi;
    // rain PSD (CM2:67-110) and its quantile bounds D = D̄·(−log1p(−Y)) (DistributionTools.jl:158-165)
    FT xr_min, xr_max, N0_min, N0_max, lam_min, lam_max, pi_rho_w, k_lo, k_hi;
    int limited;
    // local rime density (MicrophysicsP3.jl:222-239)
    FT rime_a, rime_b, rime_c, rime_rho8, rime_rho_ice;
    // compute_max_freeze_rate (P3_processes.jl:167-201): latent heats, ice saturation pressure, ventilation
    FT K_therm, D_vapor, cp_l, LH_v0, dcp_v, LH_f0, dcp_f, T_0, T_freeze_tps, qsi_frz /* p_sat,ice(T_frz)/(R_v T_frz) */,
        ps_pow, ps_b, inv_T_tr, press_tr, R_v, vent_a, vent_bc;
    FT T_freeze_p3;                 // params.T_freeze of compute_local_rime_density :281
    FT m_fac;                       // ρ_w π/6
    FT tau_wet, rho_i, inv_m_shd;   // bulk sources :612-650
    FT p_lo_m, p_hi_m, K4;          // ice_melt :64-94 inside the fused entry: FT(1e-6), FT(1 − 1e-6), 4 K_therm
    int brent_iters;
};

// host: regularised incomplete gamma P(a, x) to convergence and its inverse (for parameter-only quantiles)
static double host_gamma_P(double a, double x) {
    if (x <= 0) return 0;
    const double lg = std::lgamma(a), pf = std::exp(a * std::log(x) - x - lg);
    if (x < a + 1) {
        double term = 1 / a, sum = term;
        for (int k = 1; k < 2000; ++k) { term *= x / (a + k); sum += term; if (term < sum * 1e-17) break; }
        return pf * sum;
    }
    double b = x + 1 - a, c = 1e300, d = 1 / b, h = d;
    for (int k = 1; k < 2000; ++k) {
        const double an = -k * (k - a);
        b += 2; d = an * d + b; if (std::fabs(d) < 1e-300) d = 1e-300;
        c = b + an / c; if (std::fabs(c) < 1e-300) c = 1e-300;
        d = 1 / d; const double del = d * c; h *= del;
        if (std::fabs(del - 1) < 1e-16) break;
    }
    return 1 - pf * h;
}
static double host_gamma_inc_inv(double a, double p) {   // bisection on log x: parameter-only, called twice per entry
    double lo = -700, hi = 50;
    for (int it = 0; it < 200; ++it) {
        const double mid = 0.5 * (lo + hi);
        (host_gamma_P(a, std::exp(mid)) < p ? lo : hi) = mid;
    }
    return std::exp(0.5 * (lo + hi));
}

template <typename FT, typename IP, typename AP, typename TH>
static P3ColConsts<FT> make_p3col_consts(const IP &ip, const AP &aps, const TH &tps, uint32_t flags) {
    P3ColConsts<FT> k{};
    const double pi = 3.14159265358979323846;
    for (int j = 0; j < 3; ++j) { k.r_a[j] = ip.vel_rain.a[j]; k.r_b[j] = ip.vel_rain.b[j]; k.r_c[j] = (FT)(1000.0 * (double)ip.vel_rain.c[j]); }
    k.r_rho0 = ip.vel_rain.rho_0; k.r_brho = ip.vel_rain.b_rho; k.r_a3pow = ip.vel_rain.a3_pow;
    const auto &pc = ip.cloud_pdf;
    const double nu = pc.nu_c, mu = pc.mu_c, km = (double)pc.rho_w * pi / 6.0;
    k.nu_c = pc.nu_c; k.mu_c = pc.mu_c; k.lg_z1 = pc.loggamma_z1; k.lg_z2 = pc.loggamma_z2; k.log_mu_c = (FT)std::log(mu);
    k.z1 = (FT)((nu + 1) / mu);
    k.logN0_shift = (FT)(std::log(3.0) + (nu + 1) * std::log(km));
    k.log_km_mu = (FT)(mu * std::log(km));
    k.nu_cD = (FT)(3 * nu + 2); k.mu_cD = (FT)(3 * mu); k.inv_mu_cD = (FT)(1.0 / (3 * mu));
    // p = FT(0.00001) (P3_processes.jl:546): the quantile levels are FT numbers
    const FT p_lo = FT(0.00001), p_hi = FT(1) - p_lo;
    const double zq = (3 * nu + 3) / (3 * mu);
    k.log_zq_lo = (FT)std::log(host_gamma_inc_inv(zq, (double)p_lo));
    k.log_zq_hi = (FT)std::log(host_gamma_inc_inv(zq, (double)p_hi));
    const auto &pr = ip.rain_pdf;
    k.xr_min = pr.xr_min; k.xr_max = pr.xr_max; k.N0_min = pr.N0_min; k.N0_max = pr.N0_max; k.lam_min = pr.lambda_min; k.lam_max = pr.lambda_max;
    k.pi_rho_w = (FT)(pi * (double)pr.rho_w);
    k.k_lo = (FT)(-std::log1p(-(double)p_lo)); k.k_hi = (FT)(-std::log1p(-(double)p_hi));
    k.limited = (flags & CMX_P3_RAIN_PDF_LIMITED) != 0;
    const auto &rl = ip.rho_rim_local;
    k.rime_a = rl.a; k.rime_b = rl.b; k.rime_c = rl.c; k.rime_rho_ice = rl.rho_ice;
    k.rime_rho8 = (FT)((double)rl.a + 8.0 * (double)rl.b + 64.0 * (double)rl.c);
    k.K_therm = aps.K_therm; k.D_vapor = aps.D_vapor; k.cp_l = tps.cp_l;
    k.LH_v0 = tps.LH_v0; k.dcp_v = (FT)((double)tps.cp_v - (double)tps.cp_l);
    k.LH_f0 = (FT)((double)tps.LH_s0 - (double)tps.LH_v0); k.dcp_f = (FT)((double)tps.cp_l - (double)tps.cp_i);
    k.T_0 = tps.T_0; k.T_freeze_tps = tps.T_freeze;
    const double dcp_i = (double)tps.cp_v - (double)tps.cp_i, Rv = tps.R_v, Ttr = tps.T_triple;
    k.ps_pow = (FT)(dcp_i / Rv); k.ps_b = (FT)(((double)tps.LH_s0 - dcp_i * (double)tps.T_0) / Rv); k.inv_T_tr = (FT)(1.0 / Ttr);
    k.press_tr = tps.press_triple; k.R_v = tps.R_v;
    const double Tf = tps.T_freeze;
    const double ps_frz = (double)tps.press_triple * std::pow(Tf / Ttr, dcp_i / Rv) * std::exp(((double)tps.LH_s0 - dcp_i * (double)tps.T_0) / Rv * (1 / Ttr - 1 / Tf));
    k.qsi_frz = (FT)(ps_frz / (Rv * Tf));
    k.vent_a = ip.vent.a;
    k.vent_bc = (FT)((double)ip.vent.b * std::cbrt((double)aps.nu_air / (double)aps.D_vapor) / std::sqrt((double)aps.nu_air));
    k.T_freeze_p3 = ip.scheme.T_freeze;
    k.m_fac = (FT)((double)pc.rho_w * pi / 6.0);
    k.tau_wet = ip.scheme.tau_wet; k.rho_i = ip.scheme.rho_i;
    k.inv_m_shd = (FT)(1.0 / ((double)pc.rho_w * 1e-9 * pi / 6.0));      // 1/m_liq(D_shd = 1 mm)
    k.brent_iters = sizeof(FT) == 4 ? 8 : 10;
    k.p_lo_m = (FT)1e-6; k.p_hi_m = (FT)(1.0 - 1e-6); k.K4 = (FT)(4.0 * (double)aps.K_therm);
    return k;
}

// unregularised lower incomplete gamma γ(z₀+m, x), m = 0..5, from ONE series or continued-fraction evaluation:
//   x < z₀ + 3.5: series at z₅ = z₀+5 (x < z₅ − 1.5: fast), then γ(z,x) = (γ(z+1,x) + x^z e^{−x})/z downwards (all terms > 0);
//   otherwise   : continued fraction for Γ(z₀,x), Γ(z+1,x) = z Γ(z,x) + x^z e^{−x} upwards (all terms > 0), γ = Γ(z) − Γ(z,x)
//                 with P(z₅,x) ≳ 0.2 there — no cancellation in either branch.
template <typename FT> __device__ __forceinline__ void lower_gamma6(FT z0, FT x, FT G0 /* Γ(z₀) */, FT (&g)[6]) {
    using P = PM<FT>;
    const FT lx = P::log(x);
    if (x < z0 + FT(3.5)) {
        const FT z5 = z0 + FT(5);
        FT t = P::exp(z5 * lx - x);
        g[5] = t * gamma_series_sum<FT>(z5, x);
        const FT inv_x = P::rcp(x);
#pragma unroll
        for (int m = 4; m >= 0; --m) { t *= inv_x; g[m] = (g[m + 1] + t) * P::rcp(z0 + FT(m)); }
    } else {
        FT t = P::exp(z0 * lx - x);
        FT U = t * gamma_cf_value<FT>(z0, x), Gz = G0;
        g[0] = Gz - U;
#pragma unroll
        for (int m = 0; m < 5; ++m) { const FT z = z0 + FT(m); U = z * U + t; t *= x; Gz *= z; g[m + 1] = Gz - U; }
    }
}

// Segmented columns of the 2M + P3 `_fields_` entry (cmx_layout.hpp for the streaming kernels): the flat state index i = seg·seg_len
// + off addresses column k at p[seg·stride_k + off].  seg_len = 0: plain contiguous columns (seg = 0, off = i — every stride is ignored).
// Order of s_in: ρ, T, q_tot, q_lcl, n_lcl, q_rai, n_rai, q_ice, n_ice, q_rim, b_rim, log λ, INPC shift.
struct SegLayout {
    int64_t seg_len;
    int64_t s_in[13], s_out[8];
};
struct SegPos {
    int64_t seg, off;      // contiguous columns: seg = 0, off = i
    bool segmented;        // wave-uniform (a kernel argument): the contiguous case skips the stride arithmetic — and the scalar loads of the strides
    __device__ __forceinline__ SegPos(const SegLayout &l, int64_t i) {
        segmented = l.seg_len != 0;
        seg = 0; off = i;
        if (segmented) { seg = i / l.seg_len; off = i - seg * l.seg_len; }
    }
    template <typename T> __device__ __forceinline__ T &at(T *p, const int64_t &stride) const { return segmented ? p[seg * stride + off] : p[off]; }
    // the same with the stride read from a table in LDS (the collision kernel: 21 strides held in SGPRs across its sweeps cost it 80–160 more
    // SGPR spills and 1–2 % of the 2M + P3 step; the table is only read for segmented columns)
    template <typename T> __device__ __forceinline__ T &at_tab(T *p, const int64_t *tab, int k) const {
        if (segmented) return p[seg * tab[k] + off];
        return p[off];
    }
};
enum { SEG_RHO = 0, SEG_T, SEG_QTOT, SEG_QLCL, SEG_NLCL, SEG_QRAI, SEG_NRAI, SEG_QICE, SEG_NICE, SEG_QRIM, SEG_BRIM, SEG_LOGLAM, SEG_SHIFT };

template <typename FT> struct P3ColIO {
    const FT *rho_q, *rho_n, *x3, *x4, *L_c, *N_c, *L_r, *N_r, *rho_a, *T, *loglam;
    FT *src[7];      // ∂ₜq_c, ∂ₜq_r, ∂ₜN_c, ∂ₜN_r, ∂ₜL_rim, ∂ₜL_ice, ∂ₜB_rim   (nullable)
    FT *rates[10];   // QCFRZ, QCSHD, NCCOL, QRFRZ, QRSHD, NRCOL, ∫M_col, BCCOL, BRCOL, ∫𝟙_wet M_col   (nullable)
    // FUSED (2M+P3 entry, BMT:898-1083): per-kg prognostic columns in (ρ = rho_a, T, loglam as above), the eight tendency columns
    // (dq_lcl, dn_lcl, dq_rai, dn_rai, dq_ice, dn_ice, dq_rim, db_rim) are read-modify-written
    const FT *q_lcl, *n_lcl, *q_rai, *n_rai, *q_ice, *n_ice, *q_rim, *b_rim;
    FT *out[8];
    SegLayout lay;     // FUSED only (zero-initialised = contiguous)
};

// =====================================================================================================================
// 2M + P3 fused entry — bulk_microphysics_tendencies(::Microphysics2Moment, mp{WR, P3IceParams}, …), BMT:898-1083.
// Two launches on the caller's stream:
//   1. mp2m_p3_pointwise_kernel (lane per point, HBM-bound): clamps, warm rain with the ice content in the vapour budget
//      (sb2006_point<ICE>), F23 deposition nucleation, F23-capped Bigg freezing of cloud drops, ice sublimation/deposition,
//      ice number adjustment, Bigg freezing of rain — writes the eight tendency columns;
//   2. p3_collision_kernel<FUSED> (8 lanes per point, compute-bound): liquid–ice collisions, aggregation and melting where
//      q_ice > ϵ and n_ice > ϵ — read-modify-writes the same columns.
template <typename FT> struct PointwiseConsts {
    FT f23_b10, f23_log_a, f23_T_freeze, inv_tau_act, m_nuc, T_dep, S_thresh;           // Frostenberg 2023 (IceNucleation.jl:250-511)
    FT rf_a, rf_B, T_bigg;                                                             // RainFreezing, T_freeze − 4
    FT ps_pow, ps_b, inv_T_tr, press_tr, R_v, LH_s0, dcp_s, T_0, cp_d, cpm_qt, cpm_ql, cpm_qi, T_freeze_tps;
    FT tau_subdep;
    FT mu_c, lg_z1, lg_z2, log_km_mu, G3, G6, k3, k6, V1, rho_w_V1sq;                   // cloud PSD moments (generalized gamma)
    FT xr_min, xr_max, N0_min, N0_max, lam_min, lam_max, pi_rho_w;                      // rain PSD
    FT inv_rho_i;
};
template <typename FT, typename WR, typename IP, typename TH>
static PointwiseConsts<FT> make_pointwise_consts(const WR &wr, const IP &ip, const TH &tps) {
    PointwiseConsts<FT> k{};
    const double pi = 3.14159265358979323846;
    const auto &fr = ip.ice_nucleation;
    k.f23_b10 = (FT)(-(double)fr.b / 10.0); k.f23_log_a = fr.log_a; k.f23_T_freeze = fr.T_freeze;
    k.inv_tau_act = (FT)(1.0 / (double)ip.tau_act);
    k.m_nuc = (FT)((double)ip.scheme.rho_i * (1e-15 * pi / 6.0));                       // ρ_i · volume_sphere_D(10 µm) — BMT:999-1000
    k.T_dep = (FT)((double)fr.T_freeze - 15.0); k.S_thresh = (FT)0.05;
    k.rf_a = ip.rain_freezing.het_a; k.rf_B = ip.rain_freezing.het_B; k.T_bigg = (FT)((double)tps.T_freeze - 4.0);
    const double dcp_i = (double)tps.cp_v - (double)tps.cp_i, Rv = tps.R_v;
    k.ps_pow = (FT)(dcp_i / Rv); k.ps_b = (FT)(((double)tps.LH_s0 - dcp_i * (double)tps.T_0) / Rv); k.inv_T_tr = (FT)(1.0 / (double)tps.T_triple);
    k.press_tr = tps.press_triple; k.R_v = tps.R_v; k.LH_s0 = tps.LH_s0; k.dcp_s = (FT)dcp_i; k.T_0 = tps.T_0;
    k.cp_d = tps.cp_d; k.cpm_qt = (FT)((double)tps.cp_v - (double)tps.cp_d); k.cpm_ql = (FT)((double)tps.cp_l - (double)tps.cp_v);
    k.cpm_qi = (FT)((double)tps.cp_i - (double)tps.cp_v); k.T_freeze_tps = tps.T_freeze;
    k.tau_subdep = wr.subdep_tau_relax;
    const auto &pc = ip.cloud_pdf;
    const double nu = pc.nu_c, mu = pc.mu_c, km = (double)pc.rho_w * pi / 6.0, nuD = 3 * nu + 2, muD = 3 * mu;
    k.mu_c = pc.mu_c; k.lg_z1 = pc.loggamma_z1; k.lg_z2 = pc.loggamma_z2; k.log_km_mu = (FT)(mu * std::log(km));
    k.G3 = (FT)(std::tgamma((nuD + 4) / muD) / std::tgamma((nuD + 1) / muD));
    k.G6 = (FT)(std::tgamma((nuD + 7) / muD) / std::tgamma((nuD + 1) / muD));
    k.k3 = (FT)(-3.0 / muD); k.k6 = (FT)(-6.0 / muD);
    k.V1 = (FT)(pi / 6.0); k.rho_w_V1sq = (FT)((double)pc.rho_w * (pi / 6.0) * (pi / 6.0));
    const auto &pr = ip.rain_pdf;
    k.xr_min = pr.xr_min; k.xr_max = pr.xr_max; k.N0_min = pr.N0_min; k.N0_max = pr.N0_max; k.lam_min = pr.lambda_min; k.lam_max = pr.lambda_max;
    k.pi_rho_w = (FT)(pi * (double)pr.rho_w);
    k.inv_rho_i = (FT)(1.0 / (double)ip.scheme.rho_i);
    return k;
}
// liquid_freezing_rate(::RainFreezing, ::CloudParticlePDF_SB2006, …) — IceNucleation.jl:355-389: Bigg kinetics over the generalized-gamma
// cloud PSD, M_D^k = n λc^(−k/μ) Γ((ν+1+k)/μ)/Γ((ν+1)/μ) with the Γ ratios folded on the host
template <typename FT, typename KC = PointwiseConsts<FT>>
__device__ __forceinline__ void bigg_cloud(const KC &k, FT J_bigg, FT rho, FT q_lcl, FT n_lcl, FT N_lcl, FT T, FT &bn, FT &bq) {
    using P = PM<FT>;
    const FT eps = P::eps();
    bn = FT(0); bq = FT(0);
    if (n_lcl > eps && q_lcl > eps && T < k.T_bigg && !(N_lcl < eps)) {
        const FT log_lam_c = -k.mu_c * (P::log(rho * q_lcl / N_lcl) + k.lg_z1 - k.lg_z2) + k.log_km_mu;
        bn = J_bigg * k.V1 * (n_lcl * P::exp(k.k3 * log_lam_c) * k.G3);
        bq = J_bigg * k.rho_w_V1sq * (n_lcl * P::exp(k.k6 * log_lam_c) * k.G6);
    }
}
// liquid_freezing_rate(::RainFreezing, pdf_r, …) — IceNucleation.jl:274-311: exponential rain PSD, M_D³ = 6 n D̄³, M_D⁶ = 720 n D̄⁶
template <typename FT, bool LIMITED, typename KC = PointwiseConsts<FT>>
__device__ __forceinline__ void bigg_rain(const KC &k, FT J_bigg, FT rho, FT q_rai, FT n_rai, FT N_rai, FT T, FT &rn, FT &rq) {
    using P = PM<FT>;
    using M = Math<FT>;
    const FT eps = P::eps();
    rn = FT(0); rq = FT(0);
    if (n_rai > eps && q_rai > eps && T < k.T_bigg) {
        const FT sq = q_rai, sN = M::max(N_rai, eps), L = rho * sq;
        FT lam_r;
        if constexpr (!LIMITED) lam_r = P::exp(P::log(k.pi_rho_w / (L / sN)) / FT(3));
        else {
            const FT xt = M::min(M::max(L / sN, k.xr_min), k.xr_max);
            const FT N0 = M::min(M::max(sN * P::exp(P::log(k.pi_rho_w / xt) / FT(3)), k.N0_min), k.N0_max);
            lam_r = M::min(M::max(M::sqrt(M::sqrt(k.pi_rho_w * N0 / L)), k.lam_min), k.lam_max);
        }
        FT Dr = FT(1) / lam_r;
        if constexpr (!LIMITED) { if (N_rai < eps) Dr = FT(0); }                // gate of the not-limited PSD (CM2:83)
        const FT D3 = Dr * Dr * Dr;
        rn = J_bigg * k.V1 * (n_rai * FT(6) * D3);
        rq = J_bigg * k.rho_w_V1sq * (n_rai * FT(720) * (D3 * D3));
    }
}

#ifndef CMX_SB_INTPOW_P3
#define CMX_SB_INTPOW_P3 1
#endif
template <typename FT> struct FusedIO {
    const FT *rho, *T, *q_tot, *q_lcl, *n_lcl, *q_rai, *n_rai, *q_ice, *n_ice, *q_rim, *b_rim, *shift;
    FT *out[8];
    SegLayout lay;
};

// The pointwise part of the 2M + P3 entry for ONE state: in[11] = (ρ, T, q_tot, q_lcl, n_lcl, q_rai, n_rai, q_ice, n_ice, q_rim, b_rim) as stored
// (unclamped), shift = the INPC log shift → d[8] = (dq_lcl, dn_lcl, dq_rai, dn_rai, dq_ice, dn_ice, dq_rim, db_rim), NaN-poisoned.
// Called by the pointwise kernel (a lane per state) and by the epilogue of the collision kernel (the one-launch form).
// FENCED (the collision kernel's epilogue, which has 168 registers): the finished sums of a section are pinned behind a compiler memory fence
// before the next section starts — left alone the scheduler hoists the table reads of all the later exponentials above the warm-rain part and
// spills eight register pairs to scratch.
template <typename FT> __device__ __forceinline__ void section_fence(FT &a, FT &b, FT &c, FT &d) { asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "memory"); }
template <typename FT, bool LIMITED, bool INTPOW, bool FENCED = false, typename SC, typename KC>      // INTPOW: the integer-exponent warm-rain point function (cmx_sb2006.hpp), as the 2M entry
__device__ __forceinline__ void mp2m_p3_point(const SC &sc, const P3Consts<FT> &c, const KC &k, const FT (&in)[11], FT shift,
                                              FT (&d)[8]) {      // SC / KC: SbConsts<FT> / PointwiseConsts<FT>, by value or in the kernel-argument segment (KernArg)
    using P = PM<FT>;
    using M = Math<FT>;
    const FT eps = P::eps();
    // NaN in → NaN out (cmx_math.hpp any_nan): formed first, so that the eleven raw inputs are not alive across the point function
    const FT poison = any_nan(in[0], in[2], in[3], in[4], in[5], in[6], in[7], in[8], in[9], in[10], in[1]) ? M::nan() : FT(0);
    // clamp_to_nonneg — BMT:912-921 (T is not clamped)
    const FT rho = M::max(in[0], FT(0)), T = in[1], q_tot = M::max(in[2], FT(0));
    const FT q_lcl = M::max(in[3], FT(0)), n_lcl = M::max(in[4], FT(0)), q_rai = M::max(in[5], FT(0)), n_rai = M::max(in[6], FT(0));
    const FT q_ice = M::max(in[7], FT(0)), n_ice = M::max(in[8], FT(0)), q_rim = M::max(in[9], FT(0)), b_rim = M::max(in[10], FT(0));
    const FT N_lcl = rho * n_lcl, N_rai = rho * n_rai, inv_rho = FT(1) / rho;
    // warm rain — BMT:942 → warm_rain_tendencies_2m :707-782
    const SbRates<FT> w = sb2006_point<FT, LIMITED, VEL_NONE, true, INTPOW>(sc, rho, T, q_tot, q_lcl, q_rai, N_lcl, N_rai, n_lcl, n_rai, q_ice, k.cpm_qi);
    FT dq_lcl = (w.cond + w.au_dq_lcl) + w.ac_dq_lcl;
    FT dn_lcl = M::fma(w.lsc_plus_au + w.ac_dN_lcl, w.inv_rho, w.na_lcl);
    FT dq_rai = (w.evq + w.au_dq_rai) + w.ac_dq_rai;
    FT dn_rai = M::fma(((w.evN + w.au_dN_rai) + w.rsc) + w.rbr, w.inv_rho, w.na_rai);
    FT dq_ice = FT(0), dn_ice = FT(0), dq_rim = FT(0), db_rim = FT(0);
    if constexpr (FENCED) section_fence(dq_lcl, dn_lcl, dq_rai, dn_rai);
    // P3 state (F_rim, ρ_rim) — state_from_prognostic, P3_particle_properties.jl:101-106
    P3Point<FT> s;
    p3_rime_state<FT>(c, q_ice * rho, q_rim * rho, b_rim * rho, s.F_rim, s.rho_rim);
    // ice saturation
    const FT inv_T = FT(1) / T;
    const FT ps_i = k.press_tr * P::exp(k.ps_pow * P::log(T * k.inv_T_tr) + k.ps_b * (k.inv_T_tr - inv_T));
    const FT qsi = ps_i / (rho * k.R_v * T);
    const FT q_vap = M::max(FT(0), (q_tot - (q_lcl + q_rai)) - q_ice);
    // Frostenberg INPC per kg — INP_concentration_mean :250-253
    const FT T_c = M::min(T - k.f23_T_freeze, FT(0));
    const FT inpc_kg = P::exp(FT(9) * P::log(k.f23_b10 * T_c) - k.f23_log_a + shift) * inv_rho;
    const FT n_active = n_ice;                                                    // NIceProxyDepletion :527
    {   // deposition_rate :491-511
        const bool cond = (T < k.T_dep) && (q_vap / qsi - FT(1) > k.S_thresh);
        const FT rn = cond ? M::max(FT(0), inpc_kg - n_active) * k.inv_tau_act : FT(0);
        const FT rq = M::min(k.m_nuc * rn, M::max(FT(0), q_vap - qsi) * (FT(0.5) * k.inv_tau_act));
        dn_ice += rn; dq_ice += rq;
    }
    const FT J_bigg = k.rf_B * P::exp(k.rf_a * (k.T_freeze_tps - T));                // RainFreezing functor, parameters/IceNucleation.jl:146
    {   // Bigg freezing of cloud drops (:355-389) capped by the F23 budget (immersion_limit_rate :425-435) — BMT:1013-1034
        FT bn, bq;
        bigg_cloud<FT>(k, J_bigg, rho, q_lcl, n_lcl, N_lcl, T, bn, bq);
        const FT cap = T >= k.f23_T_freeze ? FT(0) : M::max(FT(0), inpc_kg - n_active) * k.inv_tau_act;
        const FT imm_n = M::min(bn, cap);
        const FT imm_q = bn > FT(0) ? bq * imm_n / bn : FT(0);
        dq_lcl -= imm_q; dn_lcl -= imm_n; dq_ice += imm_q; dn_ice += imm_n; dq_rim += imm_q; db_rim += imm_q * k.inv_rho_i;
    }
    {   // sublimation / deposition — BMT:1037-1054, _conv_q_vap_to_q_icl_const NonEq:168-193
        const FT L_s = k.LH_s0 + k.dcp_s * (T - k.T_0);
        const FT cp_air = k.cp_d + k.cpm_qt * q_tot + k.cpm_ql * (q_lcl + q_rai) + k.cpm_qi * q_ice;
        const FT dqsi_dT = qsi * (L_s / (k.R_v * (T * T)) - inv_T);
        const FT ts = k.tau_subdep * (FT(1) + (L_s / cp_air) * dqsi_dT);
        const FT excess = q_vap - qsi;
        FT sd = excess < FT(0) ? -M::min(-excess, q_ice) / ts : excess / ts;
        if (T > k.T_freeze_tps && sd > FT(0)) sd = FT(0);                         // INP limiter :56-58 and BMT:1045
        const FT n_per_q = q_ice > eps ? n_ice / q_ice : FT(0);
        dq_ice += sd;
        dn_ice += sd < FT(0) ? n_per_q * sd : FT(0);
        const FT sub = M::min(sd, FT(0));
        dq_rim += sub * s.F_rim;
        db_rim += s.rho_rim > FT(0) ? sub * s.F_rim / s.rho_rim : FT(0);
    }
    {   // ice number adjustment — BMT:1057-1064 (τ = 100 s, x ∈ [1e-12, 1e-5] kg), number_tendency_from_mass_limits CM2:882-891
        const FT target = q_ice < eps ? FT(0) : clampv(n_ice, q_ice * FT(1e5), q_ice * FT(1e12));
        dn_ice += (target - n_ice) * FT(0.01);
    }
    {   // Bigg freezing of rain — liquid_freezing_rate :274-311, BMT:1067-1075
        FT rn, rq;
        bigg_rain<FT, LIMITED>(k, J_bigg, rho, q_rai, n_rai, N_rai, T, rn, rq);
        dq_rai -= rq; dn_rai -= rn; dq_ice += rq; dn_ice += rn; dq_rim += rq; db_rim += rq * k.inv_rho_i;
    }
    d[0] = dq_lcl + poison; d[1] = dn_lcl + poison; d[2] = dq_rai + poison; d[3] = dn_rai + poison;
    d[4] = dq_ice + poison; d[5] = dn_ice + poison; d[6] = dq_rim + poison; d[7] = db_rim + poison;
}
template <typename FT> __device__ __forceinline__ void mp2m_p3_load(const FusedIO<FT> &io, int64_t i, FT (&in)[11], FT &shift) {
    const SegPos ps(io.lay, i);
    const FT *const col[11] = {io.rho, io.T, io.q_tot, io.q_lcl, io.n_lcl, io.q_rai, io.n_rai, io.q_ice, io.n_ice, io.q_rim, io.b_rim};
#pragma unroll
    for (int k = 0; k < 11; ++k) in[k] = ps.at(col[k], io.lay.s_in[k]);
    shift = io.shift ? ps.at(io.shift, io.lay.s_in[SEG_SHIFT]) : FT(0);
}

template <typename FT, bool LIMITED, bool INTPOW = false>
__global__ __launch_bounds__(kBlock) void mp2m_p3_pointwise_kernel(const SbConsts<FT> sc, const P3Consts<FT> c, const PointwiseConsts<FT> k,
                                                                  const FusedIO<FT> io, const int64_t n) {
    Math<FT>::prepare();   // Float64: exp2 / log2 tables → LDS (no-op for Float32)
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    FT in[11], shift, d[8];
    mp2m_p3_load<FT>(io, i, in, shift);
    mp2m_p3_point<FT, LIMITED, INTPOW>(sc, c, k, in, shift, d);
    const SegPos ps(io.lay, i);
#pragma unroll
    for (int q = 0; q < 8; ++q) ps.at(io.out[q], io.lay.s_out[q]) = d[q];
}

#ifndef CMX_COL_WAVES
#define CMX_COL_WAVES 3      // min waves per SIMD the register allocator must leave room for: 168 VGPRs at 3 waves (what the 50 KB of LDS per
                             // workgroup admit) beat 256 VGPRs at 2 (2M + P3, Float64: 29.9 → 28.0 ms per 1e6 states when measured in round 3,
                             // same-box A/B; Float32 unchanged; 1 wave: 46 ms)
#endif
// LDS per group, in FT units: quadrature copy is per block.  At the default order 16 a 32-state Float64 workgroup holds 50 KB, so THREE workgroups
// share a CU's 160 KB; twelve more values per state (tried in round 4: the raw inputs kept for the pointwise pass of the one-launch form instead
// of re-read through L2) make it 53.5 KB and two workgroups — 21.4 → 26.0 ms per 1e6 states, with FETCH_SIZE unchanged (the re-reads hit L2).
template <typename FT> struct ColLds {
    static __host__ __device__ __forceinline__ int per_group(int n) { return 6 * n + 100; }
};

// EXTRA: NoExtra, or — the ONE-launch form of the 2M + P3 entry — PointwiseExtra<FT>: the constants and the two extra columns of the pointwise part,
// which lane 0 of each group then evaluates in the epilogue (mp2m_p3_point) so that the eight tendency columns are written once instead of
// written by one kernel and read-modify-written by the next (47 → 20 column passes).  The kernel-argument segment is 4 KiB: the form is
// taken when the quadrature rule fits QuadSmall (order ≤ 32); larger rules run the two launches.
// XCD-aware tile order.  Workgroups are dispatched round-robin over the 8 XCDs (workgroup b runs on XCD b % 8) and every XCD has its own L2.  A tile
// of this kernel is only 8–32 states, i.e. 32–256 B of each column: with tile = b, the four to one tiles that share a 128-byte line would be read
// into four to one DIFFERENT L2s (the Float32 64-lane geometry measured 3.7 × the algorithmic HBM traffic that way).  Here XCD x gets the contiguous
// range [x·per, (x+1)·per) of tiles, per = gridDim.x / 8 (the launch rounds the grid up to a multiple of 8; surplus workgroups exit at once).
__device__ __forceinline__ int64_t xcd_tile(unsigned block, unsigned grid) {
    const unsigned per = grid >> 3;
    return (int64_t)(block & 7u) * per + (block >> 3);
}
struct NoExtra {};
template <typename FT, bool LIMITED_, bool INTPOW_> struct PointwiseExtra {      // the warm-rain instantiation is part of the type: one variant per kernel
    static constexpr bool LIMITED = LIMITED_, INTPOW = INTPOW_;
    SbConsts<FT> sc;
    PointwiseConsts<FT> pk;
    const FT *q_tot, *shift;
};
template <typename FT> struct QuadSmall { int32_t n; FT node[32], weight[32]; };
CMX_P3_CONTRACT_BEGIN      // the quadrature sweeps (cmx_p3.hpp); the pointwise part it calls in its epilogue (mp2m_p3_point, above) is defined outside and stays exact-as-written
template <typename FT, typename QUAD, bool ASPECT, bool FUSED, int GROUP, typename EXTRA = NoExtra>
__global__ __launch_bounds__(kBlock, CMX_COL_WAVES) void p3_collision_kernel(const P3Consts<FT> c, const P3VelConsts<FT> v, const P3ColConsts<FT> k,
                                                             const QUAD quad, const P3ColIO<FT> io, const int64_t n, const EXTRA ex) {
    constexpr bool ONE_LAUNCH = !std::is_same_v<EXTRA, NoExtra>;
    if (xcd_tile(blockIdx.x, gridDim.x) * (int64_t)(blockDim.x / GROUP) >= n) return;      // a surplus workgroup of the rounded-up grid (uniform: before any barrier)
    Math<FT>::prepare();   // Float64: exp2 / log2 tables → LDS (no-op for Float32)
    using P = PM<FT>;
    using M = Math<FT>;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    FT *lds = reinterpret_cast<FT *>(lds_raw);
    const int nq = quad.n;
    // block-wide copy of the quadrature rule: per-lane node indices need a memory the lanes can index
    FT *q_node = lds, *q_wt = lds + nq;
    __shared__ int64_t seg_tab[21];      // run strides of the segmented-column form (SegLayout): s_in[13], s_out[8]; unused for contiguous columns
    if (threadIdx.x == 0) {
        for (int j = 0; j < nq; ++j) { q_node[j] = quad.node[j]; q_wt[j] = quad.weight[j]; }
        if constexpr (FUSED) {
            if (io.lay.seg_len != 0) {
                for (int j = 0; j < 13; ++j) seg_tab[j] = io.lay.s_in[j];
                for (int j = 0; j < 8; ++j) seg_tab[13 + j] = io.lay.s_out[j];
            }
        }
    }
    const int grp = threadIdx.x / GROUP, g = threadIdx.x % GROUP;
    FT *G = lds + 2 * nq + grp * ColLds<FT>::per_group(nq);
    // per-group layout: E[48], Fm[24] at compile-time offsets from G, then one (D, v, weight) record per inner node — cloud nodes, then
    // rain nodes — so that a node's three values sit at immediate offsets of ONE address (separate arrays of run-time length nq cost a
    // pointer register each, eight in all, and most of them were spilled)
    // S[28]: the per-state constants of the rain part (curve exponents, size range, N₀, mean diameter) — needed once per outer node and in
    // the crossover solve — and the values only the epilogue needs (ρq, ρn, ρ_rim, 1/ρₐ, T), the five segment bounds of the
    // collision sweep, the quantiles of the self-collection / melting sweeps and ρ_g (the melting sweep's mass law), read back through a
    // volatile pointer so that they do not occupy twenty-five register pairs across all the sweeps
    FT *E = G, *Fm = G + 48, *S = G + 72, *cN = G + 100, *rN = G + 100 + 3 * nq;
    const volatile FT *Sv = S;
    const int64_t pt_raw = xcd_tile(blockIdx.x, gridDim.x) * (blockDim.x / GROUP) + grp;
    const bool valid = pt_raw < n;
    const int64_t i = valid ? pt_raw : n - 1;
    __syncthreads();

    // ---- per-point set-up (uniform over the group) ---------------------------------------------------------------
    // Order: the quantile solve first — it needs only ρq, ρn and log λ, and it is the register-hungriest piece of the set-up, so nothing else
    // is alive across it — then the state (p3_make_point) and the liquid-side loads.
    const SegPos ps(io.lay, i);       // (the other entries leave io.lay zero: contiguous columns)
    const FT rho_a = M::max(ps.at_tab(io.rho_a, seg_tab, SEG_RHO), FT(0));
    FT rho_q_in, rho_n_in;
    bool present;
    if constexpr (FUSED) {
        // clamp_to_nonneg and the volumetric quantities of BMT:912-932; ice processes only where q_ice > ϵₘ && n_ice > ϵₙ (:959)
        const FT q_ice = M::max(ps.at_tab(io.q_ice, seg_tab, SEG_QICE), FT(0)), n_ice = M::max(ps.at_tab(io.n_ice, seg_tab, SEG_NICE), FT(0));
        rho_q_in = q_ice * rho_a; rho_n_in = n_ice * rho_a;
        present = q_ice > P::eps() && n_ice > P::eps() && !(rho_n_in < P::eps() || rho_q_in < P::eps());
    } else {
        rho_q_in = io.rho_q[i]; rho_n_in = io.rho_n[i];
        present = !(rho_n_in < P::eps() || rho_q_in < P::eps());
    }
    const FT loglam = present ? ps.at_tab(io.loglam, seg_tab, SEG_LOGLAM) : FT(10), lam = P::exp(loglam), mu = p3_mu<FT>(c, loglam);
    // quantiles of the ice PSD (integral_bounds, P3_integral_properties.jl:34-46): one Halley solve per LANE — lanes 0/1 the
    // collision bounds (p = 1e-5), 2/3 the self-collection bounds (p = eps), 4/5 the melting bounds (p = 1e-6) — shared by shuffles
    // (the 2M+P3 entry parks the self-collection / melting bounds in S[20…23] until their sweeps)
    FT D_min, D_max;
    {
        FT plev = (g & 1) ? v.p_hi : v.p_lo;
        if constexpr (FUSED) {
            if (g == 2) plev = P::eps();
            if (g == 3) plev = FT(1) - P::eps();
            if (g == 4) plev = k.p_lo_m;
            if (g == 5) plev = k.p_hi_m;
        }
        const FT xq = gamma_inc_inv_dev<FT>(mu + FT(1), plev, FT(1) - plev) / lam;
        D_min = __shfl(xq, 0, GROUP); D_max = __shfl(xq, 1, GROUP);
        if constexpr (FUSED) {
            if (g >= 2 && g < 6) S[18 + g] = xq;
        }
    }
    P3Point<FT> s;
    FT L_c, N_c, L_r, N_r;
    const FT T = ps.at_tab(io.T, seg_tab, SEG_T);
    if constexpr (FUSED) {
        L_c = M::max(ps.at_tab(io.q_lcl, seg_tab, SEG_QLCL), FT(0)) * rho_a; N_c = M::max(ps.at_tab(io.n_lcl, seg_tab, SEG_NLCL), FT(0)) * rho_a;
        L_r = M::max(ps.at_tab(io.q_rai, seg_tab, SEG_QRAI), FT(0)) * rho_a; N_r = M::max(ps.at_tab(io.n_rai, seg_tab, SEG_NRAI), FT(0)) * rho_a;
        p3_make_point<FT>(c, rho_q_in, rho_n_in, M::max(ps.at_tab(io.q_rim, seg_tab, SEG_QRIM), FT(0)) * rho_a,
                          M::max(ps.at_tab(io.b_rim, seg_tab, SEG_BRIM), FT(0)) * rho_a, s);
        if (g == 0) { S[24] = s.rho_g; S[25] = s.bnd[1]; S[26] = s.bnd[2]; S[27] = s.bnd[3]; }   // for the two later sweeps' segment bounds
    } else {
        L_c = io.L_c[i]; N_c = io.N_c[i]; L_r = io.L_r[i]; N_r = io.N_r[i];
        p3_make_point<FT>(c, rho_q_in, rho_n_in, io.x3[i], io.x4[i], s);
    }
    const FT logN0 = P::log(present ? s.rho_n : FT(1)) - (-(mu + FT(1)) * loglam + P::lgamma(mu + FT(1)));
    const FT lra = P::log(rho_a);
    const FT sb = v.s_B + rho_a * v.s_C, se = v.s_A * lra + sb * v.ln1000;
    const FT le1 = v.l_A * lra, le2 = le1 + v.l_H * rho_a;
    if (g == 0) {
        S[15] = D_min; S[19] = D_max;
#pragma unroll
        for (int q = 1; q < 4; ++q) S[15 + q] = M::min(M::max(s.bnd[q], D_min), D_max);
    }
    const FT Fu = M::max(FT(1) - s.F_rim, P::eps());
    // (h0 carries the −½ ln π of the partially rimed regime's area^(−½) = y/√π, y = 1/√(area/π): eval_ice below)
    const FT h0 = (v.h0_num - P::log(Fu)) / FT(3) - FT(0.5723649429247001), h1 = c.beta_va / FT(3);
    const bool unrimed = s.F_rim == FT(0);
    const FT pi = FT(3.14159265358979323846), inv_pi = FT(0.3183098861837907);
    // ice fall speed (incl. aspect factor), collision radius and number density at diameter x — as in p3_self_collection_kernel
    const typename P::Coefs kc = P::coefs();          // exp / log constants pinned in VGPRs for the node loops
    // `kk`: the coefficient set of the caller's phase (P::Coefs in the collision sweep, P::LocalCoefs in the self-collection / melting sweeps)
    // SHARED: D^(σ/2) serves BOTH non-spherical laws (the mixed area's γ D^σ is its square) — a wave whose lanes sit in both regimes, i.e. most waves of
    // the inner self-collection loop, evaluates one exponential instead of one per regime (round 5).  The collision sweep calls the form with one
    // exponential per regime: one evaluation per OUTER node there, and the shared value would be two more live registers at the kernel's register peak.
    // `ik`: the area-law / aspect constants the regimes read — kernel arguments (SGPR pairs) in the collision sweep, register copies pinned for the fused sweeps
    // (IceK below: six fewer scalar pairs in the inner self-collection loop, whose scalar file is over-subscribed — spilled SGPRs are re-read with v_readlane)
    struct IceK { FT half_sigma, sqrt_gamma_pi, pi_4, gamma_area, g0, g1, bnd1; };      // bnd1 = s.bnd[1] = D_th, the one regime threshold that is a kernel constant
    const IceK ik0{v.half_sigma, v.sqrt_gamma_pi, v.pi_4, v.gamma_area, v.g0, v.g1, s.bnd[1]};
    auto eval_ice = [&](FT x, FT &vv, FT &rr, FT &nn, const auto &kk, auto shared, const IceK &ik) {
        constexpr bool SHARED = decltype(shared)::value;
        const FT logD = P::log_pos(x, kk);          // an interior quadrature node: positive, normal, finite
        const int reg = x < ik.bnd1 ? 0 : (unrimed ? 1 : (x < s.bnd[2] ? 1 : (x < s.bnd[3] ? 2 : 3)));
        // collision radius r = √(area/π) and aspect factor: spherical regimes r = D/2 exactly; unrimed non-spherical area = γ D^σ:
        // r = √(γ/π)·D^(σ/2) (one exponential, no square root); only the partially rimed regime needs the mixed area and its root
        // aspect factor = exp(eA)·mA.  Partially rimed regime: the factor carries area^(−½), and √(area/π) is formed for the collision radius anyway — its
        // reciprocal root y is the Newton iterate that square root is made of, so area^(−½) = y/√π (the 1/√π folded into h0) is free where eA −= ½ ln(area) cost a
        // logarithm (≈ 20 Float64 instructions per node of every wave with a lane in that regime; round 5)
        FT eA = FT(0), mA = FT(1);
        rr = FT(0.5) * x;
        if constexpr (SHARED) {
            if (reg == 1 || reg == 3) {
                const FT dh = P::exp(ik.half_sigma * logD, kk);
                if (reg == 1) {
                    rr = ik.sqrt_gamma_pi * dh;
                    if (ASPECT) eA = ik.g0 + ik.g1 * logD;
                } else {
                    const FT area = s.F_rim * (ik.pi_4 * x * x) + (FT(1) - s.F_rim) * (ik.gamma_area * (dh * dh));
                    const FT ap = area * inv_pi;
                    if (ASPECT) { const FT y = M::rsqrt_pos(ap); rr = ap * y; mA = y; eA = h0 + h1 * logD; }
                    else rr = M::sqrt(ap);
                }
            }
        } else if (reg == 1) {
            rr = ik.sqrt_gamma_pi * P::exp(ik.half_sigma * logD, kk);
            if (ASPECT) eA = ik.g0 + ik.g1 * logD;
        } else if (reg == 3) {
            const FT area = s.F_rim * (ik.pi_4 * x * x) + (FT(1) - s.F_rim) * (ik.gamma_area * P::exp(v.sigma_area * logD, kk));
            const FT ap = area * inv_pi;
            if (ASPECT) { const FT y = M::rsqrt_pos(ap); rr = ap * y; mA = y; eA = h0 + h1 * logD; }
            else rr = M::sqrt(ap);
        }
        const bool small = x <= v.cutoff;
        const FT E1 = small ? se + sb * logD : le1 + v.l_b1 * logD;
        const FT dE = small ? -v.s_c2 * x : (le2 - le1) + (v.l_b2 - v.l_b1) * logD - v.l_c2 * x;
        const FT A1 = small ? kpin(v.s_E) : kpin(v.l_a1), A2 = small ? kpin(v.s_F) : kpin(v.l_a2);
        const FT S = A1 + A2 * P::exp(dE, kk);
        vv = ASPECT ? P::exp(eA + E1, kk) * (mA * S) : P::exp(E1, kk) * S;      // (without the aspect factor: no 0 + E1, no 1·S)
        nn = P::exp(logN0 + mu * logD - lam * x, kk);
    };
    // rain Chen-2022 curve at ρₐ
    FT re[3], rb[3];
    {
        const FT q0 = k.r_rho0 * rho_a;
#pragma unroll
        for (int j = 0; j < 3; ++j) { rb[j] = k.r_b[j] - k.r_brho * rho_a; re[j] = q0 + rb[j] * v.ln1000; }
        re[2] += k.r_a3pow * lra;
    }
    auto v_liq = [&](FT D, FT logD) {
        return k.r_a[0] * P::exp(re[0] + rb[0] * logD - k.r_c[0] * D, kc) + k.r_a[1] * P::exp(re[1] + rb[1] * logD - k.r_c[1] * D, kc) +
               k.r_a[2] * P::exp(re[2] + rb[2] * logD - k.r_c[2] * D, kc);
    };
    // Float64 reads the rain constants back from S inside the sweep (they would otherwise hold ten register pairs: scratch spills at three
    // waves per SIMD); Float32 has the registers and keeps them there — the LDS round trip in the crossover solve's dependent chain cost
    // 3 % of the 2M + P3 step (same-box A/B, round 3: 7.54 -> 7.76 ms per 1e6 states)
    constexpr bool PARK_RAIN = sizeof(FT) == 8;
    auto v_liq_s = [&](FT D, FT logD) {      // the same with the exponents read from S (the crossover solve)
        if constexpr (!PARK_RAIN) return v_liq(D, logD);
        else return k.r_a[0] * P::exp(Sv[0] + Sv[1] * logD - k.r_c[0] * D, kc) + k.r_a[1] * P::exp(Sv[2] + Sv[3] * logD - k.r_c[1] * D, kc) +
               k.r_a[2] * P::exp(Sv[4] + Sv[5] * logD - k.r_c[2] * D, kc);
    };
    // cloud PSD in diameter — log_pdf_cloud_parameters_mass CM2:172-188, pdf_cloud_parameters :227-236
    const FT inv_rho = FT(1) / rho_a;
    const FT q_c = L_c * inv_rho, q_r = L_r * inv_rho;
    if (g == 0) { S[10] = s.rho_q; S[11] = s.rho_n; S[12] = s.rho_rim; S[13] = inv_rho; S[14] = T; }   // published by the barrier below
    const bool melts = T > k.T_freeze_tps;   // BMT:980
    const bool no_cloud = N_c < P::eps() || q_c < P::eps();
    FT logN0c, lam_c, c_lo = FT(0), c_hi = FT(0);
    {
        const FT sq = M::max(q_c, P::eps()), sN = M::max(N_c, P::eps());
        const FT logx = P::log(rho_a * sq / sN);
        const FT logB = -k.mu_c * (logx + k.lg_z1 - k.lg_z2);
        const FT logA = k.log_mu_c + P::log(sN) + k.z1 * logB - k.lg_z1;
        logN0c = logA + k.logN0_shift;
        const FT log_lam_c = logB + k.log_km_mu;
        lam_c = P::exp(log_lam_c);
        if (!no_cloud) { c_lo = P::exp((k.log_zq_lo - log_lam_c) * k.inv_mu_cD); c_hi = P::exp((k.log_zq_hi - log_lam_c) * k.inv_mu_cD); }
    }
    // rain PSD — pdf_rain_parameters CM2:67-110
    FT N0r, lam_r;
    {
        const FT sq = M::max(q_r, P::eps()), sN = M::max(N_r, P::eps());
        const FT L = rho_a * sq;
        bool gate;
        if (!k.limited) {
            lam_r = P::exp(P::log(k.pi_rho_w / (L / sN)) / FT(3)); N0r = lam_r * sN;
            gate = N_r < P::eps() || q_r < P::eps();
        } else {
            const FT xt = M::min(M::max(L / sN, k.xr_min), k.xr_max);
            N0r = M::min(M::max(sN * P::exp(P::log(k.pi_rho_w / xt) / FT(3)), k.N0_min), k.N0_max);
            lam_r = M::min(M::max(M::sqrt(M::sqrt(k.pi_rho_w * N0r / L)), k.lam_min), k.lam_max);
            gate = N_r < P::eps() && q_r < P::eps();
        }
        if (gate) { N0r = FT(0); lam_r = FT(0); }
    }
    const FT Dr_mean = lam_r > FT(0) ? FT(1) / lam_r : FT(0);
    const FT r_lo = Dr_mean * k.k_lo, r_hi = Dr_mean * k.k_hi;
    const bool has_cloud = present && c_lo < c_hi, has_rain = present && N0r != FT(0) && r_hi > r_lo;
    // compute_max_freeze_rate — :167-201
    const FT T_C = T - k.T_freeze_p3, inv_2TC = FT(1000000) / (FT(2) * T_C);
    const bool above_freezing = T >= k.T_freeze_tps;
    FT mfr_fac;
    bool freeze_all;                 // Musil denominator ≤ 0 (T ≲ 220 K): floatmax in the reference, i.e. f_frz = 1
    {
        const FT L_v = k.LH_v0 + k.dcp_v * (T - k.T_0), L_f = k.LH_f0 + k.dcp_f * (T - k.T_0);
        const FT dT = k.T_freeze_tps - T;
        const FT ps = k.press_tr * P::exp(k.ps_pow * P::log(T * k.inv_T_tr) + k.ps_b * (k.inv_T_tr - FT(1) / T));
        const FT drho_v = k.qsi_frz - ps / (k.R_v * T);
        const FT denom = L_f - k.cp_l * dT;
        freeze_all = !(denom > FT(0));
        mfr_fac = FT(2) * pi * (k.K_therm * dT + L_v * k.D_vapor * drho_v) / denom;
    }

    // ---- inner-node caches: lane j ↔ inner node j -------------------------------------------------------------------
    if (has_cloud) {
        const FT sc = (c_hi - c_lo) / FT(2), sh = (c_lo + c_hi) / FT(2);
        for (int j = g; j < nq; j += GROUP) {
            const FT D = sc * q_node[j] + sh, logD = P::log(D);
            const FT nD = P::exp(logN0c + k.nu_cD * logD - lam_c * P::exp(k.mu_cD * logD));
            cN[3 * j] = D; cN[3 * j + 1] = v_liq(D, logD); cN[3 * j + 2] = q_wt[j] * sc * nD;
        }
    }
    if (has_rain) {
        if (g == 0) {
#pragma unroll
            for (int j = 0; j < 3; ++j) { S[2 * j] = re[j]; S[2 * j + 1] = rb[j]; }
            S[6] = r_lo; S[7] = r_hi; S[8] = Dr_mean; S[9] = N0r;
        }
        const FT sc = (r_hi - r_lo) / FT(2), sh = (r_lo + r_hi) / FT(2);
        for (int j = g; j < nq; j += GROUP) {
            const FT D = sc * q_node[j] + sh, logD = P::log(D);
            rN[3 * j] = D; rN[3 * j + 1] = v_liq(D, logD);
            rN[3 * j + 2] = q_wt[j] * sc * (N0r * P::exp(-D * lam_r)) * (k.m_fac * (D * D * D));
        }
        if (g < 8) {
            // families of the closed form (closed_rain_inner_NM :343-369): (α, z₀) = (λ, 1) for the v_i term and (λ + c_j, 1 + b_j)
            // for the three terms of the rain curve.  lane ↔ (family f, end-point): the end-point incomplete gammas → E, and
            // the family record F[f] = (α, z₀, Γ(z₀), w, λ/α, (λ/α)^z₀) with w = −a_j e^{e_j} λ^{−b_j} (the j-th term of v_l at the
            // mean diameter): all powers of α are taken relative to λ so that nothing leaves the Float32 range
            const int f = g >> 1;
            const FT bj = f == 0 ? FT(0) : (f == 1 ? rb[0] : (f == 2 ? rb[1] : rb[2]));
            const FT cj = f == 0 ? FT(0) : (f == 1 ? k.r_c[0] : (f == 2 ? k.r_c[1] : k.r_c[2]));
            const FT aj = f == 0 ? FT(0) : (f == 1 ? k.r_a[0] : (f == 2 ? k.r_a[1] : k.r_a[2]));
            const FT ej = f == 0 ? FT(0) : (f == 1 ? re[0] : (f == 2 ? re[1] : re[2]));
            const FT al = lam_r + cj, z0 = FT(1) + bj;
            const FT G0 = f == 0 ? FT(1) : P::exp(P::lgamma(z0));
            FT gg[6];
            lower_gamma6<FT>(z0, al * ((g & 1) ? r_hi : r_lo), G0, gg);
#pragma unroll
            for (int m = 0; m < 6; ++m) E[g * 6 + m] = gg[m];
            if (!(g & 1)) {
                const FT rr = lam_r / al;
                Fm[f * 6 + 0] = al; Fm[f * 6 + 1] = z0; Fm[f * 6 + 2] = G0; Fm[f * 6 + 3] = -aj * P::exp(ej - bj * P::log(lam_r));
                Fm[f * 6 + 4] = rr; Fm[f * 6 + 5] = P::exp(z0 * P::log(rr));
            }
        }
    }
    __syncthreads();

    // ---- outer nodes: lane ↔ node of the current ice segment ---------------------------------------------------------
    FT acc[10];
#pragma unroll
    for (int q = 0; q < 10; ++q) acc[q] = FT(0);
    if (present) {
        for (int sg = 0; sg < 4; ++sg) {
            const FT a = Sv[15 + sg], b = Sv[16 + sg];
            if (!(a < b)) continue;
            const FT scale = (b - a) / FT(2), shift = (a + b) / FT(2);
            for (int o = g; o < nq; o += GROUP) {
                const FT Di = scale * q_node[o] + shift, w = q_wt[o] * scale;
                FT v_i, r_i, n_i;
                eval_ice(Di, v_i, r_i, n_i, kc, std::false_type{}, ik0);
                const FT K0 = pi * (r_i * r_i), K1 = pi * r_i, K2 = FT(0.7853981633974483);
                FT Nc = FT(0), Mc = FT(0), Bc = FT(0), Nr = FT(0), Mr = FT(0), Br = FT(0);
                if (has_cloud) {
                    for (int j = 0; j < nq; ++j) {
                        const FT D = cN[3 * j], dv = P::abs(v_i - cN[3 * j + 1]);
                        const FT t1 = M::fma(D, M::fma(D, K2, K1), K0) * dv * cN[3 * j + 2];
                        const FT t2 = t1 * (k.m_fac * (D * D * D));
                        const FT Ri = M::min(M::max(D * dv * inv_2TC, FT(1)), FT(12));
                        const FT rho_p = Ri <= FT(8) ? k.rime_a + k.rime_b * Ri + k.rime_c * (Ri * Ri)
                                                     : k.rime_rho8 + (Ri - FT(8)) * FT(0.25) * (k.rime_rho_ice - k.rime_rho8);
                        Nc += t1; Mc += t2; Bc += t2 * P::rcp(rho_p);
                    }
                }
                if (has_rain) {
                    // crossover_diameter — :325-334: Brent on v_l(D) − v_i over [r_lo, r_hi], fixed iteration budget
                    FT Dstar;
                    {
                        FT xa = PARK_RAIN ? Sv[6] : r_lo, xb = PARK_RAIN ? Sv[7] : r_hi;
                        FT fa = v_liq_s(xa, P::log(xa, kc)) - v_i, fb = v_liq_s(xb, P::log(xb, kc)) - v_i;
                        if (!(fa * fb <= FT(0))) Dstar = P::abs(fa) <= P::abs(fb) ? xa : xb;
                        else {
                            Zeroin<FT> z;                       // cmx_p3.hpp: Brent's zeroin under the fixed evaluation budget
                            z.start(xa, xb, fa, fb);
                            bool live = true;
                            for (int it = 0; it < k.brent_iters; ++it) {
                                if (!z.order()) { live = false; break; }
                                const FT sx = z.propose();
                                z.accept(v_liq_s(sx, P::log(sx, kc)) - v_i);
                            }
                            if (live) z.order();
                            Dstar = z.b;
                        }
                    }
                    // crossing(p) = Σ_f coef_f Σ_i K_i α_f^{−z} [2γ(z, α_f D*) − γ(z, α_f D_lo) − γ(z, α_f D_hi)],  z = z₀_f + p + i,
                    // with every α^{−z} written as λ^{−z} (λ/α)^z: N = N₀r/λ · S₀, M = N₀r/λ · ρ_w π/6 · D̄³ · S₃
                    FT S0 = FT(0), S3 = FT(0);
                    const FT Dr_m = PARK_RAIN ? Sv[8] : Dr_mean;
                    const FT Kt1 = K1 * Dr_m, Kt2 = K2 * (Dr_m * Dr_m);
#pragma unroll 1
                    for (int f = 0; f < 4; ++f) {
                        const FT al = Fm[f * 6 + 0], z0 = Fm[f * 6 + 1], G0 = Fm[f * 6 + 2], rr = Fm[f * 6 + 4];
                        const FT wq = f == 0 ? v_i : Fm[f * 6 + 3];
                        FT pw = Fm[f * 6 + 5];
                        FT gs[6], I[6];
                        lower_gamma6<FT>(z0, al * Dstar, G0, gs);
#pragma unroll
                        for (int m = 0; m < 6; ++m) {
                            I[m] = pw * (FT(2) * gs[m] - E[(2 * f) * 6 + m] - E[(2 * f + 1) * 6 + m]);
                            pw *= rr;
                        }
                        S0 += wq * (K0 * I[0] + Kt1 * I[1] + Kt2 * I[2]);
                        S3 += wq * (K0 * I[3] + Kt1 * I[4] + Kt2 * I[5]);
                    }
                    const FT N0_lam = (PARK_RAIN ? Sv[9] : N0r) * Dr_m;
                    Nr = N0_lam * S0; Mr = N0_lam * (k.m_fac * (Dr_m * Dr_m * Dr_m)) * S3;
                    if (!(isfinite(Nr) && isfinite(Mr))) { Nr = FT(0); Mr = FT(0); }
                    else {
                        for (int j = 0; j < nq; ++j) {
                            const FT D = rN[3 * j], dv = P::abs(v_i - rN[3 * j + 1]);
                            const FT t2 = M::fma(D, M::fma(D, K2, K1), K0) * dv * rN[3 * j + 2];
                            const FT Ri = M::min(M::max(D * dv * inv_2TC, FT(1)), FT(12));
                            const FT rho_p = Ri <= FT(8) ? k.rime_a + k.rime_b * Ri + k.rime_c * (Ri * Ri)
                                                         : k.rime_rho8 + (Ri - FT(8)) * FT(0.25) * (k.rime_rho_ice - k.rime_rho8);
                            Br += t2 * P::rcp(rho_p);
                        }
                    }
                }
                // outer integrand — :451-486
                const FT M_col = Mc + Mr;
                FT M_max;
                if (above_freezing) M_max = FT(0);
                else if (freeze_all) M_max = M_col;                                        // min(M_col, floatmax)
                else M_max = mfr_fac * Di * (k.vent_a + k.vent_bc * M::sqrt(M::max(Di * v_i, FT(0))));
                const FT M_frz = M::min(M_col, M_max);
                const FT f_frz = M_col == FT(0) ? FT(0) : M_frz / M_col;
                const FT nw = n_i * w;
                acc[0] += nw * Mc * f_frz; acc[1] += nw * Mc * (FT(1) - f_frz); acc[2] += nw * Nc;
                acc[3] += nw * Mr * f_frz; acc[4] += nw * Mr * (FT(1) - f_frz); acc[5] += nw * Nr;
                if constexpr (FUSED) {
                    // the 2M + P3 entry does not report the ten rates: Σ n w M_col is the sum of the four mass sums and the two rime-volume
                    // sums are only ever added — eight accumulators instead of ten
                    acc[7] += nw * (Bc + Br) * f_frz;
                } else {
                    acc[6] += nw * M_col;      acc[7] += nw * Bc * f_frz;           acc[8] += nw * Br * f_frz;
                }
                acc[9] += M_col > M_frz ? nw * M_col : FT(0);
            }
        }
    }
    // ---- 2M+P3 entry only: aggregation (ice_self_collection :676-712) and melting (ice_melt :64-94), outer node per lane ------
    FT acc_sc = FT(0), acc_m = FT(0);
    if constexpr (FUSED) {
        const FT D_lo_sc = Sv[20], D_hi_sc = Sv[21];
        // these two sweeps run far below the kernel's register count (≈ 107 of 163–168 VGPRs): their integrands read the second polynomial coefficient
        // of exp / log from register pairs pinned HERE (cmx_p3.hpp coefs_local) — 5 v_mov_b64 fewer per inner node of the self-collection integral
        const typename P::LocalCoefs kl = P::coefs_local(D_lo_sc);
        IceK ikp = ik0;
        if constexpr (CMX_P3_LOCAL_COEFS) { P::pin(ikp.half_sigma); P::pin(ikp.sqrt_gamma_pi); P::pin(ikp.pi_4); P::pin(ikp.gamma_area); P::pin(ikp.g0); P::pin(ikp.g1); P::pin(ikp.bnd1); }
        if (present) {
            FT bs[5];
            bs[0] = D_lo_sc; bs[4] = D_hi_sc;
#pragma unroll
            for (int q = 1; q < 4; ++q) bs[q] = M::min(M::max(Sv[24 + q], D_lo_sc), D_hi_sc);
            for (int sg = 0; sg < 4; ++sg) {
                const FT a = bs[sg], b = bs[sg + 1];
                if (!(a < b)) continue;
                const FT scale = (b - a) / FT(2), shift = (a + b) / FT(2);
                for (int o = g; o < nq; o += GROUP) {
                    const FT D1 = scale * q_node[o] + shift;
                    FT v1, r1, n1;
                    eval_ice(D1, v1, r1, n1, kl, std::true_type{}, ikp);
                    FT inner = FT(0);
                    for (int h = 0; h < 2; ++h) {   // inner integral split at the |v₁ − v₂| cusp D₂ = D₁
                        const FT ia = h == 0 ? D_lo_sc : D1, ib = h == 0 ? D1 : D_hi_sc;
                        if (!(ia < ib)) continue;
                        const FT sc2 = (ib - ia) / FT(2), sh2 = (ia + ib) / FT(2);
                        FT r_in = FT(0);
                        for (int j = 0; j < nq; ++j) {
                            FT v2, r2, n2;
                            eval_ice(sc2 * q_node[j] + sh2, v2, r2, n2, kl, std::true_type{}, ikp);
                            const FT rs = r1 + r2;
                            r_in += rs * rs * P::abs(v1 - v2) * n2 * q_wt[j];
                        }
                        inner += sc2 * r_in;
                    }
                    acc_sc += inner * n1 * (q_wt[o] * scale);
                }
            }
            if (melts) {
                const FT D_lo_m = Sv[22], D_hi_m = Sv[23];
                s.rho_g = Sv[24];
                // segments one at a time (not unrolled: each segment's mass law is derived inside its iteration, see p3_segment_mass_law);
                // the thresholds are picked with selects — a run-time index into a local array would put the array into scratch
#pragma unroll 1
                for (int sg = 0; sg < 4; ++sg) {
                    const FT t_lo = sg == 0 ? FT(0) : Sv[24 + sg];
                    const FT t_hi = sg == 3 ? FT(INFINITY) : Sv[25 + sg];
                    const FT a = sg == 0 ? D_lo_m : M::min(M::max(t_lo, D_lo_m), D_hi_m);
                    const FT b = sg == 3 ? D_hi_m : M::min(M::max(t_hi, D_lo_m), D_hi_m);
                    if (!(a < b)) continue;
                    const FT scale = (b - a) / FT(2), shift = (a + b) / FT(2);
                    FT ma, mb;
                    p3_mass_law_at<FT>(c, s, (t_lo + t_hi) / FT(2), ma, mb);   // re-derived here: P3Point::log_a / b are not kept across the sweeps
                    for (int o = g; o < nq; o += GROUP) {
                        const FT x = scale * q_node[o] + shift;
                        FT vD, rD_, nD;
                        eval_ice(x, vD, rD_, nD, kl, std::true_type{}, ikp);
                        const FT Fv = k.vent_a + k.vent_bc * M::sqrt(M::max(x * vD, FT(0)));
                        const FT dm_over_D = mb == FT(3) ? ma * mb * x : ma * mb * P::exp((mb - FT(2)) * P::log(x));   // ∂m/∂D / D
                        acc_m += dm_over_D * Fv * nD * (q_wt[o] * scale);
                    }
                }
            }
        }
    }
    // ---- group reduction + bulk sources (:600-655) -----------------------------------------------------------------------
#pragma unroll
    for (int q = 0; q < 10; ++q) {
        FT x = acc[q];
#pragma unroll
        for (int d = GROUP / 2; d >= 1; d >>= 1) x += __shfl_xor(x, d, GROUP);
        acc[q] = x;
    }
    if constexpr (FUSED) {
#pragma unroll
        for (int d = GROUP / 2; d >= 1; d >>= 1) { acc_sc += __shfl_xor(acc_sc, d, GROUP); acc_m += __shfl_xor(acc_m, d, GROUP); }
    }
    if constexpr (FUSED) acc[6] = (acc[0] + acc[1]) + (acc[3] + acc[4]);
    // the state's index once more, from an opaque copy of the lane number: two integer instructions here instead of a register pair
    // (or a scratch slot) held across the sweeps
    int lane = threadIdx.x;
    asm volatile("" : "+v"(lane));
    const int64_t i_out = xcd_tile(blockIdx.x, gridDim.x) * (blockDim.x / GROUP) + lane / GROUP;
    if (g == 0 && i_out < n) {
        const int64_t i = i_out;
        const FT e_rho_q = Sv[10], e_rho_n = Sv[11], e_rho_rim = Sv[12], e_inv_rho = Sv[13], e_T = Sv[14];
        const FT f_wet = acc[6] == FT(0) ? FT(0) : acc[9] / acc[6];
        const FT NRSHD = acc[4] * k.inv_m_shd;
        const FT B_rim = e_rho_rim == FT(0) ? FT(0) : (e_rho_q * s.F_rim) / e_rho_rim;
        const FT QIWET = present ? f_wet * e_rho_q * (FT(1) - s.F_rim) / k.tau_wet : FT(0);
        const FT BIWET = present ? f_wet * (e_rho_q / k.rho_i - B_rim) / k.tau_wet : FT(0);
        const FT o[7] = {(-acc[0] - acc[1]) * e_inv_rho, (-acc[3] + acc[1]) * e_inv_rho, -acc[2], -acc[5] + NRSHD,
                         acc[0] + acc[3] + QIWET, acc[0] + acc[3], acc[7] + acc[8] + BIWET};
        if constexpr (!FUSED) {
#pragma unroll
            for (int q = 0; q < 10; ++q)
                if (io.rates[q]) io.rates[q][i] = acc[q];
#pragma unroll
            for (int q = 0; q < 7; ++q)
                if (io.src[q]) io.src[q][i] = o[q];
        } else if (present || ONE_LAUNCH) {
            // BMT:966-994: collisions, aggregation (½π factored out of the sums), melting (ice → rain; rime drains in proportion)
            const FT agg = FT(0.5) * pi * acc_sc;
            const FT L_f = k.LH_f0 + k.dcp_f * (e_T - k.T_0);
            const FT mL = melts ? M::max(FT(0), k.K4 / L_f * (e_T - k.T_freeze_p3) * acc_m) : FT(0);
            const FT mq = mL * e_inv_rho, mn = (e_rho_n / e_rho_q * mL) * e_inv_rho;
            FT d[8] = {o[0], o[2] * e_inv_rho, o[1] + mq, o[3] * e_inv_rho + mn, o[5] * e_inv_rho - mq, -agg * e_inv_rho - mn,
                       o[4] * e_inv_rho - mq * s.F_rim, o[6] * e_inv_rho - (e_rho_rim > FT(0) ? mq * s.F_rim / e_rho_rim : FT(0))};
            if constexpr (!ONE_LAUNCH) {
#pragma unroll
                for (int q = 0; q < 8; ++q) SegPos(io.lay, i).at_tab(io.out[q], seg_tab, 13 + q) += d[q];
            } else {
                // park the ice-process sums in the state's LDS block (the rain constants there are dead now) for the pointwise pass below
#pragma unroll
                for (int q = 0; q < 8; ++q) S[q] = present ? d[q] : FT(0);
            }
        }
    }
    if constexpr (ONE_LAUNCH) {
        // The pointwise part of the entry for the workgroup's states, one state per LANE of the first wave(s) instead of one per group: lane 0
        // of a group doing it would run the ≈ 1 500 instructions with 8 of 64 lanes active in every wave (+1.1 % Float64, +2.5 % Float32 on the
        // 2M + P3 step, same-box A/B); here 32 states share one pass of them.  Each tendency column is written once.
        __syncthreads();
        const int nst = blockDim.x / GROUP;
        const int64_t i2 = xcd_tile(blockIdx.x, gridDim.x) * nst + threadIdx.x;
        if ((int)threadIdx.x < nst && i2 < n) {
            const volatile FT *S2 = lds + 2 * nq + threadIdx.x * ColLds<FT>::per_group(nq) + 72;
            FT in[11], pw[8];
            const SegPos p2(io.lay, i2);
            const FT *const col[11] = {io.rho_a, io.T, ex.q_tot, io.q_lcl, io.n_lcl, io.q_rai, io.n_rai, io.q_ice, io.n_ice, io.q_rim, io.b_rim};
#pragma unroll
            for (int kk = 0; kk < 11; ++kk) in[kk] = p2.at_tab(col[kk], seg_tab, kk);
            const FT shift = ex.shift ? p2.at_tab(ex.shift, seg_tab, SEG_SHIFT) : FT(0);
            if constexpr (sizeof(FT) == 8 && CMX_PHASE_CONSTS) {
                // Float64: the pointwise constants (≈ 150 doubles) are read THROUGH the kernel-argument segment and only now — taken from the by-value
                // argument the compiler loads them in the entry block and parks them in VGPR lanes across all the sweeps (cmx_math.hpp consts_after)
                // (its offset comes from THIS kernel's parameter list — kernarg_offset_of_last, cmx_launch.hpp — not from a hand-kept mirror struct)
                constexpr size_t ex_off = kernarg_offset_of_last<EXTRA, decltype(&p3_collision_kernel<FT, QUAD, ASPECT, FUSED, GROUP, EXTRA>)>();
                const auto *base = (const __attribute__((address_space(4))) unsigned char *)__builtin_amdgcn_kernarg_segment_ptr();
                KernArg<EXTRA> &exk = *(KernArg<EXTRA> *)(base + ex_off);
                const auto &exl = consts_after(exk, in[0]);
                mp2m_p3_point<FT, EXTRA::LIMITED, EXTRA::INTPOW, true>(exl.sc, c, exl.pk, in, shift, pw);
            } else
                mp2m_p3_point<FT, EXTRA::LIMITED, EXTRA::INTPOW, true>(ex.sc, c, ex.pk, in, shift, pw);
#pragma unroll
            for (int q = 0; q < 8; ++q) p2.at_tab(io.out[q], seg_tab, 13 + q) = pw[q] + S2[q];
        }
    }
}
CMX_P3_CONTRACT_END

// launch geometry: 256 lanes (32 states) per workgroup unless the per-state LDS caches (6n + 100 values) would not fit — then 128
// Returns false when the rounded-up tile count does not fit HIP's 2^31 − 1 workgroups: a Float32 tile is only 8 (group 8) or 4 (group 16) states,
// so that can happen below kMaxPoints (the cast to unsigned truncated it silently); the callers return CMX_ERR_UNSUPPORTED.
template <typename FT> static bool collision_geometry(int group, int nq, int64_t n, dim3 &grid, dim3 &block, size_t &lds) {
    // lanes per workgroup to start from (the kernel reads blockDim.x): Float64 256 — 128 or 64 lanes cost 15–17 % (21.4 → 24.6 / 25.0 ms per 1e6 states
    // of the 2M + P3 entry: the pointwise pass then runs with 16 or 8 of a wave's 64 lanes) — Float32 64: 7.14 → 6.86 ms (same-box A/B, round 4,
    // profiles/r04_ab_sessions.txt, session 19).  -DCMX_COL_THREADS=n forces one size for both.
#ifdef CMX_COL_THREADS
    int threads = CMX_COL_THREADS;
#else
    int threads = sizeof(FT) == 4 ? 64 : kBlock;
#endif
    auto bytes = [&](int t) { return sizeof(FT) * (size_t)(2 * nq + (t / group) * ColLds<FT>::per_group(nq)); };
    while (threads > 64 && bytes(threads) > 150 * 1024) threads /= 2;
    const int ppb = threads / group;
    const int64_t tiles = (n + ppb - 1) / ppb, rounded = (tiles + 7) / 8 * 8;                        // a multiple of 8: xcd_tile
    if (rounded > (int64_t)0x7fffffff) return false;
    grid = dim3((unsigned)rounded); block = dim3(threads); lds = bytes(threads);
    return true;
}

// one launch site for both entries: picks the group width from the quadrature order, sizes the workgroup so the LDS caches fit,
// raises the dynamic-LDS limit when needed
template <typename FT, typename QUAD, bool FUSED, typename EXTRA = NoExtra>
static int32_t launch_collision_kernel(const P3Consts<FT> &c, const P3VelConsts<FT> &v, const P3ColConsts<FT> &k, const QUAD &quad,
                                       const P3ColIO<FT> &io, int64_t n, bool aspect, hipStream_t st, const EXTRA &ex = EXTRA{}) {
    const int group = collision_group(quad.n);
    dim3 grid, block;
    size_t lds;
    if (!collision_geometry<FT>(group, quad.n, n, grid, block, lds)) return CMX_ERR_UNSUPPORTED;
    auto go = [&](auto kern) -> int32_t {
        if (lds > 48 * 1024) CMX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, grid, block, lds, st, c, v, k, quad, io, n, ex);
        CMX_HIP_TRY(hipGetLastError());
        return CMX_OK;
    };
    if (group == 8)
        return aspect ? go(&p3_collision_kernel<FT, QUAD, true, FUSED, 8, EXTRA>) : go(&p3_collision_kernel<FT, QUAD, false, FUSED, 8, EXTRA>);
    // 16 lanes per state: rules of order >= 48 (collision_group) — never a QuadSmall rule (order <= 32), so the one-launch form has no such
    // instantiation
    if constexpr (std::is_same_v<QUAD, QuadSmall<FT>>) return CMX_ERR_BAD_ARG;
    else return aspect ? go(&p3_collision_kernel<FT, QUAD, true, FUSED, 16, EXTRA>) : go(&p3_collision_kernel<FT, QUAD, false, FUSED, 16, EXTRA>);
}

template <typename FT, typename IP, typename AP, typename TH, typename QUAD>
static int32_t p3_collision_entry(const IP *ip, const AP *aps, const TH *tps, const QUAD *quad, uint32_t flags, int64_t n, const FT *rho_q,
                                  const FT *rho_n, const FT *x3, const FT *x4, const FT *L_c, const FT *N_c, const FT *L_r, const FT *N_r,
                                  const FT *rho_a, const FT *T, const FT *loglam, FT *const *sources, FT *const *rates, void *stream) {
    if (!ip || !aps || !tps || !quad || n < 0 ||
        (flags & ~(CMX_P3_INPUT_IS_STATE | CMX_P3_SLOPE_CONSTANT | CMX_P3_NO_ASPECT_RATIO | CMX_P3_RAIN_PDF_LIMITED)))
        return CMX_ERR_BAD_ARG;
    if (n > kMaxPoints) return CMX_ERR_UNSUPPORTED;      // one launch cannot express the grid (cmx_launch.hpp)
    if (quad->n < 1 || quad->n > CMX_QUAD_MAX) return CMX_ERR_BAD_ARG;
    if (n == 0) return CMX_OK;
    if (!rho_q || !rho_n || !x3 || !x4 || !L_c || !N_c || !L_r || !N_r || !rho_a || !T || !loglam || (!sources && !rates)) return CMX_ERR_BAD_ARG;
    P3Consts<FT> c = make_p3_consts<FT>(ip->scheme, flags);
    c.brent_iters = 0;
    P3VelConsts<FT> v = make_p3_vel_consts<FT>(ip->scheme, ip->vel_ice, 1e-5);
    v.p_lo = FT(0.00001); v.p_hi = FT(1) - v.p_lo;
    const P3ColConsts<FT> k = make_p3col_consts<FT>(*ip, *aps, *tps, flags);
    P3ColIO<FT> io{};
    io.rho_q = rho_q; io.rho_n = rho_n; io.x3 = x3; io.x4 = x4; io.L_c = L_c; io.N_c = N_c; io.L_r = L_r; io.N_r = N_r;
    io.rho_a = rho_a; io.T = T; io.loglam = loglam;
    for (int q = 0; q < 7; ++q) io.src[q] = sources ? sources[q] : nullptr;
    for (int q = 0; q < 10; ++q) io.rates[q] = rates ? rates[q] : nullptr;
    return launch_collision_kernel<FT, QUAD, false>(c, v, k, *quad, io, n, !(flags & CMX_P3_NO_ASPECT_RATIO), reinterpret_cast<hipStream_t>(stream));
}

// stand-alone Bigg freezing rates (the KA kernel test_rain_freezing_kernel!, test/gpu_tests.jl:463-468): cloud = generalized-gamma PSD
template <typename FT, bool CLOUD, bool LIMITED>
__global__ __launch_bounds__(kBlock) void liquid_freezing_kernel(const PointwiseConsts<FT> k, const FT *__restrict__ q, const FT *__restrict__ rho,
                                                                const FT *__restrict__ N, const FT *__restrict__ T, FT *__restrict__ dn,
                                                                FT *__restrict__ dq, const int64_t n) {
    Math<FT>::prepare();   // Float64: exp2 / log2 tables → LDS (no-op for Float32)
    using P = PM<FT>;
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const FT r = rho[i], Ni = N[i], Ti = T[i], qi = q[i];
    const FT J = k.rf_B * P::exp(k.rf_a * (k.T_freeze_tps - Ti));
    FT a, b;
    if constexpr (CLOUD) bigg_cloud<FT>(k, J, r, qi, Ni / r, Ni, Ti, a, b);
    else bigg_rain<FT, LIMITED>(k, J, r, qi, Ni / r, Ni, Ti, a, b);
    if (dn) dn[i] = a;
    if (dq) dq[i] = b;
}
template <typename FT, typename IP, typename TH>
static int32_t liquid_freezing_entry(const IP *ip, const TH *tps, uint32_t flags, int64_t n, const FT *q, const FT *rho, const FT *N, const FT *T,
                                     FT *dn, FT *dq, void *stream) {
    if (!ip || !tps || n < 0 || (flags & ~(uint32_t)(CMX_FREEZE_CLOUD_PSD | CMX_P3_RAIN_PDF_LIMITED))) return CMX_ERR_BAD_ARG;
    if (n > kMaxPoints) return CMX_ERR_UNSUPPORTED;      // one launch cannot express the grid (cmx_launch.hpp)
    if (n == 0) return CMX_OK;
    if (!q || !rho || !N || !T || (!dn && !dq)) return CMX_ERR_BAD_ARG;
    using WR = std::conditional_t<std::is_same_v<FT, float>, cmx_warm_rain_2m_f32, cmx_warm_rain_2m_f64>;
    WR wr{};
    const PointwiseConsts<FT> k = make_pointwise_consts<FT>(wr, *ip, *tps);
    const dim3 grid((unsigned)((n + kBlock - 1) / kBlock)), block(kBlock);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (flags & CMX_FREEZE_CLOUD_PSD) hipLaunchKernelGGL((liquid_freezing_kernel<FT, true, true>), grid, block, 0, st, k, q, rho, N, T, dn, dq, n);
    else if (flags & CMX_P3_RAIN_PDF_LIMITED) hipLaunchKernelGGL((liquid_freezing_kernel<FT, false, true>), grid, block, 0, st, k, q, rho, N, T, dn, dq, n);
    else hipLaunchKernelGGL((liquid_freezing_kernel<FT, false, false>), grid, block, 0, st, k, q, rho, N, T, dn, dq, n);
    CMX_HIP_TRY(hipGetLastError());
    return CMX_OK;
}

template <typename FT, typename WR, typename IP, typename TH>
static int32_t mp2m_p3_entry(const WR *wr, const IP *ip, const TH *tps, uint32_t flags, int64_t n, const FT *rho, const FT *T, const FT *q_tot,
                             const FT *q_lcl, const FT *n_lcl, const FT *q_rai, const FT *n_rai, const FT *q_ice, const FT *n_ice, const FT *q_rim,
                             const FT *b_rim, const FT *loglam, const FT *shift, FT *const *out, void *stream, const SegLayout *lay = nullptr) {
    if (!wr || !ip || !tps || n < 0 || (flags & ~(CMX_P3_SLOPE_CONSTANT | CMX_P3_NO_ASPECT_RATIO | CMX_P3_RAIN_PDF_LIMITED))) return CMX_ERR_BAD_ARG;
    if (n > kMaxPoints) return CMX_ERR_UNSUPPORTED;      // one launch cannot express the grid (cmx_launch.hpp)
    if (ip->quad.n < 1 || ip->quad.n > CMX_QUAD_MAX) return CMX_ERR_BAD_ARG;
    if (n == 0) return CMX_OK;
    if (!rho || !T || !q_tot || !q_lcl || !n_lcl || !q_rai || !n_rai || !q_ice || !n_ice || !q_rim || !b_rim || !loglam || !out) return CMX_ERR_BAD_ARG;
    for (int q = 0; q < 8; ++q) if (!out[q]) return CMX_ERR_BAD_ARG;
    const bool limited = (flags & CMX_P3_RAIN_PDF_LIMITED) != 0;
    using RV = std::conditional_t<std::is_same_v<FT, float>, cmx_rain_vel_f32, cmx_rain_vel_f64>;
    if ((flags & CMX_P3_RAIN_PDF_LIMITED) && !sb_limiters_ok(*wr)) return CMX_ERR_BAD_ARG;      // clamp_ordered needs ordered limiter pairs
    const SbConsts<FT> sc = make_sb_consts<FT>(*wr, *tps, (const RV *)nullptr, (double)Math<FT>::eps_1m());
    P3Consts<FT> c = make_p3_consts<FT>(ip->scheme, flags & CMX_P3_SLOPE_CONSTANT);
    c.brent_iters = 0;
    const PointwiseConsts<FT> pk = make_pointwise_consts<FT>(*wr, *ip, *tps);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const bool intpow = sb_integer_exponents(*wr) && CMX_SB_INTPOW_P3;   // the same warm-rain instantiation as cmx_sb2006_warm_rain_tendencies_* picks
    // ice processes
    P3VelConsts<FT> v = make_p3_vel_consts<FT>(ip->scheme, ip->vel_ice, 1e-5);
    v.p_lo = FT(0.00001); v.p_hi = FT(1) - v.p_lo;
    const P3ColConsts<FT> k = make_p3col_consts<FT>(*ip, wr->air_properties, *tps, flags);
    P3ColIO<FT> io{};
    io.rho_a = rho; io.T = T; io.loglam = loglam;
    io.q_lcl = q_lcl; io.n_lcl = n_lcl; io.q_rai = q_rai; io.n_rai = n_rai; io.q_ice = q_ice; io.n_ice = n_ice; io.q_rim = q_rim; io.b_rim = b_rim;
    for (int q = 0; q < 8; ++q) io.out[q] = out[q];
    if (lay) io.lay = *lay;
    using QUAD = std::remove_cv_t<std::remove_reference_t<decltype(ip->quad)>>;
#ifndef CMX_MP2M_P3_ONE_LAUNCH
#define CMX_MP2M_P3_ONE_LAUNCH 1      // 0: always the two launches (A/B switch)
#endif
    if (CMX_MP2M_P3_ONE_LAUNCH && ip->quad.n <= 32) {
        // ONE launch: the collision kernel's epilogue evaluates the pointwise part too (PointwiseExtra) — each tendency column is written once
        QuadSmall<FT> qs{};
        qs.n = ip->quad.n;
        for (int j = 0; j < qs.n; ++j) { qs.node[j] = ip->quad.node[j]; qs.weight[j] = ip->quad.weight[j]; }
        const bool aspect = !(flags & CMX_P3_NO_ASPECT_RATIO);
        auto go = [&](auto ex) -> int32_t {
            ex.sc = sc; ex.pk = pk; ex.q_tot = q_tot; ex.shift = shift;
            return launch_collision_kernel<FT, QuadSmall<FT>, true, decltype(ex)>(c, v, k, qs, io, n, aspect, st, ex);
        };
        if (limited) return intpow ? go(PointwiseExtra<FT, true, true>{}) : go(PointwiseExtra<FT, true, false>{});
        return intpow ? go(PointwiseExtra<FT, false, true>{}) : go(PointwiseExtra<FT, false, false>{});
    }
    // TWO launches (quadrature rules above order 32 — the kernel-argument segment is 4 KiB): pointwise kernel, then the collision kernel
    // read-modify-writes the eight columns
    FusedIO<FT> fio{rho, T, q_tot, q_lcl, n_lcl, q_rai, n_rai, q_ice, n_ice, q_rim, b_rim, shift, {}};
    for (int q = 0; q < 8; ++q) fio.out[q] = out[q];
    if (lay) fio.lay = *lay;
    const dim3 block(kBlock), grid1((unsigned)((n + kBlock - 1) / kBlock));
    if (intpow) {
        if (limited) hipLaunchKernelGGL((mp2m_p3_pointwise_kernel<FT, true, true>), grid1, block, 0, st, sc, c, pk, fio, n);
        else hipLaunchKernelGGL((mp2m_p3_pointwise_kernel<FT, false, true>), grid1, block, 0, st, sc, c, pk, fio, n);
    } else {
        if (limited) hipLaunchKernelGGL((mp2m_p3_pointwise_kernel<FT, true>), grid1, block, 0, st, sc, c, pk, fio, n);
        else hipLaunchKernelGGL((mp2m_p3_pointwise_kernel<FT, false>), grid1, block, 0, st, sc, c, pk, fio, n);
    }
    CMX_HIP_TRY(hipGetLastError());
    return launch_collision_kernel<FT, QUAD, true>(c, v, k, ip->quad, io, n, !(flags & CMX_P3_NO_ASPECT_RATIO), st);
}

// the same entry on segmented columns (the host model's fields in place): in[13] / in_seg_stride[13] in the order of SegLayout::s_in
template <typename FT, typename WR, typename IP, typename TH>
static int32_t mp2m_p3_fields_entry(const WR *wr, const IP *ip, const TH *tps, uint32_t flags, int64_t n_seg, int64_t seg_len, const FT *const *in,
                                    const int64_t *in_seg_stride, FT *const *out, const int64_t *out_seg_stride, void *stream) {
    if (!in || !in_seg_stride || !out || !out_seg_stride || n_seg < 0 || seg_len < 0) return CMX_ERR_BAD_ARG;
    if (n_seg == 0 || seg_len == 0) return (wr && ip && tps) ? CMX_OK : CMX_ERR_BAD_ARG;
    if (n_seg > kMaxPoints / seg_len) return CMX_ERR_UNSUPPORTED;
    SegLayout lay{};
    lay.seg_len = seg_len;
    for (int k = 0; k < 13; ++k) {
        if (k < 12 && !in[k]) return CMX_ERR_BAD_ARG;                      // the INPC shift (k = 12) is optional
        if (in[k] && n_seg > 1 && in_seg_stride[k] < seg_len) return CMX_ERR_BAD_ARG;   // runs must not overlap
        lay.s_in[k] = in_seg_stride[k];
    }
    for (int q = 0; q < 8; ++q) {
        if (!out[q] || (n_seg > 1 && out_seg_stride[q] < seg_len)) return CMX_ERR_BAD_ARG;
        lay.s_out[q] = out_seg_stride[q];
    }
    return mp2m_p3_entry<FT>(wr, ip, tps, flags, n_seg * seg_len, in[0], in[1], in[2], in[3], in[4], in[5], in[6], in[7], in[8], in[9], in[10], in[11],
                             in[12], out, stream, &lay);
}

}  // namespace cmx

extern "C" {

int32_t cmx_microphysics_2m_p3_tendencies_fields_f32(const cmx_warm_rain_2m_f32 *warm_rain, const cmx_p3_ice_params_f32 *ice, const cmx_thermo_f32 *tps,
                                                     uint32_t flags, int64_t n_seg, int64_t seg_len, const float *const *in,
                                                     const int64_t *in_seg_stride, float *const *out, const int64_t *out_seg_stride, void *stream) {
    return cmx::mp2m_p3_fields_entry<float>(warm_rain, ice, tps, flags, n_seg, seg_len, in, in_seg_stride, out, out_seg_stride, stream);
}
int32_t cmx_microphysics_2m_p3_tendencies_fields_f64(const cmx_warm_rain_2m_f64 *warm_rain, const cmx_p3_ice_params_f64 *ice, const cmx_thermo_f64 *tps,
                                                     uint32_t flags, int64_t n_seg, int64_t seg_len, const double *const *in,
                                                     const int64_t *in_seg_stride, double *const *out, const int64_t *out_seg_stride, void *stream) {
    return cmx::mp2m_p3_fields_entry<double>(warm_rain, ice, tps, flags, n_seg, seg_len, in, in_seg_stride, out, out_seg_stride, stream);
}

int32_t cmx_p3_liquid_ice_collisions_f32(const cmx_p3_ice_params_f32 *ice, const cmx_air_properties_f32 *aps, const cmx_thermo_f32 *tps,
                                         const cmx_quadrature_f32 *quad, uint32_t flags, int64_t n, const float *rho_q_ice,
                                         const float *rho_n_ice, const float *x3, const float *x4, const float *L_c, const float *N_c,
                                         const float *L_r, const float *N_r, const float *rho_air, const float *T, const float *log_lambda,
                                         float *const *sources, float *const *rates, void *stream) {
    return cmx::p3_collision_entry<float>(ice, aps, tps, quad, flags, n, rho_q_ice, rho_n_ice, x3, x4, L_c, N_c, L_r, N_r, rho_air, T, log_lambda,
                                          sources, rates, stream);
}
int32_t cmx_p3_liquid_ice_collisions_f64(const cmx_p3_ice_params_f64 *ice, const cmx_air_properties_f64 *aps, const cmx_thermo_f64 *tps,
                                         const cmx_quadrature_f64 *quad, uint32_t flags, int64_t n, const double *rho_q_ice,
                                         const double *rho_n_ice, const double *x3, const double *x4, const double *L_c, const double *N_c,
                                         const double *L_r, const double *N_r, const double *rho_air, const double *T, const double *log_lambda,
                                         double *const *sources, double *const *rates, void *stream) {
    return cmx::p3_collision_entry<double>(ice, aps, tps, quad, flags, n, rho_q_ice, rho_n_ice, x3, x4, L_c, N_c, L_r, N_r, rho_air, T, log_lambda,
                                           sources, rates, stream);
}

int32_t cmx_microphysics_2m_p3_tendencies_f32(const cmx_warm_rain_2m_f32 *warm_rain, const cmx_p3_ice_params_f32 *ice, const cmx_thermo_f32 *tps,
                                              uint32_t flags, int64_t n, const float *rho, const float *T, const float *q_tot, const float *q_lcl,
                                              const float *n_lcl, const float *q_rai, const float *n_rai, const float *q_ice, const float *n_ice,
                                              const float *q_rim, const float *b_rim, const float *log_lambda, const float *inpc_log_shift,
                                              float *const *tendencies, void *stream) {
    return cmx::mp2m_p3_entry<float>(warm_rain, ice, tps, flags, n, rho, T, q_tot, q_lcl, n_lcl, q_rai, n_rai, q_ice, n_ice, q_rim, b_rim, log_lambda,
                                     inpc_log_shift, tendencies, stream);
}
int32_t cmx_microphysics_2m_p3_tendencies_f64(const cmx_warm_rain_2m_f64 *warm_rain, const cmx_p3_ice_params_f64 *ice, const cmx_thermo_f64 *tps,
                                              uint32_t flags, int64_t n, const double *rho, const double *T, const double *q_tot, const double *q_lcl,
                                              const double *n_lcl, const double *q_rai, const double *n_rai, const double *q_ice, const double *n_ice,
                                              const double *q_rim, const double *b_rim, const double *log_lambda, const double *inpc_log_shift,
                                              double *const *tendencies, void *stream) {
    return cmx::mp2m_p3_entry<double>(warm_rain, ice, tps, flags, n, rho, T, q_tot, q_lcl, n_lcl, q_rai, n_rai, q_ice, n_ice, q_rim, b_rim, log_lambda,
                                      inpc_log_shift, tendencies, stream);
}

int32_t cmx_liquid_freezing_rate_f32(const cmx_p3_ice_params_f32 *ice, const cmx_thermo_f32 *tps, uint32_t flags, int64_t n, const float *q,
                                     const float *rho, const float *N, const float *T, float *dn_frz, float *dq_frz, void *stream) {
    return cmx::liquid_freezing_entry<float>(ice, tps, flags, n, q, rho, N, T, dn_frz, dq_frz, stream);
}
int32_t cmx_liquid_freezing_rate_f64(const cmx_p3_ice_params_f64 *ice, const cmx_thermo_f64 *tps, uint32_t flags, int64_t n, const double *q,
                                     const double *rho, const double *N, const double *T, double *dn_frz, double *dq_frz, void *stream) {
    return cmx::liquid_freezing_entry<double>(ice, tps, flags, n, q, rho, N, T, dn_frz, dq_frz, stream);
}

}  // extern "C"
