/* cmx_oracle_diag_impl.h — TEST INFRASTRUCTURE (see cmx_oracle.c): the reference's cloud diagnostics, operation by operation.
 *   src/CloudDiagnostics.jl   radar_reflectivity_1M :31-46, radar_reflectivity_2M :64-84, effective_radius_2M :100-125,
 *                             effective_radius_Liu_Hallet_97 :143-163
 * with  CM2.pdf_rain_parameters_mass src/Microphysics2M.jl:141-146 (o_pdf_rain_Br, cmx_oracle_impl.h), log_pdf_cloud_parameters_mass :176-192 /
 * pdf_cloud_parameters_mass :200-203 (restated here: the mass-based B, not the diameter-based λ of o_pdf_cloud_parameters),
 * DT.generalized_gamma_Mⁿ src/DistributionTools.jl:109-112, CM1.get_n0 / lambda_inverse src/Microphysics1M.jl:83-152 (o_lambda_inverse,
 * cmx_oracle_1m_impl.h).  Included once per float type by cmx_oracle_impl.h (FT, FN, TY as there). */

/* CM2.pdf_cloud_parameters_mass(pdf_c, q, ρₐ, N).Bc = exp(logB); (N < ϵ_N || q < ϵ_M) → logB = +Inf → Bc = Inf */
static inline FT FN(o_pdf_cloud_Bc)(const TY(cmx_cloud_pdf_sb2006) * pdf, FT q, FT rho, FT N, const TY(cmxo_thresholds) * th) {
    FT safe_q = FN(o_max)(q, th->eps_m), safe_N = FN(o_max)(N, th->eps_n);
    FT L = rho * safe_q;
    FT logx = M_LOG(L / safe_N);
    FT logB = -pdf->mu_c * (logx + pdf->loggamma_z1 - pdf->loggamma_z2);
    if (N < th->eps_n || q < th->eps_m) logB = (FT)INFINITY;
    return M_EXP(logB);
}
/* DT.generalized_gamma_Mⁿ(ν, μ, B, N, n) = N B^(−n/μ) Γ((ν+1+n)/μ)/Γ((ν+1)/μ) */
static inline FT FN(o_gg_Mn)(FT nu, FT mu, FT B, FT N, FT n) {
    return N * M_POW(B, -n / mu) * M_TGAMMA((nu + 1 + n) / mu) / M_TGAMMA((nu + 1) / mu);
}
static inline FT FN(o_log10)(FT x) { return sizeof(FT) == 4 ? (FT)log10f((float)x) : (FT)log10((double)x); }
static inline int FN(o_notvalid)(FT B) { return B == 0 || !isfinite((double)B); }     /* notvalid(B) = iszero(B) || !isfinite(B)  :72 */

/* CMD.radar_reflectivity_1M((; pdf, mass)::Rain, q, ρ)  :31-46 */
static inline FT FN(o_radar_reflectivity_1m)(const TY(cmx_rain) * rain, FT q, FT rho, const TY(cmxo_thresholds) * th) {
    FT n0 = rain->n0 * (FT)1e-12;                                                   /* change units for accuracy */
    FT lam_inv = FN(o_lambda_inverse)(rain->n0, &rain->mass, q, rho, th->eps_1m) / (FT)1e-3;
    FT Z = 720 * n0 * M_POW(lam_inv, (FT)7);
    FT log_10_Z0 = (FT)-18;
    FT log_Z = (FT)10 * (FN(o_log10)(Z) - log_10_Z0 - (FT)9);
    return FN(o_max)((FT)-150, log_Z);
}
/* CMD.radar_reflectivity_2M((; pdf_c, pdf_r)::SB2006, q_lcl, q_rai, N_lcl, N_rai, ρ)  :64-84 */
static inline FT FN(o_radar_reflectivity_2m)(const TY(cmx_cloud_pdf_sb2006) * pc, const TY(cmx_rain_pdf_sb2006) * pr, int limited, FT q_lcl, FT q_rai,
                                            FT N_lcl, FT N_rai, FT rho, const TY(cmxo_thresholds) * th) {
    FT C = (FT)(4.0 / 3.0 * M_PI * (double)pr->rho_w);                              /* FT(4 / 3 * π * ρw): formed in Float64, rounded once */
    FT Br = FN(o_pdf_rain_Br)(pr, limited, q_rai, rho, N_rai, th);
    FT Bc = FN(o_pdf_cloud_Bc)(pc, q_lcl, rho, N_lcl, th);
    FT Zc = FN(o_notvalid)(Bc) ? (FT)0 : FN(o_gg_Mn)(pc->nu_c, pc->mu_c, Bc, N_lcl, (FT)2) / (C * C);
    FT Zr = FN(o_notvalid)(Br) ? (FT)0 : FN(o_gg_Mn)(pr->nu_r, pr->mu_r, Br, N_rai, (FT)2) / (C * C);
    FT lz = FN(o_log10)(FN(o_max)((FT)0, Zc + Zr));
    return FN(o_max)((FT)-150, 10 * (lz + 18));
}
/* CMD.effective_radius_2M(SB2006, q_lcl, q_rai, N_lcl, N_rai, ρ)  :100-125 */
static inline FT FN(o_effective_radius_2m)(const TY(cmx_cloud_pdf_sb2006) * pc, const TY(cmx_rain_pdf_sb2006) * pr, int limited, FT q_lcl, FT q_rai,
                                          FT N_lcl, FT N_rai, FT rho, const TY(cmxo_thresholds) * th) {
    FT C = (FT)(4.0 / 3.0 * M_PI * (double)pr->rho_w);
    FT Br = FN(o_pdf_rain_Br)(pr, limited, q_rai, rho, N_rai, th);
    FT Bc = FN(o_pdf_cloud_Bc)(pc, q_lcl, rho, N_lcl, th);
    int nvc = FN(o_notvalid)(Bc), nvr = FN(o_notvalid)(Br);
    FT M3_c = nvc ? (FT)0 : FN(o_gg_Mn)(pc->nu_c, pc->mu_c, Bc, N_lcl, (FT)1) / C;
    FT M3_r = nvr ? (FT)0 : FN(o_gg_Mn)(pr->nu_r, pr->mu_r, Br, N_rai, (FT)1) / C;
    FT n23 = (FT)2 / 3;
    FT M2_c = nvc ? (FT)0 : FN(o_gg_Mn)(pc->nu_c, pc->mu_c, Bc, N_lcl, n23) / M_POW(C, n23);
    FT M2_r = nvr ? (FT)0 : FN(o_gg_Mn)(pr->nu_r, pr->mu_r, Br, N_rai, n23) / M_POW(C, n23);
    return M2_c + M2_r <= th->eps_1m ? (FT)0 : (M3_c + M3_r) / (M2_c + M2_r);
}
/* CMD.effective_radius_Liu_Hallet_97((; ρw), ρ_air, q_lcl, N_lcl, q_rai, N_rai)  :143-163 */
static inline FT FN(o_effective_radius_lh97)(FT rho_w, FT rho, FT q_lcl, FT N_lcl, FT q_rai, FT N_rai, const TY(cmxo_thresholds) * th) {
    FT k = (FT)0.8;
    FT r_vol = (N_lcl + N_rai) < th->eps_1m ? (FT)0
                                            : M_POW(((FT)3 * (q_lcl + q_rai) * rho) / ((FT)4 * (FT)M_PI * rho_w * (N_lcl + N_rai)), (FT)(1.0 / 3.0));
    return r_vol / M_POW(k, (FT)(1.0 / 3.0));
}

/* over columns; any output may be NULL; rain / (pc, pr) may be NULL with the outputs that need them */
void FN(cmxo_cloud_diagnostics)(const TY(cmx_rain) * rain, const TY(cmx_cloud_pdf_sb2006) * pc, const TY(cmx_rain_pdf_sb2006) * pr, FT rho_w, int limited,
                                const TY(cmxo_thresholds) * th, int64_t n, const FT *rho, const FT *q_lcl, const FT *q_rai, const FT *N_lcl,
                                const FT *N_rai, FT *Z_1m, FT *Z_2m, FT *reff_2m, FT *reff_lh97) {
    for (int64_t i = 0; i < n; ++i) {
        if (Z_1m) Z_1m[i] = FN(o_radar_reflectivity_1m)(rain, q_rai[i], rho[i], th);
        if (Z_2m) Z_2m[i] = FN(o_radar_reflectivity_2m)(pc, pr, limited, q_lcl[i], q_rai[i], N_lcl[i], N_rai[i], rho[i], th);
        if (reff_2m) reff_2m[i] = FN(o_effective_radius_2m)(pc, pr, limited, q_lcl[i], q_rai[i], N_lcl[i], N_rai[i], rho[i], th);
        if (reff_lh97)
            reff_lh97[i] = FN(o_effective_radius_lh97)(rho_w, rho[i], q_lcl[i], N_lcl ? N_lcl[i] : (FT)100, q_rai ? q_rai[i] : (FT)0, N_rai ? N_rai[i] : (FT)0, th);
    }
}
