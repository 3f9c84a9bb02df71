/* cmx_oracle_dist_impl.h — TEST INFRASTRUCTURE (see cmx_oracle.c): the reference's size-distribution helpers.
 *   src/DistributionTools.jl   generalized_gamma_quantile :44-47, generalized_gamma_cdf :75-82, exponential_cdf :124-129,
 *                              exponential_quantile :146-151
 *   src/Microphysics2M.jl      size_distribution (rain) :270-277, (cloud) :295-303, size_distribution_value :312-315,
 *                              get_size_distribution_bounds (rain) :336-345, (cloud) :346-354
 * Included once per float type by cmx_oracle_impl.h (FT, FN, TY as there); uses o_pdf_rain_parameters (cmx_oracle_impl.h),
 * o_pdf_cloud_parameters (cmx_oracle_p3col_impl.h), o_gamma_inc / o_gamma_inc_inv (cmx_oracle_p3_impl.h). */

/* DT.generalized_gamma_quantile(ν, μ, B, Y) = (gamma_inc_inv((ν+1)/μ, Y, 1−Y)/B)^(1/μ) */
static inline FT FN(o_gg_quantile)(FT nu, FT mu, FT B, FT Y, FT eps) {
    FT z = FN(o_gamma_inc_inv)((nu + 1) / mu, Y, 1 - Y, sizeof(FT) == 4 ? 20 : 30, eps);
    return M_POW(z / B, 1 / mu);
}
/* DT.generalized_gamma_cdf(ν, μ, B, x): 0 for x ≤ 0, else P((ν+1)/μ, B x^μ)  (the DomainErrors of μ ≤ 0, B ≤ 0 → NaN) */
static inline FT FN(o_gg_cdf)(FT nu, FT mu, FT B, FT x) {
    if (!(mu > 0) || !(B > 0)) return (FT)NAN;
    if (x <= 0) return 0;
    FT P, Q;
    FN(o_gamma_inc)((nu + 1) / mu, B * M_POW(x, mu), sizeof(FT) == 4 ? 20 : 30, &P, &Q);
    return P;
}
/* LEF.log1mexp(x) = log(1 − eˣ): o_log1mexp of cmx_oracle_1m_impl.h */
/* LEF.cloglog(Y) = log(−log(1 − Y)) = log(−log1p(−Y)) */
static inline FT FN(o_cloglog)(FT y) { return M_LOG(-M_LOG1P(-y)); }
/* DT.exponential_cdf(D_mean, D): 0 for D < 0, exp(log1mexp(−D/D_mean)); DomainError (D_mean ≤ 0) → NaN */
static inline FT FN(o_exp_cdf)(FT D_mean, FT D) {
    if (!(D_mean > 0)) return (FT)NAN;
    if (D < 0) return 0;
    return M_EXP(FN(o_log1mexp)(-D / D_mean));
}
/* DT.exponential_quantile(D_mean, Y) = exp(log D_mean + cloglog(Y)); DomainErrors → NaN */
static inline FT FN(o_exp_quantile)(FT D_mean, FT Y) {
    if (!(Y >= 0 && Y <= 1) || !(D_mean > 0)) return (FT)NAN;
    return M_EXP(M_LOG(D_mean) + FN(o_cloglog)(Y));
}

void FN(cmxo_generalized_gamma)(FT nu, FT mu, int64_t n, const FT *B, const FT *Y, const FT *x, FT *quantile, FT *cdf) {
    for (int64_t i = 0; i < n; ++i) {
        if (quantile) quantile[i] = FN(o_gg_quantile)(nu, mu, B[i], Y[i], M_EPS);
        if (cdf) cdf[i] = FN(o_gg_cdf)(nu, mu, B[i], x[i]);
    }
}
void FN(cmxo_exponential_distribution)(int64_t n, const FT *D_mean, const FT *Y, const FT *D, FT *quantile, FT *cdf) {
    for (int64_t i = 0; i < n; ++i) {
        if (quantile) quantile[i] = FN(o_exp_quantile)(D_mean[i], Y[i]);
        if (cdf) cdf[i] = FN(o_exp_cdf)(D_mean[i], D[i]);
    }
}
/* CM2.size_distribution_value(pdf, q, ρₐ, N, D) and CM2.get_size_distribution_bounds(pdf, q, ρₐ, N, p) over columns; cloud != 0 selects
 * the cloud (generalized-gamma) PSD.  D / n_D and D_min / D_max are optional. */
void FN(cmxo_sb2006_size_distribution)(const TY(cmx_cloud_pdf_sb2006) * pdf_c, const TY(cmx_rain_pdf_sb2006) * pdf_r, int cloud, int limited, FT p,
                                       const TY(cmxo_thresholds) * th, int64_t n, const FT *q, const FT *rho, const FT *N, const FT *D, FT *n_D,
                                       FT *D_min, FT *D_max) {
    for (int64_t i = 0; i < n; ++i) {
        if (cloud) {
            FT logN0c, lam_c, nu_cD, mu_cD;
            FN(o_pdf_cloud_parameters)(pdf_c, q[i], rho[i], N[i], th, &logN0c, &lam_c, &nu_cD, &mu_cD);
            if (n_D) n_D[i] = (logN0c == -(FT)INFINITY) ? (FT)0 : M_EXP(logN0c + nu_cD * M_LOG(D[i]) - lam_c * M_POW(D[i], mu_cD));
            if (D_min) D_min[i] = FN(o_gg_quantile)(nu_cD, mu_cD, lam_c, p, th->eps_m);
            if (D_max) D_max[i] = FN(o_gg_quantile)(nu_cD, mu_cD, lam_c, 1 - p, th->eps_m);
        } else {
            TY(cmxo_rain_pdf) r = FN(o_pdf_rain_parameters)(pdf_r, limited, q[i], rho[i], N[i], th);
            if (n_D) n_D[i] = r.N0r == 0 ? (FT)0 : r.N0r * M_EXP(-D[i] / r.Dr_mean);
            if (r.Dr_mean == 0) {
                if (D_min) D_min[i] = 0;
                if (D_max) D_max[i] = 0;
            } else {
                if (D_min) D_min[i] = FN(o_exp_quantile)(r.Dr_mean, p);
                if (D_max) D_max[i] = FN(o_exp_quantile)(r.Dr_mean, 1 - p);
            }
        }
    }
}
