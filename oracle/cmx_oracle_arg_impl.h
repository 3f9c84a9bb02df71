/*
 * cmx_oracle_arg_impl.h — oracle (TEST INFRASTRUCTURE) for Abdul-Razzak & Ghan (2000) aerosol activation.
 * Included from cmx_oracle_impl.h (once per float type).  Restates src/AerosolActivation.jl of the reference
 * operation by operation; Thermodynamics.jl pieces (gas_constant_air, air_density — TDI:17,71-72) from their
 * published formulas: R_m = R_d (1 + (R_v/R_d − 1) q_tot − R_v/R_d (q_liq + q_ice)),  ρ = p / (R_m T).
 * SpecialFunctions erf / erfc (AerosolActivation.jl:257,319) → libm.
 *
 * PARITY PINNING: the reference holds no absolute known-answer value for this path: it is pinned
 * only by the κ-vs-B consistency test (test/gpu_tests.jl:580-587), the digitised Fig. 1 of ARG2000 at rtol 0.05–0.1
 * (test/aerosol_activation_tests.jl:236-299) and an independent mpmath restatement (tests/test_arg2000_oracle.py).
 */

/* TD.gas_constant_air (TDI:17) */
static inline FT FN(o_gas_constant_air)(const TY(cmx_thermo) * p, FT q_tot, FT q_liq, FT q_ice) {
    FT Rv_over_Rd = p->R_v / p->R_d;
    return p->R_d * (1 + (Rv_over_Rd - 1) * q_tot - Rv_over_Rd * (q_liq + q_ice));
}
/* AA.coeff_of_curvature — src/AerosolActivation.jl:35-40 */
static inline FT FN(o_coeff_of_curvature)(const TY(cmx_aerosol_activation_params) * ap, FT T) {
    return (FT)2 * ap->sigma * ap->M_w / ap->rho_w / ap->R / T;
}
/* AA.critical_supersaturation — :107-118 (hygroscopicity = mean_hygroscopicity_parameter, host-evaluated) */
static inline FT FN(o_critical_supersaturation)(const TY(cmx_aerosol_activation_params) * ap,
                                               const TY(cmx_aerosol_mode) * m, FT T) {
    FT A = FN(o_coeff_of_curvature)(ap, T);
    return 2 / M_SQRT(m->hygroscopicity) * M_POW(A / 3 / m->r_dry, (FT)(3.0 / 2.0));
}
/* AA.max_supersaturation — :138-200 */
static inline FT FN(o_max_supersaturation)(const TY(cmx_aerosol_activation_params) * ap,
                                          const TY(cmx_aerosol_distribution) * ad,
                                          const TY(cmx_air_properties) * aip, const TY(cmx_thermo) * tps,
                                          const TY(cmxo_thresholds) * th, FT T, FT p, FT w, FT q_tot, FT q_liq,
                                          FT q_ice, FT N_liq, FT N_ice, FT *cond) {
    const FT pi = (FT)M_PI;
    FT R_v = tps->R_v;
    FT R_m = FN(o_gas_constant_air)(tps, q_tot, q_liq, q_ice);
    FT cp_m = FN(o_cp_m)(tps, q_tot, q_liq, q_ice);
    FT L_v = FN(o_latent_heat_vapor)(tps, T);
    FT rho_air = p / (R_m * T);                                   /* TD.air_density */
    FT p_v = (q_tot - q_liq - q_ice) * rho_air * R_v * T;
    FT p_vs = FN(o_psat_liquid)(tps, T);
    FT G = FN(o_G_func_liquid)(aip, tps, T, th) / ap->rho_w;
    FT alpha = p_v / p_vs * (L_v * ap->g / R_v / cp_m / (T * T) - ap->g / R_m / T);
    FT gamma = R_v * T / p_vs + p_v / p_vs * R_m * (L_v * L_v) / R_v / cp_m / T / p;
    FT A = FN(o_coeff_of_curvature)(ap, T);
    FT zeta = 2 * A / 3 * M_SQRT(alpha * w / G);
    FT tmp = 0;
    for (int i = 0; i < ad->n_modes; ++i) {
        const TY(cmx_aerosol_mode) *m = &ad->modes[i];
        FT Sm = FN(o_critical_supersaturation)(ap, m, T);
        FT ls = M_LOG(m->stdev);
        FT f = ap->f1 * M_EXP(ap->f2 * (ls * ls));
        FT g = ap->g1 + ap->g2 * ls;
        FT sq = M_SQRT(alpha * w / G);
        FT eta = (sq * sq * sq) / ((FT)(2 * M_PI) * ap->rho_w * gamma * m->N);
        tmp += 1 / (Sm * Sm) * (f * M_POW(zeta / eta, ap->p1) + g * M_POW((Sm * Sm) / (eta + 3 * zeta), ap->p2));
    }
    FT S_max_ARG = (FT)1 / M_SQRT(tmp);
    FT r_liq = N_liq < th->eps_ft ? (FT)0 : M_CBRT(rho_air * q_liq / N_liq / ap->rho_w / (FT)(4.0 / 3.0 * M_PI));
    FT K_liq = (FT)(4 * M_PI) * ap->rho_w * N_liq * r_liq * G * gamma;
    FT L_s = FN(o_latent_heat_sublim)(tps, T);
    FT gamma_i = R_v * T / p_vs + p_v / p_vs * R_m * L_v * L_s / R_v / cp_m / T / p;
    FT r_ice = N_ice < th->eps_ft ? (FT)0 : M_CBRT(rho_air * q_ice / N_ice / ap->rho_i / (FT)(4.0 / 3.0 * M_PI));
    FT rho_i_G_i = FN(o_G_func_ice)(aip, tps, T, th);
    FT xi = FN(o_psat_liquid)(tps, T) / FN(o_psat_ice)(tps, T);
    FT K_ice = (FT)(4 * M_PI) * N_ice * r_ice * rho_i_G_i * gamma_i;
    FT S_max = S_max_ARG * (alpha * w - K_ice * (xi - (FT)1)) / (alpha * w + (K_liq + K_ice * xi) * S_max_ARG);
    (void)pi;
    /* conditioning of the numerator αw − K_ice(ξ−1) (≥ 1; = 1 without ice): how much a relative rounding error of
     * its operands is amplified in S_max */
    if (cond) *cond = (M_ABS(alpha * w) + M_ABS(K_ice * (xi - (FT)1))) / M_ABS(alpha * w - K_ice * (xi - (FT)1));
    return FN(o_max)((FT)0, S_max);
}

/* oracle twin of cmx_arg2000_activation_*: N_activated_per_mode (:235-259), M_activated_per_mode (:294-321) */
void FN(cmxo_arg2000_activation)(const TY(cmx_aerosol_activation_params) * ap, const TY(cmx_aerosol_distribution) * ad,
                                const TY(cmx_air_properties) * aip, const TY(cmx_thermo) * tps,
                                const TY(cmxo_thresholds) * th, int64_t n, const FT *T, const FT *p, const FT *w,
                                const FT *q_tot, const FT *q_liq, const FT *q_ice, const FT *N_liq, const FT *N_ice,
                                FT *const *N_act, FT *const *M_act, FT *S_max, FT *S_cond, int32_t nthreads) {
    (void)nthreads;
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
    for (int64_t i = 0; i < n; ++i) {
        FT ql = q_liq ? q_liq[i] : (FT)0, qi = q_ice ? q_ice[i] : (FT)0;
        FT Nl = N_liq ? N_liq[i] : (FT)0, Ni = N_ice ? N_ice[i] : (FT)0;
        FT cond = 1;
        FT smax = FN(o_max_supersaturation)(ap, ad, aip, tps, th, T[i], p[i], w[i], q_tot[i], ql, qi, Nl, Ni, &cond);
        if (S_max) S_max[i] = smax;
        if (S_cond) S_cond[i] = cond;
        for (int k = 0; k < ad->n_modes; ++k) {
            const TY(cmx_aerosol_mode) *m = &ad->modes[k];
            FT sm = FN(o_critical_supersaturation)(ap, m, T[i]);
            if (N_act && N_act[k]) {
                FT u = 2 * M_LOG(sm / smax) / 3 / M_SQRT((FT)2) / M_LOG(m->stdev);
                N_act[k][i] = m->N * (FT)0.5 * (1 - M_ERF(u));
            }
            if (M_act && M_act[k]) {
                FT fac = 3 * M_LOG(m->stdev) * M_SQRT((FT)2) / 2;
                FT u = M_LOG(sm / smax) / fac;
                M_act[k][i] = m->molar_mass_mix / 2 * M_ERFC(u - fac);
            }
        }
    }
}

/* The argument u_i of the error function in N_activated_per_mode (:254) per mode and state — for the parity tests: the reference forms the activated
 * number as N ½ (1 − erf u), which cancels to an ABSOLUTE accuracy of eps(FT)·N/2 in the tail (u ≳ 4 in Float32, ≳ 6 in Float64); the tests compare the
 * device also with N ½ erfc(u) of THIS u, the same quantity without the cancellation. */
void FN(cmxo_arg2000_erf_argument)(const TY(cmx_aerosol_activation_params) * ap, const TY(cmx_aerosol_distribution) * ad,
                                  const TY(cmx_air_properties) * aip, const TY(cmx_thermo) * tps, const TY(cmxo_thresholds) * th, int64_t n,
                                  const FT *T, const FT *p, const FT *w, const FT *q_tot, const FT *q_liq, const FT *q_ice, const FT *N_liq,
                                  const FT *N_ice, FT *const *u_out) {
    for (int64_t i = 0; i < n; ++i) {
        FT cond = 1;
        FT smax = FN(o_max_supersaturation)(ap, ad, aip, tps, th, T[i], p[i], w[i], q_tot[i], q_liq ? q_liq[i] : (FT)0, q_ice ? q_ice[i] : (FT)0,
                                            N_liq ? N_liq[i] : (FT)0, N_ice ? N_ice[i] : (FT)0, &cond);
        for (int k = 0; k < ad->n_modes; ++k) {
            FT sm = FN(o_critical_supersaturation)(ap, &ad->modes[k], T[i]);
            u_out[k][i] = 2 * M_LOG(sm / smax) / 3 / M_SQRT((FT)2) / M_LOG(ad->modes[k].stdev);
        }
    }
}

/* oracle twin of cmx_arg2000_activation_columns_*: the aerosol modes vary in space — (r_dry, stdev, N, hygroscopicity,
 * molar_mass_mix) are columns per mode, as the reference's own GPU test passes them (test/gpu_tests.jl:45-79) */
void FN(cmxo_arg2000_activation_columns)(const TY(cmx_aerosol_activation_params) * ap, const TY(cmx_air_properties) * aip,
                                        const TY(cmx_thermo) * tps, const TY(cmxo_thresholds) * th, int32_t n_modes, int64_t n,
                                        const FT *T, const FT *p, const FT *w, const FT *q_tot, const FT *q_liq, const FT *q_ice,
                                        const FT *N_liq, const FT *N_ice, const FT *const *r_dry, const FT *const *stdev,
                                        const FT *const *N_mode, const FT *const *hyg, const FT *const *mmix, FT *const *N_act,
                                        FT *const *M_act, FT *S_max, int32_t nthreads) {
    (void)nthreads;
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
    for (int64_t i = 0; i < n; ++i) {
        TY(cmx_aerosol_distribution) ad;
        ad.n_modes = n_modes;
        for (int k = 0; k < n_modes; ++k) {
            ad.modes[k].r_dry = r_dry[k][i]; ad.modes[k].stdev = stdev[k][i]; ad.modes[k].N = N_mode[k][i];
            ad.modes[k].hygroscopicity = hyg[k][i]; ad.modes[k].molar_mass_mix = mmix ? mmix[k][i] : (FT)0;
        }
        FT *na[CMX_ARG_MAX_MODES], *ma[CMX_ARG_MAX_MODES];
        for (int k = 0; k < n_modes; ++k) { na[k] = N_act ? N_act[k] + i : NULL; ma[k] = M_act ? M_act[k] + i : NULL; }
        FN(cmxo_arg2000_activation)(ap, &ad, aip, tps, th, 1, T + i, p + i, w + i, q_tot + i, q_liq ? q_liq + i : NULL,
                                    q_ice ? q_ice + i : NULL, N_liq ? N_liq + i : NULL, N_ice ? N_ice + i : NULL,
                                    N_act ? na : NULL, M_act ? ma : NULL, S_max ? S_max + i : NULL, NULL, 1);
    }
}
