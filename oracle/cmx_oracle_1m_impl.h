/*
 * cmx_oracle_1m_impl.h — oracle (TEST INFRASTRUCTURE) for the 1-moment Marshall–Palmer scheme.
 * Included from cmx_oracle_impl.h (so once per float type; FT / FN / TY / M_* macros are live).
 *
 * Restates, operation by operation: src/Microphysics1M.jl (CM1), src/MicrophysicsNonEq.jl (NonEq),
 * src/Common.jl logistic functions, src/BulkMicrophysicsTendencies.jl:141-252 (BMT) of the reference.
 * LogExpFunctions.jl (un-vendored; log1pexp, log1mexp — Common.jl:135,166,169) is restated from its
 * published definitions log(1+eˣ), log(1−eˣ) in their cancellation-free forms.
 */

static inline FT FN(o_log1pexp)(FT x) { return x > 0 ? x + M_LOG1P(M_EXP(-x)) : M_LOG1P(M_EXP(x)); }
static inline FT FN(o_log1mexp)(FT x) {   /* x < 0 */
    return x > (FT)-0.6931471805599453 ? M_LOG(-M_EXPM1(x)) : M_LOG1P(-M_EXP(x));
}
/* CO.logistic_function_integral — src/Common.jl:157-173 */
/* *scale (optional) receives the size of the two terms whose difference is returned: below the threshold the
 * result is log1pexp(kt)/k − trnslt ≈ trnslt − trnslt, so its accuracy is that of trnslt·x_0, not of the result. */
static inline FT FN(o_logistic_function_integral)(FT x, FT x_0, FT k, FT eps, FT *scale) {
    x = FN(o_max)((FT)0, x);
    FT x_safe = FN(o_max)(x, eps);
    FT x_0_safe = FN(o_max)(x_0, eps);
    FT trnslt = -FN(o_log1mexp)(-k) / k;
    FT kt = k * (x_safe / x_0_safe - 1 + trnslt);
    FT result = (FN(o_log1pexp)(kt) / k - trnslt) * x_0_safe;
    if (scale) *scale = (x < eps || x_0 < eps) ? (FT)0 : (FN(o_log1pexp)(kt) / k + trnslt) * x_0_safe;
    return x < eps ? (FT)0 : (x_0 < eps ? x : result);
}
/* TD.latent_heat_fusion: L_f(T) = (LH_s0 − LH_v0) + (cp_l − cp_i)(T − T_0)  (TDI:19) */
static inline FT FN(o_latent_heat_fusion)(const TY(cmx_thermo) * p, FT T) {
    return (p->LH_s0 - p->LH_v0) + (p->cp_l - p->cp_i) * (T - p->T_0);
}
/* TDI.supersaturation_over_ice (TDI:122-125) */
static inline FT FN(o_supersaturation_over_ice)(const TY(cmx_thermo) * p, FT q_tot, FT q_liq, FT q_ice, FT rho, FT T) {
    FT q_v = FN(o_q_vap)(q_tot, q_liq, q_ice);
    FT p_v = q_v * (rho * p->R_v * T);
    return p_v / FN(o_psat_ice)(p, T) - 1;
}
/* CO.G_func_ice — src/Common.jl:83-102 */
static inline FT FN(o_G_func_ice)(const TY(cmx_air_properties) * aps, const TY(cmx_thermo) * tps, FT T,
                                 const TY(cmxo_thresholds) * th) {
    FT R_v = tps->R_v;
    FT L = FN(o_latent_heat_sublim)(tps, T);
    FT p_vs = FN(o_psat_ice)(tps, T);
    FT p_vs_safe = FN(o_max)(p_vs, th->eps_1m);
    FT D_safe = FN(o_max)(aps->D_vapor, th->eps_1m);
    FT K_safe = FN(o_max)(aps->K_therm, th->eps_1m);
    return 1 / (L / K_safe / T * (L / R_v / T - 1) + R_v * T / D_safe / p_vs_safe);
}
/* CMNonEq._conv_q_vap_to_q_icl_const — src/MicrophysicsNonEq.jl:168-193 incl. INP_limiter :56-58 */
static inline FT FN(o_conv_q_vap_to_q_icl_const)(FT tau, const TY(cmx_thermo) * tps, FT q_tot, FT q_lcl, FT q_icl,
                                                FT q_rai, FT q_sno, FT rho, FT T, FT *scale) {
    FT R_v = tps->R_v;
    FT L_s = FN(o_latent_heat_sublim)(tps, T);
    FT cp_air = FN(o_cp_m)(tps, q_tot, q_lcl + q_rai, q_icl + q_sno);
    FT q_v = FN(o_q_vap)(q_tot, q_lcl + q_rai, q_icl + q_sno);
    FT q_sat = FN(o_qsat_ice)(tps, T, rho);
    FT dqsi_dT = q_sat * (L_s / (R_v * (T * T)) - 1 / T);
    FT Gamma_i = 1 + (L_s / cp_air) * dqsi_dT;
    FT sat_excess = q_v - q_sat;
    FT timescale = tau * Gamma_i;
    if (scale) *scale = (M_ABS(q_v) + M_ABS(q_sat)) / M_ABS(timescale);
    FT tendency = sat_excess < 0 ? -FN(o_min)(-sat_excess, FN(o_max)((FT)0, q_icl)) / timescale
                                 : sat_excess / timescale;
    int limiter = (T > tps->T_freeze) && (tendency > 0);
    return limiter ? (FT)0 : tendency;
}

/* CMI_het.INP_concentration_mean — src/IceNucleation.jl:250-253 */
static inline FT FN(o_INP_concentration_mean)(const TY(cmx_frostenberg2023) * ip, FT T) {
    FT T_celsius = FN(o_min)(T - ip->T_freeze, (FT)0);
    return 9 * M_LOG(-ip->b * T_celsius / 10) - ip->log_a;
}
/* CMNonEq.τ_relax — src/MicrophysicsNonEq.jl:32-50 (cbrt from libm, as Julia's) */
static inline FT FN(o_tau_relax)(FT rho_i, FT D_vapor, const TY(cmx_frostenberg2023) * ip, FT q_icl, FT T, FT eps) {
    FT N_icl = M_EXP(FN(o_INP_concentration_mean)(ip, T));
    FT safe_N_icl = FN(o_max)(N_icl, eps);
    FT r = N_icl > eps ? M_CBRT((3 * q_icl) / (4 * (FT)M_PI * safe_N_icl * rho_i)) : (FT)0;
    FT r0 = (FT)1e-6;
    FT r_safe = FN(o_max)(r, r0);
    return 1 / (4 * (FT)M_PI * D_vapor * N_icl * r_safe);
}
/* CMNonEq.conv_q_vap_to_q_icl(::TemperatureDependent, …) — src/MicrophysicsNonEq.jl:194-224 */
static inline FT FN(o_conv_q_vap_to_q_icl_tdep)(const TY(cmx_microphysics_1m) * mp, const TY(cmx_thermo) * tps, FT q_tot, FT q_lcl,
                                               FT q_icl, FT q_rai, FT q_sno, FT rho, FT T, FT eps, FT *scale) {
    const TY(cmx_process_params_1m) *pp = &mp->process_params;
    FT tau_sub = pp->cloud_ice_formation_tau_relax;
    FT tau_dep = FN(o_tau_relax)(mp->cloud_ice.rho_i, mp->air_properties.D_vapor, &pp->cloud_ice_formation_frostenberg, q_icl, T, eps);
    FT R_v = tps->R_v;
    FT L_s = FN(o_latent_heat_sublim)(tps, T);
    FT cp_air = FN(o_cp_m)(tps, q_tot, q_lcl + q_rai, q_icl + q_sno);
    FT q_v = FN(o_q_vap)(q_tot, q_lcl + q_rai, q_icl + q_sno);
    FT q_sat = FN(o_qsat_ice)(tps, T, rho);
    FT dqsi_dT = q_sat * (L_s / (R_v * (T * T)) - 1 / T);
    FT Gamma_i = 1 + (L_s / cp_air) * dqsi_dT;
    FT sat_excess = q_v - q_sat;
    FT sublimation_timescale = tau_sub * Gamma_i;
    FT deposition_timescale = tau_dep * Gamma_i;
    if (scale) *scale = (M_ABS(q_v) + M_ABS(q_sat)) / M_ABS(sat_excess < 0 ? sublimation_timescale : deposition_timescale);
    FT tendency = sat_excess < 0 ? -FN(o_min)(-sat_excess, FN(o_max)((FT)0, q_icl)) / sublimation_timescale
                                 : sat_excess / deposition_timescale;
    int limiter = (T > tps->T_freeze) && (tendency > 0);
    return limiter ? (FT)0 : tendency;
}

/* CM1.get_n0(::ParticlePDFSnow) — src/Microphysics1M.jl:83-86 */
static inline FT FN(o_get_n0_snow)(const TY(cmx_snow) * s, FT q_sno, FT rho, FT eps) {
    FT safe_q = FN(o_max)(q_sno, eps);
    return q_sno > eps ? s->mu * M_POW(rho * safe_q, s->nu) : (FT)0;
}
/* CM1.get_v0(::Blk1MVelTypeRain, ρ) — :101-104 */
static inline FT FN(o_get_v0_rain)(const TY(cmx_blk1m_vel_rain) * v, FT rho) {
    FT density_factor = FN(o_max)(v->rho_w / rho - 1, (FT)0);
    return M_SQRT((FT)(8.0 / 3.0) / v->C_drag * density_factor * v->grav * v->r0);
}
/* CM1.lambda_inverse — :126-152 */
static inline FT FN(o_lambda_inverse)(FT n0, const TY(cmx_particle_mass) * m, FT q, FT rho, FT eps) {
    FT qp = FN(o_max)((FT)0, q);
    FT rp = FN(o_max)((FT)0, rho);
    FT denom = m->chi_m * m->m0 * FN(o_max)(n0, eps) * m->gamma_coeff;
    FT lam_inv = M_POW(rp * qp * M_POW(m->r0, m->me + m->delta_m) / denom, 1 / (m->me + m->delta_m + 1));
    return FN(o_max)(m->r0 * (FT)1e-5, lam_inv);
}
/* CM1.terminal_velocity(precip, ::Blk1MVelType, ρ, q, v0, λ_inv) — :223-238 */
static inline FT FN(o_terminal_velocity_blk1m)(FT chi_v, FT ve, FT delta_v, FT gamma_term,
                                              const TY(cmx_particle_mass) * m, FT q, FT v0, FT lam_inv, FT eps) {
    FT fall_w = chi_v * v0 * M_POW(lam_inv / m->r0, ve + delta_v) * gamma_term / m->gamma_coeff;
    return q > eps ? fall_w : (FT)0;
}
/* CM1.terminal_velocity(rain, ::Chen2022VelTypeRain, ρ, q) — :251-270 */
static inline FT FN(o_terminal_velocity_rain_chen)(const TY(cmx_rain) * rain, const TY(cmx_chen2022_rain_vel) * c,
                                                  FT rho, FT q, FT eps) {
    FT aiu[3], bi[3], ciu[3];
    FN(o_chen2022_rain_coeffs)(c, rho, aiu, bi, ciu);
    FT lam_inv_r = FN(o_lambda_inverse)(rain->n0, &rain->mass, q, rho, eps);
    FT lam_inv_d = 2 * lam_inv_r;
    FT w = FN(o_chen2022_exponential_pdf)(aiu[0], bi[0], ciu[0], lam_inv_d, 3) +
           FN(o_chen2022_exponential_pdf)(aiu[1], bi[1], ciu[1], lam_inv_d, 3) +
           FN(o_chen2022_exponential_pdf)(aiu[2], bi[2], ciu[2], lam_inv_d, 3);
    w = FN(o_max)((FT)0, w);
    return q > eps ? w : (FT)0;
}

typedef struct TY(cmxo_sd_1m) {   /* CM1.size_distr_parameters — :375-388 */
    FT lam_inv_rai, n0_rai, v0_rai, lam_inv_sno, n0_sno, v0_sno, lam_inv_icl, n0_icl;
} TY(cmxo_sd_1m);

/* CM1.accretion kernel — :491-514 */
static inline FT FN(o_accretion_1m)(const TY(cmx_particle_mass) * pm, const TY(cmx_particle_area) * pa, FT chi_v,
                                   FT ve, FT delta_v, FT gamma_accr, FT E, FT q_clo, FT q_pre, FT n0, FT v0,
                                   FT lam_inv, FT eps) {
    FT rate = q_clo * E * n0 * pa->a0 * v0 * pa->chi_a * chi_v * lam_inv * gamma_accr /
              M_POW(pm->r0 / lam_inv, pa->ae + ve + pa->delta_a + delta_v);
    return (q_clo > eps && q_pre > eps) ? rate : (FT)0;
}
/* CM1.accretion_snow_rain kernel — :604-644 */
static inline FT FN(o_accretion_snow_rain)(const TY(cmx_particle_mass) * mass_j, FT v_ti, FT v_tj, FT E, FT coeff_disp,
                                          FT q_i, FT q_j, FT rho, FT n0_i, FT n0_j, FT li, FT lj, FT eps) {
    const FT pi = (FT)M_PI;
    FT delta = mass_j->me + mass_j->delta_m;
    FT dv = M_SQRT((v_ti - v_tj) * (v_ti - v_tj) + coeff_disp * (v_ti * v_ti + v_tj * v_tj));
    FT rate = pi / rho * n0_i * n0_j * mass_j->m0 * mass_j->chi_m * E * dv * mass_j->gamma_coeff /
              M_POW(mass_j->r0, delta) *
              (2 * (li * li * li) * M_POW(lj, delta + 1) + 2 * (delta + 1) * (li * li) * M_POW(lj, delta + 2) +
               (delta + 2) * (delta + 1) * li * M_POW(lj, delta + 3));
    return (q_i > eps && q_j > eps) ? rate : (FT)0;
}
/* CM1.warm_accretion_melt_factor — :458-465 */
static inline FT FN(o_warm_accretion_melt_factor)(const TY(cmx_thermo) * tps, FT T) {
    FT L_f = FN(o_latent_heat_fusion)(tps, T);
    return (T <= tps->T_freeze) ? (FT)0 : tps->cv_l / L_f * (T - tps->T_freeze);
}
/* ventilated Marshall–Palmer integral shared by rain evaporation, snow sublimation/deposition and snow melt
 * (CM1:948-956, 1025-1033, 1127-1135): a + b ∛Sc / (r0/λ⁻¹)^((ve+Δv)/2) √(2 v0 χv/ν λ⁻¹) Γ_vent */
static inline FT FN(o_vent_factor)(FT a, FT b, FT Sc, FT r0, FT lam_inv, FT ve, FT dv, FT v0, FT chi_v, FT nu_air,
                                  FT gamma_vent) {
    return a + b * M_CBRT(Sc) / M_POW(r0 / lam_inv, (ve + dv) / 2) * M_SQRT(2 * v0 * chi_v / nu_air * lam_inv) *
                   gamma_vent;
}

/* s[] = the 18 source terms; scale_* = size of the cancelling operands of the terms that are small differences of
 * large quantities: q_v − q_sat (vap_lcl, vap_icl), S = p_v/p_sat − 1 (vap_rai, vap_sno), T − T_freeze (the four
 * melt terms).  Every other term is a product of positive factors (scale = |term|). */
typedef struct TY(cmxo_src_1m) {
    FT s[CMX_MP1M_NSRC];
    FT scale_vap_lcl, scale_vap_icl, scale_vap_rai, scale_vap_sno;
    FT scale_melt_icl, scale_melt_sno, scale_accr_melt_lcl_sno, scale_accr_melt_rai_sno;
    FT scale_acnv_rai, scale_acnv_sno;   /* Kessler-type logistic integrals (0 for the other variants: |term| is used) */
} TY(cmxo_src_1m);

/* _microphysics_source_terms — src/BulkMicrophysicsTendencies.jl:141-217 */
static inline TY(cmxo_src_1m) FN(o_source_terms_1m)(const TY(cmx_microphysics_1m) * mp, const TY(cmx_thermo) * tps,
                                                   uint32_t flags, const TY(cmxo_thresholds) * th, FT rho, FT T,
                                                   FT q_tot, FT q_lcl, FT q_icl, FT q_rai, FT q_sno) {
    TY(cmxo_src_1m) o;
    const FT pi = (FT)M_PI;
    const FT eps = th->eps_1m;
    const TY(cmx_process_params_1m) *pp = &mp->process_params;
    const TY(cmx_air_properties) *aps = &mp->air_properties;
    for (int k = 0; k < CMX_MP1M_NSRC; ++k) o.s[k] = 0;
    o.scale_vap_lcl = o.scale_vap_icl = o.scale_vap_rai = o.scale_vap_sno = 0;
    o.scale_melt_icl = o.scale_melt_sno = o.scale_accr_melt_lcl_sno = o.scale_accr_melt_rai_sno = 0;
    o.scale_acnv_rai = o.scale_acnv_sno = 0;
    /* (|T| + T_freeze)/|T − T_freeze| carried through the melt prefactors: size of the operands of T − T_freeze */
    const FT dT_amp = M_ABS(T) + tps->T_freeze;
    rho = FN(o_max)((FT)0, rho);               /* BMT:147-152 */
    q_tot = FN(o_max)((FT)0, q_tot);
    q_lcl = FN(o_max)((FT)0, q_lcl);
    q_icl = FN(o_max)((FT)0, q_icl);
    q_rai = FN(o_max)((FT)0, q_rai);
    q_sno = FN(o_max)((FT)0, q_sno);
    /* size_distr_parameters — CM1:375-388 */
    TY(cmxo_sd_1m) sd;
    sd.n0_rai = mp->rain.n0;
    sd.lam_inv_rai = FN(o_lambda_inverse)(sd.n0_rai, &mp->rain.mass, q_rai, rho, eps);
    sd.v0_rai = FN(o_get_v0_rain)(&mp->vel_rain, rho);
    sd.n0_sno = FN(o_get_n0_snow)(&mp->snow, q_sno, rho, eps);
    sd.lam_inv_sno = FN(o_lambda_inverse)(sd.n0_sno, &mp->snow.mass, q_sno, rho, eps);
    sd.v0_sno = mp->vel_snow.v0;
    sd.n0_icl = mp->cloud_ice.n0;
    sd.lam_inv_icl = FN(o_lambda_inverse)(sd.n0_icl, &mp->cloud_ice.mass, q_icl, rho, eps);

    if (flags & CMX_1M_CLOUD_LIQUID_FORMATION)
        o.s[CMX_1M_S_PHASE_CHANGE_VAP_LCL] = FN(o_conv_q_vap_to_q_lcl_const)(
            pp->cloud_liquid_formation_tau_relax, tps, q_tot, q_lcl, q_icl, q_rai, q_sno, rho, T, &o.scale_vap_lcl);
    if (flags & CMX_1M_CLOUD_ICE_FORMATION_CONST)
        o.s[CMX_1M_S_PHASE_CHANGE_VAP_ICL] = FN(o_conv_q_vap_to_q_icl_const)(
            pp->cloud_ice_formation_tau_relax, tps, q_tot, q_lcl, q_icl, q_rai, q_sno, rho, T, &o.scale_vap_icl);
    else if (flags & CMX_1M_CLOUD_ICE_FORMATION_TDEP)
        o.s[CMX_1M_S_PHASE_CHANGE_VAP_ICL] =
            FN(o_conv_q_vap_to_q_icl_tdep)(mp, tps, q_tot, q_lcl, q_icl, q_rai, q_sno, rho, T, eps, &o.scale_vap_icl);
    /* rain autoconversion — CM1:354-364 */
    if (flags & CMX_1M_RAIN_ACNV_KESSLER) {
        o.s[CMX_1M_S_ACNV_LCL_RAI] = FN(o_logistic_function_integral)(q_lcl, pp->rain_autoconversion.q_threshold,
                                                                     pp->rain_autoconversion.k, eps, &o.scale_acnv_rai) /
                                     pp->rain_autoconversion.tau;
        o.scale_acnv_rai /= pp->rain_autoconversion.tau;
    } else if (flags & CMX_1M_RAIN_ACNV_PRESCRIBED_ND)
        o.s[CMX_1M_S_ACNV_LCL_RAI] = FN(o_max)((FT)0, q_lcl) /
                                     (pp->rain_autoconversion_nd.tau *
                                      M_POW(pp->rain_autoconversion_nd.Nc / 100000000, pp->rain_autoconversion_nd.alpha));
    /* snow autoconversion — CM1:414-446 */
    if (flags & CMX_1M_SNOW_ACNV_NO_SUPERSAT) {
        o.s[CMX_1M_S_ACNV_ICL_SNO] = FN(o_logistic_function_integral)(q_icl, pp->snow_autoconversion.q_threshold,
                                                                     pp->snow_autoconversion.k, eps, &o.scale_acnv_sno) /
                                     pp->snow_autoconversion.tau;
        o.scale_acnv_sno /= pp->snow_autoconversion.tau;
    } else if (flags & CMX_1M_SNOW_ACNV_WITH_SUPERSAT) {
        FT S = FN(o_supersaturation_over_ice)(tps, q_tot, q_lcl + q_rai, q_icl + q_sno, rho, T);
        FT G = FN(o_G_func_ice)(aps, tps, T, th);
        FT li = sd.lam_inv_icl, r_is = pp->r_ice_snow;
        FT rate = 4 * pi * S * G * sd.n0_icl / rho * M_EXP(-r_is / li) *
                  (r_is * r_is / (mp->cloud_ice.mass.me + mp->cloud_ice.mass.delta_m) + (r_is / li + 1) * (li * li));
        o.s[CMX_1M_S_ACNV_ICL_SNO] = (q_icl > eps && S > 0 && T < tps->T_freeze) ? rate : (FT)0;
        /* ∝ S = p_v/p_sat − 1: operand-sized scale, as for the other supersaturation-driven terms */
        o.scale_acnv_sno = (q_icl > eps && T < tps->T_freeze && S != 0) ? M_ABS(rate / S * (S + 2)) : (FT)0;
    }
    const int is_warm = T >= tps->T_freeze;    /* BMT:171 */
    const TY(cmx_blk1m_vel_rain) *vr = &mp->vel_rain;
    const TY(cmx_blk1m_vel_snow) *vs = &mp->vel_snow;
    if (flags & CMX_1M_ACCR_LCL_RAI)
        o.s[CMX_1M_S_ACCR_LCL_RAI] =
            FN(o_accretion_1m)(&mp->rain.mass, &mp->rain.area, vr->chi_v, vr->ve, vr->delta_v, vr->gamma_accr,
                               pp->e_lcl_rai, q_lcl, q_rai, sd.n0_rai, sd.v0_rai, sd.lam_inv_rai, eps);
    if (flags & CMX_1M_ACCR_LCL_SNO) {
        FT S = FN(o_accretion_1m)(&mp->snow.mass, &mp->snow.area, vs->chi_v, vs->ve, vs->delta_v, vs->gamma_accr,
                                  pp->e_lcl_sno, q_lcl, q_sno, sd.n0_sno, sd.v0_sno, sd.lam_inv_sno, eps);
        FT alpha = FN(o_warm_accretion_melt_factor)(tps, T);
        o.s[CMX_1M_S_ACCR_LCL_SNO_COLD] = is_warm ? (FT)0 : S;
        o.s[CMX_1M_S_ACCR_LCL_SNO_WARM] = is_warm ? S : (FT)0;
        o.s[CMX_1M_S_ACCR_MELT_LCL_SNO] = alpha * S;
        o.scale_accr_melt_lcl_sno = M_ABS(tps->cv_l / FN(o_latent_heat_fusion)(tps, T) * dT_amp * S);
    }
    if (flags & CMX_1M_ACCR_ICL_RAI) {
        o.s[CMX_1M_S_ACCR_ICL_RAI] =
            FN(o_accretion_1m)(&mp->rain.mass, &mp->rain.area, vr->chi_v, vr->ve, vr->delta_v, vr->gamma_accr,
                               pp->e_icl_rai, q_icl, q_rai, sd.n0_rai, sd.v0_rai, sd.lam_inv_rai, eps);
        /* accretion_rain_sink — CM1:535-561 */
        const TY(cmx_particle_mass) *m = &mp->rain.mass;
        const TY(cmx_particle_area) *a = &mp->rain.area;
        FT rate = pp->e_icl_rai / rho * sd.n0_rai * sd.n0_icl * m->m0 * a->a0 * sd.v0_rai * m->chi_m * a->chi_a *
                  vr->chi_v * sd.lam_inv_icl * sd.lam_inv_rai * vr->gamma_accr_rain_sink /
                  M_POW(m->r0 / sd.lam_inv_rai, m->me + a->ae + vr->ve + m->delta_m + a->delta_a + vr->delta_v);
        o.s[CMX_1M_S_ACCR_FREEZE_ICL_RAI] = (q_icl > eps && q_rai > eps) ? rate : (FT)0;
    }
    if (flags & CMX_1M_ACCR_ICL_SNO)
        o.s[CMX_1M_S_ACCR_ICL_SNO] =
            FN(o_accretion_1m)(&mp->snow.mass, &mp->snow.area, vs->chi_v, vs->ve, vs->delta_v, vs->gamma_accr,
                               pp->e_icl_sno, q_icl, q_sno, sd.n0_sno, sd.v0_sno, sd.lam_inv_sno, eps);
    if (flags & CMX_1M_ACCR_RAI_SNO) {   /* CM1:815-867 */
        FT v_sno = FN(o_terminal_velocity_blk1m)(vs->chi_v, vs->ve, vs->delta_v, vs->gamma_term, &mp->snow.mass, q_sno,
                                                sd.v0_sno, sd.lam_inv_sno, eps);
        FT v_rai = FN(o_terminal_velocity_blk1m)(vr->chi_v, vr->ve, vr->delta_v, vr->gamma_term, &mp->rain.mass, q_rai,
                                                sd.v0_rai, sd.lam_inv_rai, eps);
        /* S_rai_sno: type_i = snow, type_j = rain ; S_sno_rai: type_i = rain, type_j = snow */
        FT S_rai_sno = FN(o_accretion_snow_rain)(&mp->rain.mass, v_sno, v_rai, pp->e_rai_sno, pp->coeff_disp, q_sno,
                                                q_rai, rho, sd.n0_sno, sd.n0_rai, sd.lam_inv_sno, sd.lam_inv_rai, eps);
        FT S_sno_rai = FN(o_accretion_snow_rain)(&mp->snow.mass, v_rai, v_sno, pp->e_rai_sno, pp->coeff_disp, q_rai,
                                                q_sno, rho, sd.n0_rai, sd.n0_sno, sd.lam_inv_rai, sd.lam_inv_sno, eps);
        FT alpha = FN(o_warm_accretion_melt_factor)(tps, T);
        FT S_melt = alpha * S_rai_sno;
        o.s[CMX_1M_S_ACCR_RAI_SNO_COLD] = is_warm ? (FT)0 : S_rai_sno;
        o.s[CMX_1M_S_ACCR_RAI_SNO_WARM] = is_warm ? S_sno_rai : (FT)0;
        o.s[CMX_1M_S_ACCR_MELT_RAI_SNO] = is_warm ? S_melt : (FT)0;
        o.scale_accr_melt_rai_sno = M_ABS(tps->cv_l / FN(o_latent_heat_fusion)(tps, T) * dT_amp * S_rai_sno);
    }
    const FT Sc = aps->nu_air / FN(o_max)(aps->D_vapor, eps);
    if (flags & CMX_1M_RAIN_EVAPORATION) {   /* CM1:917-960 */
        FT S = FN(o_supersaturation_over_liquid)(tps, q_tot, q_lcl + q_rai, q_icl + q_sno, rho, T);
        FT G = FN(o_G_func_liquid)(aps, tps, T, th);
        FT li = sd.lam_inv_rai;
        FT F = FN(o_vent_factor)(mp->rain.vent.a, mp->rain.vent.b, Sc, mp->rain.mass.r0, li, vr->ve, vr->delta_v,
                                 sd.v0_rai, vr->chi_v, aps->nu_air, vr->gamma_vent);
        FT rate = 4 * pi * sd.n0_rai / rho * S * G * (li * li) * F;
        o.s[CMX_1M_S_PHASE_CHANGE_VAP_RAI] = FN(o_min)((FT)0, (q_rai > eps && S < 0) ? rate : (FT)0);
        o.scale_vap_rai = (q_rai > eps) ? M_ABS(4 * pi * sd.n0_rai / rho * (S + 2) * G * (li * li) * F) : (FT)0;
    }
    if (flags & (CMX_1M_SNOW_SUBLIMATION_ONLY | CMX_1M_SNOW_DEP_AND_SUBL)) {   /* CM1:979-1037 */
        FT S = FN(o_supersaturation_over_ice)(tps, q_tot, q_lcl + q_rai, q_icl + q_sno, rho, T);
        FT G = FN(o_G_func_ice)(aps, tps, T, th);
        FT li = sd.lam_inv_sno;
        FT F = FN(o_vent_factor)(mp->snow.vent.a, mp->snow.vent.b, Sc, mp->snow.mass.r0, li, vs->ve, vs->delta_v,
                                 sd.v0_sno, vs->chi_v, aps->nu_air, vs->gamma_vent);
        FT rate = (q_sno > eps) ? 4 * pi * sd.n0_sno / rho * S * G * (li * li) * F : (FT)0;
        o.s[CMX_1M_S_PHASE_CHANGE_VAP_SNO] = (flags & CMX_1M_SNOW_DEP_AND_SUBL) ? rate : FN(o_min)((FT)0, rate);
        o.scale_vap_sno = (q_sno > eps) ? M_ABS(4 * pi * sd.n0_sno / rho * (S + 2) * G * (li * li) * F) : (FT)0;
    }
    if (flags & CMX_1M_CLOUD_ICE_MELT) {   /* CM1:1055-1077 */
        FT L = FN(o_latent_heat_fusion)(tps, T);
        FT li = sd.lam_inv_icl;
        FT rate = 4 * pi * mp->cloud_ice.n0 / rho * aps->K_therm / L * (T - tps->T_freeze) * (li * li);
        o.s[CMX_1M_S_MELT_ICL_LCL] = (q_icl > eps && T > tps->T_freeze) ? rate : (FT)0;
        o.scale_melt_icl = (q_icl > eps) ? M_ABS(4 * pi * mp->cloud_ice.n0 / rho * aps->K_therm / L * dT_amp * (li * li)) : (FT)0;
    }
    if (flags & CMX_1M_SNOW_MELT) {   /* CM1:1094-1139 */
        FT L = FN(o_latent_heat_fusion)(tps, T);
        FT li = sd.lam_inv_sno;
        FT F = FN(o_vent_factor)(mp->snow.vent.a, mp->snow.vent.b, Sc, mp->snow.mass.r0, li, vs->ve, vs->delta_v,
                                 sd.v0_sno, vs->chi_v, aps->nu_air, vs->gamma_vent);
        FT rate = 4 * pi * sd.n0_sno / rho * aps->K_therm / L * (T - tps->T_freeze) * (li * li) * F;
        o.s[CMX_1M_S_MELT_SNO_RAI] = (q_sno > eps && T > tps->T_freeze) ? rate : (FT)0;
        o.scale_melt_sno = (q_sno > eps) ? M_ABS(4 * pi * sd.n0_sno / rho * aps->K_therm / L * dT_amp * (li * li) * F) : (FT)0;
    }
    return o;
}

/* _aggregate_tendencies — BMT:227-252; out[4] = dq_lcl, dq_icl, dq_rai, dq_sno; scale[4] = Σ|terms| (with the
 * cancellation-aware scales of the four vapour phase-change terms) */
static inline void FN(o_aggregate_1m)(const TY(cmxo_src_1m) * src, FT out[4], FT scale[4]) {
    const FT *s = src->s;
    out[0] = s[CMX_1M_S_PHASE_CHANGE_VAP_LCL] - s[CMX_1M_S_ACNV_LCL_RAI] - s[CMX_1M_S_ACCR_LCL_RAI] -
             s[CMX_1M_S_ACCR_LCL_SNO_COLD] - s[CMX_1M_S_ACCR_LCL_SNO_WARM] + s[CMX_1M_S_MELT_ICL_LCL];
    out[1] = s[CMX_1M_S_PHASE_CHANGE_VAP_ICL] - s[CMX_1M_S_ACNV_ICL_SNO] - s[CMX_1M_S_ACCR_ICL_RAI] -
             s[CMX_1M_S_ACCR_ICL_SNO] - s[CMX_1M_S_MELT_ICL_LCL];
    out[2] = s[CMX_1M_S_ACNV_LCL_RAI] + s[CMX_1M_S_ACCR_LCL_RAI] + s[CMX_1M_S_ACCR_LCL_SNO_WARM] +
             s[CMX_1M_S_ACCR_MELT_LCL_SNO] - s[CMX_1M_S_ACCR_FREEZE_ICL_RAI] - s[CMX_1M_S_ACCR_RAI_SNO_COLD] +
             s[CMX_1M_S_ACCR_RAI_SNO_WARM] + s[CMX_1M_S_ACCR_MELT_RAI_SNO] + s[CMX_1M_S_PHASE_CHANGE_VAP_RAI] +
             s[CMX_1M_S_MELT_SNO_RAI];
    out[3] = s[CMX_1M_S_ACNV_ICL_SNO] + s[CMX_1M_S_ACCR_LCL_SNO_COLD] - s[CMX_1M_S_ACCR_MELT_LCL_SNO] +
             s[CMX_1M_S_ACCR_ICL_RAI] + s[CMX_1M_S_ACCR_FREEZE_ICL_RAI] + s[CMX_1M_S_ACCR_ICL_SNO] +
             s[CMX_1M_S_ACCR_RAI_SNO_COLD] - s[CMX_1M_S_ACCR_RAI_SNO_WARM] - s[CMX_1M_S_ACCR_MELT_RAI_SNO] +
             s[CMX_1M_S_PHASE_CHANGE_VAP_SNO] - s[CMX_1M_S_MELT_SNO_RAI];
    if (scale) {
#define A(k) M_ABS(s[k])
        const FT a_rai = FN(o_max)(A(CMX_1M_S_ACNV_LCL_RAI), src->scale_acnv_rai);
        const FT a_sno = FN(o_max)(A(CMX_1M_S_ACNV_ICL_SNO), src->scale_acnv_sno);
        scale[0] = src->scale_vap_lcl + a_rai + A(CMX_1M_S_ACCR_LCL_RAI) + A(CMX_1M_S_ACCR_LCL_SNO_COLD) +
                   A(CMX_1M_S_ACCR_LCL_SNO_WARM) + src->scale_melt_icl;
        scale[1] = src->scale_vap_icl + a_sno + A(CMX_1M_S_ACCR_ICL_RAI) + A(CMX_1M_S_ACCR_ICL_SNO) +
                   src->scale_melt_icl;
        scale[2] = a_rai + A(CMX_1M_S_ACCR_LCL_RAI) + A(CMX_1M_S_ACCR_LCL_SNO_WARM) +
                   src->scale_accr_melt_lcl_sno + A(CMX_1M_S_ACCR_FREEZE_ICL_RAI) + A(CMX_1M_S_ACCR_RAI_SNO_COLD) +
                   A(CMX_1M_S_ACCR_RAI_SNO_WARM) + src->scale_accr_melt_rai_sno + src->scale_vap_rai +
                   src->scale_melt_sno;
        scale[3] = a_sno + A(CMX_1M_S_ACCR_LCL_SNO_COLD) + src->scale_accr_melt_lcl_sno +
                   A(CMX_1M_S_ACCR_ICL_RAI) + A(CMX_1M_S_ACCR_FREEZE_ICL_RAI) + A(CMX_1M_S_ACCR_ICL_SNO) +
                   A(CMX_1M_S_ACCR_RAI_SNO_COLD) + A(CMX_1M_S_ACCR_RAI_SNO_WARM) + src->scale_accr_melt_rai_sno +
                   src->scale_vap_sno + src->scale_melt_sno;
#undef A
    }
}

/* ---- exported array drivers ---- */
/* oracle twin of cmx_mp1m_tendencies_* and cmx_mp1m_source_terms_*: `tend`/`scale` = 4 optional columns each,
 * `src` = CMX_MP1M_NSRC optional columns; near_T_freeze marks |T − T_freeze| ≤ margin (the is_warm routing is a
 * genuine discontinuity: a state within rounding of T_freeze may land on either arm in another precision). */
void FN(cmxo_mp1m)(const TY(cmx_microphysics_1m) * mp, const TY(cmx_thermo) * tps, uint32_t flags,
                  const TY(cmxo_thresholds) * th, int64_t n, const FT *rho, const FT *T, const FT *q_tot,
                  const FT *q_lcl, const FT *q_icl, const FT *q_rai, const FT *q_sno, FT *const *tend, FT *const *scale,
                  FT *const *src, int32_t nthreads) {
    (void)nthreads;
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
    for (int64_t i = 0; i < n; ++i) {
        TY(cmxo_src_1m) s = FN(o_source_terms_1m)(mp, tps, flags, th, rho[i], T[i], q_tot[i], q_lcl[i], q_icl[i],
                                                  q_rai[i], q_sno[i]);
        FT t[4], sc[4];
        FN(o_aggregate_1m)(&s, t, sc);
        for (int k = 0; k < 4; ++k) {
            if (tend && tend[k]) tend[k][i] = t[k];
            if (scale && scale[k]) scale[k][i] = sc[k];
        }
        if (src)
            for (int k = 0; k < CMX_MP1M_NSRC; ++k)
                if (src[k]) src[k][i] = s.s[k];
    }
}

/* ---- LinearizedAverage mode — src/BulkMicrophysicsTendencies.jl:255-465, 572-632 ------------------------------ */
typedef struct TY(cmxo_lin_1m) { FT M11, M12, M22, M31, M33, M34, M41, M42, M43, M44, e1, e2, e4; } TY(cmxo_lin_1m);
/* _linearize — :269-379 (donor-based: D = S / max(q_min, q_donor)) */
static inline TY(cmxo_lin_1m) FN(o_linearize_1m)(const FT *S, FT q_lcl, FT q_icl, FT q_rai, FT q_sno, FT q_min) {
    TY(cmxo_lin_1m) L = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const FT dl = FN(o_max)(q_min, q_lcl), di = FN(o_max)(q_min, q_icl), dr = FN(o_max)(q_min, q_rai), ds = FN(o_max)(q_min, q_sno);
    FT D;
    D = S[CMX_1M_S_PHASE_CHANGE_VAP_LCL] / dl;
    if (S[CMX_1M_S_PHASE_CHANGE_VAP_LCL] >= 0) L.e1 += S[CMX_1M_S_PHASE_CHANGE_VAP_LCL]; else L.M11 += D;
    D = S[CMX_1M_S_PHASE_CHANGE_VAP_ICL] / di;
    if (S[CMX_1M_S_PHASE_CHANGE_VAP_ICL] >= 0) L.e2 += S[CMX_1M_S_PHASE_CHANGE_VAP_ICL]; else L.M22 += D;
    D = S[CMX_1M_S_MELT_ICL_LCL] / di;          L.M22 -= D; L.M12 += D;
    D = S[CMX_1M_S_ACNV_LCL_RAI] / dl;          L.M11 -= D; L.M31 += D;
    D = S[CMX_1M_S_ACNV_ICL_SNO] / di;          L.M22 -= D; L.M42 += D;
    D = S[CMX_1M_S_ACCR_LCL_RAI] / dl;          L.M11 -= D; L.M31 += D;
    {
        FT Dc = S[CMX_1M_S_ACCR_LCL_SNO_COLD] / dl, Dw = S[CMX_1M_S_ACCR_LCL_SNO_WARM] / dl;
        L.M11 -= Dc + Dw; L.M31 += Dw; L.M41 += Dc;
    }
    D = S[CMX_1M_S_ACCR_MELT_LCL_SNO] / ds;     L.M44 -= D; L.M34 += D;
    D = S[CMX_1M_S_ACCR_ICL_RAI] / di;          L.M22 -= D; L.M42 += D;
    D = S[CMX_1M_S_ACCR_ICL_SNO] / di;          L.M22 -= D; L.M42 += D;
    D = S[CMX_1M_S_ACCR_FREEZE_ICL_RAI] / dr;   L.M33 -= D; L.M43 += D;
    D = S[CMX_1M_S_ACCR_RAI_SNO_WARM] / ds;     L.M44 -= D; L.M34 += D;
    D = S[CMX_1M_S_ACCR_MELT_RAI_SNO] / ds;     L.M44 -= D; L.M34 += D;
    D = S[CMX_1M_S_ACCR_RAI_SNO_COLD] / dr;     L.M33 -= D; L.M43 += D;
    D = (-S[CMX_1M_S_PHASE_CHANGE_VAP_RAI]) / dr; L.M33 -= D;
    D = S[CMX_1M_S_PHASE_CHANGE_VAP_SNO] / ds;
    if (S[CMX_1M_S_PHASE_CHANGE_VAP_SNO] >= 0) L.e4 += S[CMX_1M_S_PHASE_CHANGE_VAP_SNO]; else L.M44 += D;
    D = S[CMX_1M_S_MELT_SNO_RAI] / ds;          L.M44 -= D; L.M34 += D;
    return L;
}
/* _linearized_implicit_step — :381-465; out = (dq_lcl_dt, dq_icl_dt, dq_rai_dt, dq_sno_dt) */
static inline void FN(o_linearized_implicit_step_1m)(const TY(cmx_microphysics_1m) * mp, const TY(cmx_thermo) * tps, uint32_t flags,
                                                    const TY(cmxo_thresholds) * th, FT q_min, FT rho, FT T, FT q_tot, FT q_lcl, FT q_icl,
                                                    FT q_rai, FT q_sno, FT dt, FT out[4]) {
    TY(cmxo_src_1m) src = FN(o_source_terms_1m)(mp, tps, flags, th, rho, T, q_tot, q_lcl, q_icl, q_rai, q_sno);
    TY(cmxo_lin_1m) L = FN(o_linearize_1m)(src.s, q_lcl, q_icl, q_rai, q_sno, q_min);
    const FT inv_dt = 1 / dt;
    const FT q_sat_min = FN(o_min)(FN(o_psat_liquid)(tps, T) / (rho * tps->R_v * T), FN(o_psat_ice)(tps, T) / (rho * tps->R_v * T));
    const FT q_v = q_tot - q_lcl - q_icl - q_rai - q_sno;
    const FT alpha = FN(o_min)((FT)1, FN(o_max)((FT)0, q_v - q_sat_min) * inv_dt / FN(o_max)(L.e1 + L.e2 + L.e4, th->eps_ft));
    const FT a11 = inv_dt - L.M11, a12 = -L.M12, a22 = inv_dt - L.M22, a31 = -L.M31, a33 = inv_dt - L.M33, a34 = -L.M34;
    const FT a41 = -L.M41, a42 = -L.M42, a43 = -L.M43, a44 = inv_dt - L.M44;
    const FT b1 = alpha * L.e1 + inv_dt * q_lcl, b2 = alpha * L.e2 + inv_dt * q_icl, b3 = inv_dt * q_rai, b4 = alpha * L.e4 + inv_dt * q_sno;
    const FT det12 = a11 * a22;
    const FT q_lcl_new = (b1 * a22 - a12 * b2) / det12, q_icl_new = a11 * b2 / det12;
    const FT r3 = M_FMA(-a31, q_lcl_new, b3);
    const FT r4 = M_FMA(-a41, q_lcl_new, M_FMA(-a42, q_icl_new, b4));
    const FT det = M_FMA(-a34, a43, a33 * a44);
    const FT q_rai_new = (r3 * a44 - a34 * r4) / det, q_sno_new = (a33 * r4 - r3 * a43) / det;
    out[0] = (q_lcl_new - q_lcl) * inv_dt; out[1] = (q_icl_new - q_icl) * inv_dt;
    out[2] = (q_rai_new - q_rai) * inv_dt; out[3] = (q_sno_new - q_sno) * inv_dt;
}
/* bulk_microphysics_tendencies(::LinearizedAverage, …, Δt, nsub) — :572-632; oracle twin of cmx_mp1m_linearized_average_* */
void FN(cmxo_mp1m_linearized_average)(const TY(cmx_microphysics_1m) * mp, const TY(cmx_thermo) * tps, uint32_t flags,
                                     const TY(cmxo_thresholds) * th, FT q_min, FT dt, int32_t nsub, int64_t n, const FT *rho,
                                     const FT *T, const FT *q_tot, const FT *q_lcl, const FT *q_icl, const FT *q_rai,
                                     const FT *q_sno, FT *const *tend, int32_t nthreads) {
    (void)nthreads;
    const FT dt_sub = dt / (FT)nsub;
    const FT Lv_over_cp = tps->LH_v0 / tps->cp_d, Ls_over_cp = tps->LH_s0 / tps->cp_d;
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
    for (int64_t i = 0; i < n; ++i) {
        FT Ti = T[i], ql = q_lcl[i], qi = q_icl[i], qr = q_rai[i], qs = q_sno[i];
        for (int k = 0; k < nsub; ++k) {
            FT r[4];
            FN(o_linearized_implicit_step_1m)(mp, tps, flags, th, q_min, rho[i], Ti, q_tot[i], ql, qi, qr, qs, dt_sub, r);
            ql += r[0] * dt_sub; qi += r[1] * dt_sub; qr += r[2] * dt_sub; qs += r[3] * dt_sub;
            Ti += (Lv_over_cp * (r[0] + r[2]) + Ls_over_cp * (r[1] + r[3])) * dt_sub;
        }
        tend[0][i] = (ql - q_lcl[i]) / dt; tend[1][i] = (qi - q_icl[i]) / dt;
        tend[2][i] = (qr - q_rai[i]) / dt; tend[3][i] = (qs - q_sno[i]) / dt;
    }
}
/* probe for the reference's "solves the linearized system" test (test/bulk_tendencies_tests.jl:850-882): M and e */
void FN(cmxo_mp1m_linearize)(const TY(cmx_microphysics_1m) * mp, const TY(cmx_thermo) * tps, uint32_t flags, const TY(cmxo_thresholds) * th,
                            FT q_min, FT rho, FT T, FT q_tot, FT q_lcl, FT q_icl, FT q_rai, FT q_sno, FT out[13]) {
    TY(cmxo_src_1m) src = FN(o_source_terms_1m)(mp, tps, flags, th, rho, T, q_tot, q_lcl, q_icl, q_rai, q_sno);
    TY(cmxo_lin_1m) L = FN(o_linearize_1m)(src.s, q_lcl, q_icl, q_rai, q_sno, q_min);
    const FT v[13] = {L.M11, L.M12, L.M22, L.M31, L.M33, L.M34, L.M41, L.M42, L.M43, L.M44, L.e1, L.e2, L.e4};
    for (int k = 0; k < 13; ++k) out[k] = v[k];
}

/* oracle twin of cmx_mp1m_terminal_velocity_* */
void FN(cmxo_mp1m_terminal_velocity)(const TY(cmx_microphysics_1m) * mp, const TY(cmx_chen2022_rain_vel) * chen,
                                    const TY(cmxo_thresholds) * th, int64_t n, const FT *rho, const FT *q_rai,
                                    const FT *q_sno, FT *vt_rai_blk1m, FT *vt_sno_blk1m, FT *vt_rai_chen) {
    const FT eps = th->eps_1m;
    const TY(cmx_blk1m_vel_rain) *vr = &mp->vel_rain;
    const TY(cmx_blk1m_vel_snow) *vs = &mp->vel_snow;
    for (int64_t i = 0; i < n; ++i) {
        if (vt_rai_blk1m) {
            FT v0 = FN(o_get_v0_rain)(vr, rho[i]);
            FT li = FN(o_lambda_inverse)(mp->rain.n0, &mp->rain.mass, q_rai[i], rho[i], eps);
            vt_rai_blk1m[i] = FN(o_terminal_velocity_blk1m)(vr->chi_v, vr->ve, vr->delta_v, vr->gamma_term, &mp->rain.mass,
                                                           q_rai[i], v0, li, eps);
        }
        if (vt_sno_blk1m) {
            FT n0 = FN(o_get_n0_snow)(&mp->snow, q_sno[i], rho[i], eps);
            FT li = FN(o_lambda_inverse)(n0, &mp->snow.mass, q_sno[i], rho[i], eps);
            vt_sno_blk1m[i] = FN(o_terminal_velocity_blk1m)(vs->chi_v, vs->ve, vs->delta_v, vs->gamma_term, &mp->snow.mass,
                                                           q_sno[i], vs->v0, li, eps);
        }
        if (vt_rai_chen) vt_rai_chen[i] = FN(o_terminal_velocity_rain_chen)(&mp->rain, chen, rho[i], q_rai[i], eps);
    }
}
FT FN(cmxo_logistic_function_integral)(FT x, FT x_0, FT k, FT eps) { return FN(o_logistic_function_integral)(x, x_0, k, eps, NULL); }

/* ---- 0-moment scheme: CM0.remove_precipitation / ∂remove_precipitation_∂q_tot — src/Microphysics0M.jl:35-75, behind the 0M methods
 * of bulk_microphysics_tendencies (src/BulkMicrophysicsTendencies.jl:658-680: both condensate inputs clamped to ≥ 0 first).
 * q_vap_sat == NULL selects the qc_0 threshold, otherwise S_0·q_vap_sat. */
void FN(cmxo_mp0m_tendencies)(const TY(cmx_parameters_0m) * p, int64_t n, const FT *q_lcl, const FT *q_icl, const FT *q_vap_sat,
                              FT *dq_tot_dt, FT *ddq_dq_tot) {
    for (int64_t i = 0; i < n; ++i) {
        FT ql = q_lcl[i] > 0 ? q_lcl[i] : (FT)0, qi = q_icl[i] > 0 ? q_icl[i] : (FT)0;
        FT thr = q_vap_sat ? p->S_0 * q_vap_sat[i] : p->qc_0;
        FT ex = ql + qi - thr;
        dq_tot_dt[i] = -(ex > 0 ? ex : (FT)0) / p->tau_precip;
        if (ddq_dq_tot) ddq_dq_tot[i] = ql + qi > thr ? (FT)-1 / p->tau_precip : (FT)0;
    }
}
