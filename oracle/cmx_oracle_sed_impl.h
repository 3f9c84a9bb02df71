/*
 * cmx_oracle_sed_impl.h — oracle (TEST INFRASTRUCTURE) for the bulk sedimentation velocities a host model precomputes
 * (ClimaAtmos `set_sedimentation_precomputed_quantities`, mirrored by test/gpu_clima_core_test.jl:36-45 and the KA
 * kernel test_chen2022_terminal_velocity_kernel!, test/gpu_tests.jl:608-630).  Restates
 *   CMNonEq.terminal_velocity(::CloudLiquid, ::StokesRegimeVelType, ρ, q)        src/MicrophysicsNonEq.jl:250-265
 *   CMNonEq.terminal_velocity(::CloudIce, ::Chen2022VelTypeSmallIce, ρ, q)       src/MicrophysicsNonEq.jl:267-281
 *   CM1.terminal_velocity(::Snow, ::Chen2022VelTypeLargeIce, ρ, q)               src/Microphysics1M.jl:272-297
 * (the rain member, CM1.terminal_velocity(::Rain, ::Chen2022VelTypeRain, …), is in cmx_oracle_1m_impl.h).
 * Included from cmx_oracle_impl.h after cmx_oracle_p3_impl.h (Chen-2022 ice coefficient reductions live there).
 */
static inline FT FN(o_sed_velocity_cloud_liquid)(const TY(cmx_cloud_liquid) * cl, const TY(cmx_stokes_vel) * v, FT rho, FT q, FT eps) {
    FT pref = (FT)(1.0 / 18.0) * (v->rho_w / rho - 1) * v->grav / v->nu_air;      /* CO.particle_terminal_velocity, Common.jl:456-462 */
    FT sq = FN(o_max)((FT)0, q);
    FT D = M_CBRT((FT)(6.0 / M_PI) * rho * sq / cl->N_0 / cl->rho_w);
    return q > eps ? pref * D * D : (FT)0;
}
static inline FT FN(o_sed_velocity_cloud_ice)(const TY(cmx_cloud_ice) * ci, const TY(cmx_chen2022_small_ice_vel) * v, FT rho, FT q, FT eps) {
    FT a[2], b[2], c[2];
    FN(o_chen_small_ice)(v, rho, ci->rho_i, a, b, c);
    FT sq = FN(o_max)((FT)0, q);
    FT D = M_CBRT((FT)(6.0 / M_PI) * rho * sq / ci->N_0 / ci->rho_i);
    FT w = a[0] * M_POW(D, b[0]) * M_EXP(-c[0] * D) + a[1] * M_POW(D, b[1]) * M_EXP(-c[1] * D);
    return q > eps ? FN(o_max)((FT)0, w) : (FT)0;
}
static inline FT FN(o_sed_velocity_snow_chen)(const TY(cmx_snow) * s, const TY(cmx_chen2022_large_ice_vel) * v, FT rho, FT q, FT eps) {
    FT a[2], b[2], c[2];
    FN(o_chen_large_ice)(v, rho, s->rho_i, a, b, c);
    FT n0 = FN(o_get_n0_snow)(s, q, rho, eps);
    FT lam_inv_d = 2 * FN(o_lambda_inverse)(n0, &s->mass, q, rho, eps);
    FT pk = M_POW(s->phi, s->kappa);
    FT w = pk * FN(o_chen2022_exponential_pdf)(a[0], b[0], c[0], lam_inv_d, 3) + pk * FN(o_chen2022_exponential_pdf)(a[1], b[1], c[1], lam_inv_d, 3);
    return q > eps ? FN(o_max)((FT)0, w) : (FT)0;
}
/* oracle twin of cmx_sedimentation_velocities_* (any q / w pair may be NULL) */
void FN(cmxo_sedimentation_velocities)(const TY(cmx_microphysics_1m) * mp, const TY(cmx_stokes_vel) * stokes,
                                      const TY(cmx_chen2022_rain_vel) * chen_rain, const TY(cmx_chen2022_ice_vel) * chen_ice,
                                      const TY(cmxo_thresholds) * th, int64_t n, const FT *rho, const FT *q_lcl, const FT *q_icl,
                                      const FT *q_rai, const FT *q_sno, FT *w_lcl, FT *w_icl, FT *w_rai, FT *w_sno) {
    const FT eps = th->eps_1m;
    for (int64_t i = 0; i < n; ++i) {
        if (w_lcl) w_lcl[i] = FN(o_sed_velocity_cloud_liquid)(&mp->cloud_liquid, stokes, rho[i], q_lcl[i], eps);
        if (w_icl) w_icl[i] = FN(o_sed_velocity_cloud_ice)(&mp->cloud_ice, &chen_ice->small_ice, rho[i], q_icl[i], eps);
        if (w_rai) w_rai[i] = FN(o_terminal_velocity_rain_chen)(&mp->rain, chen_rain, rho[i], q_rai[i], eps);
        if (w_sno) w_sno[i] = FN(o_sed_velocity_snow_chen)(&mp->snow, &chen_ice->large_ice, rho[i], q_sno[i], eps);
    }
}
