/* cmx_oracle_p3col_impl.h — P3 liquid–ice collisions (src/P3_processes.jl:96-655) and the immersion / deposition
 * ice-nucleation rates of the 2M+P3 entry (src/IceNucleation.jl:136-511).  TEST INFRASTRUCTURE, included by
 * cmx_oracle_impl.h after cmx_oracle_p3_impl.h.  Pinned by the reference's own KATs: the ten collision integrals,
 * compute_max_freeze_rate and the local rime density of test/p3_tests.jl:695-870 (tests/golden/p3_kats.json), P3_het_N_i,
 * P3_deposition_N_i (test/gpu_tests.jl:1001-1037) and the Frostenberg means (test/heterogeneous_ice_nucleation_tests.jl:
 * 246-275).
 *
 * RootSolvers.jl (compat "0.3, 0.4, 1", un-vendored): crossover_diameter's Brent solve uses the same restated Brent
 * iteration as the shape solver (o_brent_fixed, cmx_oracle_p3_impl.h); without a sign change the end point with the smaller |f| is
 * returned, which is what the reference's own test demands (test/p3_tests.jl:1057-1058: D_min for a target below the
 * curve, D_max above it). */

/* LocalRimeDensity functor — src/parameters/MicrophysicsP3.jl:222-239 */
static inline FT FN(o_local_rime_density)(const TY(cmx_local_rime_density) * r, FT Ri) {
    Ri = FN(o_clamp)(Ri, (FT)1, (FT)12);
    if (Ri <= 8) return r->a + r->b * Ri + r->c * (Ri * Ri);
    FT rho8 = r->a + r->b * 8 + r->c * 64;
    FT f = (Ri - 8) / (12 - 8);
    return (1 - f) * rho8 + f * r->rho_ice;
}
/* Chen2022VelocityCurve of the rain coefficients — src/Common.jl:381-382 with Chen2022_vel_coeffs(::Rain) :290-302 */
static inline FT FN(o_chen_rain_particle_velocity)(const FT a[3], const FT b[3], const FT c[3], FT D) {
    return a[0] * M_POW(D, b[0]) * M_EXP(-c[0] * D) + a[1] * M_POW(D, b[1]) * M_EXP(-c[1] * D) + a[2] * M_POW(D, b[2]) * M_EXP(-c[2] * D);
}
/* gamma_inc_moment — src/P3_size_distribution.jl:121-133 */
static inline FT FN(o_gamma_inc_moment)(FT D1, FT D2, FT p, FT alpha, int gi_iters) {
    if (!(D2 > D1)) return 0;
    if (!(alpha > 0)) return (FT)NAN;
    FT z = p + 1, x1 = alpha * D1, x2 = alpha * D2, p1, q1, p2, q2;
    FN(o_gamma_inc)(z, x1, gi_iters, &p1, &q1);
    FN(o_gamma_inc)(z, x2, gi_iters, &p2, &q2);
    FT dq = x2 < z + 1 ? p2 - p1 : q1 - q2;
    dq = FN(o_max)(dq, (FT)0);
    return M_TGAMMA(z) * dq / M_POW(alpha, z);
}
/* test switch: the crossover solve's iteration count (0 = the reference's 8 / 10) — how far the budget-limited crossover diameter is from the root, and what
 * that does to the collision rates (tests/test_p3_collisions_oracle.py) */
static int FN(g_crossover_iters) = 0;
void FN(cmxo_set_crossover_iters)(int32_t n) { FN(g_crossover_iters) = n; }
/* crossover_diameter — src/P3_processes.jl:325-334: root of v_l(D) − v_target on [D_min, D_max], Brent, fixed 8 / 10 iterations */
typedef struct TY(cmxo_cross_ctx) { const FT *ra, *rb, *rc; FT v_target; } TY(cmxo_cross_ctx);
static FT FN(o_cross_problem)(FT D, const void *ctx) {
    const TY(cmxo_cross_ctx) *k = (const TY(cmxo_cross_ctx) *)ctx;
    return FN(o_chen_rain_particle_velocity)(k->ra, k->rb, k->rc, D) - k->v_target;
}
static inline FT FN(o_crossover_diameter)(FT v_target, const FT ra[3], const FT rb[3], const FT rc[3], FT D_min, FT D_max, int maxiters) {
    TY(cmxo_cross_ctx) k = {ra, rb, rc, v_target};
    FT a = D_min, b = D_max, fa = FN(o_cross_problem)(a, &k), fb = FN(o_cross_problem)(b, &k);
    if (!isfinite(fa) || !isfinite(fb) || fa * fb > 0) return M_ABS(fa) <= M_ABS(fb) ? a : b;
    return FN(o_brent_fixed)(FN(o_cross_problem), &k, a, b, fa, fb, maxiters);
}

/* everything ∫liquid_ice_collisions (src/P3_processes.jl:527-562) sets up once per point */
typedef struct TY(cmxo_p3col) {
    const TY(cmx_p3_ice_params) * ip;
    TY(cmxo_p3_state) s;
    TY(cmxo_p3_vterm) vt;             /* ice particle velocity */
    FT ra[3], rb[3], rc[3];           /* rain Chen-2022 curve at ρₐ */
    FT mu, lam, logN0;                /* ice PSD */
    FT ice_bnd[5];
    FT logN0c, lam_c, nu_cD, mu_cD;   /* cloud PSD in diameter — pdf_cloud_parameters CM2:227-236 */
    FT c_lo, c_hi;                    /* cloud bounds */
    FT N0r, Dr_mean, r_lo, r_hi;      /* rain PSD and bounds */
    FT T_C;                           /* T − T_freeze of compute_local_rime_density :281 */
    FT mfr_amp;                       /* Σ|operands| / |result| of mfr_coef = K (T_frz − T) + L_v D ρ (q_sat,i(T_frz) − q_sat,i(T)): both differences cancel as T → T_frz */
    FT dT_amp;                        /* (|T| + T_frz) / |T − T_frz| */
    FT mfr_coef, denom; int above_freezing;   /* compute_max_freeze_rate :167-201 */
    FT cbrt_Nsc, nu_air;
    FT rho_w;
    int gi_iters, brent_iters;
} TY(cmxo_p3col);

/* log_pdf_cloud_parameters_mass CM2:172-188 + pdf_cloud_parameters :227-236 */
static inline void FN(o_pdf_cloud_parameters)(const TY(cmx_cloud_pdf_sb2006) * pdf, FT q, FT rho, FT N, const TY(cmxo_thresholds) * th,
                                             FT *logN0c, FT *lam_c, FT *nu_cD, FT *mu_cD) {
    FT safe_q = FN(o_max)(q, th->eps_m), safe_N = FN(o_max)(N, th->eps_n);
    FT L = rho * safe_q;
    FT logx = M_LOG(L / safe_N);
    FT z1 = (pdf->nu_c + 1) / pdf->mu_c;
    FT logB = -pdf->mu_c * (logx + pdf->loggamma_z1 - pdf->loggamma_z2);
    FT logA = M_LOG(pdf->mu_c) + M_LOG(safe_N) + z1 * logB - pdf->loggamma_z1;
    if (N < th->eps_n || q < th->eps_m) { logA = -(FT)INFINITY; logB = (FT)INFINITY; }
    FT k_m = pdf->rho_w * (FT)M_PI / 6;
    *logN0c = logA + M_LOG((FT)3) + (pdf->nu_c + 1) * M_LOG(k_m);
    *lam_c = M_EXP(logB) * M_POW(k_m, pdf->mu_c);
    *nu_cD = 3 * pdf->nu_c + 2;
    *mu_cD = 3 * pdf->mu_c;
}
static inline FT FN(o_cloud_psd)(const TY(cmxo_p3col) * k, FT D) {   /* DT.size_distribution(::CloudParticlePDF_SB2006) CM2:295-303 */
    if (k->logN0c == -(FT)INFINITY) return 0;
    return M_EXP(k->logN0c + k->nu_cD * M_LOG(D) - k->lam_c * M_POW(D, k->mu_cD));
}
static inline FT FN(o_rain_psd)(const TY(cmxo_p3col) * k, FT D) {    /* DT.size_distribution(::RainParticlePDF_SB2006) CM2:270-277 */
    if (k->N0r == 0) return 0;
    return k->N0r * M_EXP(-D / k->Dr_mean);
}
/* compute_max_freeze_rate(…)(Dᵢ) — src/P3_processes.jl:167-201 */
static inline FT FN(o_max_freeze_rate)(const TY(cmxo_p3col) * k, FT Di, FT v_i) {
    if (k->above_freezing) return 0;
    if (!(k->denom > 0)) return sizeof(FT) == 4 ? (FT)FLT_MAX : (FT)DBL_MAX;
    FT Fv = k->ip->vent.a + k->ip->vent.b * k->cbrt_Nsc * M_SQRT(Di * v_i / k->nu_air);
    return 2 * ((FT)M_PI * Di) * Fv * k->mfr_coef / k->denom;
}
static inline void FN(o_p3col_setup)(TY(cmxo_p3col) * k, const TY(cmx_p3_ice_params) * ip, const TY(cmx_air_properties) * aps,
                                    const TY(cmx_thermo) * tps, uint32_t flags, const TY(cmxo_thresholds) * th,
                                    const TY(cmxo_p3_state) * s, FT L_c, FT N_c, FT L_r, FT N_r, FT rho_a, FT T, FT loglam) {
    const TY(cmx_p3_params) *pr = &ip->scheme;
    k->ip = ip; k->s = *s;
    k->gi_iters = sizeof(FT) == 4 ? 20 : 30;
    k->brent_iters = FN(g_crossover_iters) > 0 ? FN(g_crossover_iters) : (s->eps > (FT)1e-10 ? 8 : 10);
    FN(o_chen_small_ice)(&ip->vel_ice.small_ice, rho_a, (FT)916.7, k->vt.as, k->vt.bs, k->vt.cs);
    FN(o_chen_large_ice)(&ip->vel_ice.large_ice, rho_a, (FT)916.7, k->vt.al, k->vt.bl, k->vt.cl);
    k->vt.cutoff = ip->vel_ice.small_ice.cutoff;
    k->vt.aspect = !(flags & CMX_P3_NO_ASPECT_RATIO);
    FN(o_chen2022_rain_coeffs)(&ip->vel_rain, rho_a, k->ra, k->rb, k->rc);
    k->mu = FN(o_p3_mu)(pr, flags, loglam); k->lam = M_EXP(loglam);
    k->logN0 = M_LOG(s->rho_n_ice) - FN(o_loggamma_moment)(k->mu, loglam, (FT)0);
    /* p = FT(0.00001) — :546; integral_bounds src/P3_integral_properties.jl:34-46 */
    FT Y1 = (FT)0.00001, Y2 = 1 - Y1;
    if (s->eps > (FT)1e-10) { Y1 = (FT)(float)0.00001f; Y2 = (FT)(1.0f - (float)Y1); }
    FT Q1 = 1 - Y1, Q2 = 1 - Y2;
    if (s->eps > (FT)1e-10) { Q1 = (FT)(1.0f - (float)Y1); Q2 = (FT)(1.0f - (float)Y2); }
    FT D_min = FN(o_gamma_inc_inv)(k->mu + 1, Y1, Q1, k->gi_iters, s->eps) / k->lam;
    FT D_max = FN(o_gamma_inc_inv)(k->mu + 1, Y2, Q2, k->gi_iters, s->eps) / k->lam;
    k->ice_bnd[0] = D_min; k->ice_bnd[1] = FN(o_clamp)(s->D_th, D_min, D_max); k->ice_bnd[2] = FN(o_clamp)(s->D_gr, D_min, D_max);
    k->ice_bnd[3] = FN(o_clamp)(s->D_cr, D_min, D_max); k->ice_bnd[4] = D_max;
    /* cloud: get_size_distribution_bounds CM2:347-355 → generalized_gamma_quantile DistributionTools.jl:44-48 */
    FN(o_pdf_cloud_parameters)(&ip->cloud_pdf, L_c / rho_a, rho_a, N_c, th, &k->logN0c, &k->lam_c, &k->nu_cD, &k->mu_cD);
    FT zq = (k->nu_cD + 1) / k->mu_cD;
    k->c_lo = M_POW(FN(o_gamma_inc_inv)(zq, Y1, Q1, k->gi_iters, s->eps) / k->lam_c, 1 / k->mu_cD);
    k->c_hi = M_POW(FN(o_gamma_inc_inv)(zq, Y2, Q2, k->gi_iters, s->eps) / k->lam_c, 1 / k->mu_cD);
    /* rain: pdf_rain_parameters + exponential_quantile DistributionTools.jl:158-165 (cloglog(Y) = log(−log1p(−Y))) */
    TY(cmxo_rain_pdf) rp = FN(o_pdf_rain_parameters)(&ip->rain_pdf, (flags & CMX_P3_RAIN_PDF_LIMITED) != 0, L_r / rho_a, rho_a, N_r, th);
    k->N0r = rp.N0r; k->Dr_mean = rp.Dr_mean;
    if (rp.Dr_mean == 0) { k->r_lo = 0; k->r_hi = 0; }
    else {
        k->r_lo = M_EXP(M_LOG(rp.Dr_mean) + M_LOG(-M_LOG1P(-Y1)));
        k->r_hi = M_EXP(M_LOG(rp.Dr_mean) + M_LOG(-M_LOG1P(-Y2)));
    }
    k->T_C = T - pr->T_freeze;
    /* compute_max_freeze_rate :167-201 */
    FT T_frz = tps->T_freeze;
    FT L_v = FN(o_latent_heat_vapor)(tps, T), L_f = FN(o_latent_heat_fusion)(tps, T);
    FT dT = T_frz - T;
    FT drho_v = rho_a * (FN(o_qsat_ice)(tps, T_frz, rho_a) - FN(o_qsat_ice)(tps, T, rho_a));
    k->denom = L_f - tps->cp_l * dT;
    k->mfr_coef = aps->K_therm * dT + L_v * aps->D_vapor * drho_v;
    {   /* conditioning of the two T-differences (parity scale only: not part of the reference's arithmetic) */
        FT operands = aps->K_therm * (T_frz + M_ABS(T)) + L_v * aps->D_vapor * rho_a * (FN(o_qsat_ice)(tps, T_frz, rho_a) + FN(o_qsat_ice)(tps, T, rho_a));
        k->mfr_amp = k->mfr_coef != 0 ? operands / M_ABS(k->mfr_coef) : (FT)0;
        k->dT_amp = T != T_frz ? (M_ABS(T) + T_frz) / M_ABS(T - T_frz) : (FT)0;
    }
    k->above_freezing = T >= T_frz;
    k->cbrt_Nsc = M_CBRT(aps->nu_air / aps->D_vapor); k->nu_air = aps->nu_air;
    k->rho_w = ip->cloud_pdf.rho_w;
}
/* get_liquid_integrals(…)(Dᵢ) — :307-322: (∂ₜN_col, ∂ₜM_col, ∂ₜB_col) over one liquid species; which = 0 cloud, 1 rain, 2 rain
 * rime-volume integral only (the closed form supplies N and M) */
static inline void FN(o_liquid_integrals)(const TY(cmxo_p3col) * k, const TY(cmx_quadrature) * quad, int which, FT v_i, const FT K[3],
                                         FT *N, FT *M, FT *B) {
    FT a = which == 0 ? k->c_lo : k->r_lo, b = which == 0 ? k->c_hi : k->r_hi;
    *N = 0; *M = 0; *B = 0;
    if (!(a < b)) return;
    FT scale = (b - a) / 2, shift = (a + b) / 2, r1 = 0, r2 = 0, r3 = 0;
    for (int j = 0; j < Q_N(quad); ++j) {
        FT D = scale * Q_NODE(quad, j) + shift, w = Q_WEIGHT(quad, j);
        FT dv = M_ABS(v_i - FN(o_chen_rain_particle_velocity)(k->ra, k->rb, k->rc, D));
        FT Kc = M_FMA(D, M_FMA(D, K[2], K[1]), K[0]);                                  /* evalpoly :123-124 */
        FT V = (FT)1 * Kc * dv;                                                       /* E = 1 :142-144 */
        FT nD = which == 0 ? FN(o_cloud_psd)(k, D) : FN(o_rain_psd)(k, D);
        FT m = k->rho_w * (D * D * D * (FT)M_PI / 6);                                  /* m_liq :612, volume_sphere_D Common.jl:475 */
        FT Ri = (D * 1000000 * dv) / (2 * k->T_C);                                     /* :287-288 */
        FT t1 = V * nD, t2 = t1 * m, t3 = t2 / FN(o_local_rime_density)(&k->ip->rho_rim_local, Ri);
        r1 += t1 * w; r2 += t2 * w; r3 += t3 * w;
    }
    *N = scale * r1; *M = scale * r2; *B = scale * r3;
}
/* closed_rain_inner_NM — :343-369 */
static inline void FN(o_closed_rain_inner_NM)(const TY(cmxo_p3col) * k, FT v_i, const FT K[3], FT *N, FT *M) {
    FT lam = 1 / k->Dr_mean;
    FT Dstar = FN(o_crossover_diameter)(v_i, k->ra, k->rb, k->rc, k->r_lo, k->r_hi, k->brent_iters);
    FT cross[2];
    for (int ip = 0; ip < 2; ++ip) {
        FT p = ip == 0 ? (FT)0 : (FT)3, fl[2];
        for (int h = 0; h < 2; ++h) {
            FT a = h == 0 ? k->r_lo : Dstar, b = h == 0 ? Dstar : k->r_hi;
#define IP(pp, al) (K[0] * FN(o_gamma_inc_moment)(a, b, (pp), (al), k->gi_iters) + K[1] * FN(o_gamma_inc_moment)(a, b, (pp) + 1, (al), k->gi_iters) + \
                    K[2] * FN(o_gamma_inc_moment)(a, b, (pp) + 2, (al), k->gi_iters))
            FT s = v_i * IP(p, lam);
            for (int j = 0; j < 3; ++j) s -= k->ra[j] * IP(p + k->rb[j], lam + k->rc[j]);
#undef IP
            fl[h] = s;
        }
        cross[ip] = fl[0] - fl[1];
    }
    FT mfac = k->rho_w * ((FT)M_PI / 6);
    *N = k->N0r * cross[0];
    *M = k->N0r * mfac * cross[1];
}
/* (cond[2], optional, parity scale only: ∫ n M_frz and ∫ n (B_c + B_r) f_frz over the nodes where the maximum freezing rate is the
 * active limit — the part of the freeze / shed split that is proportional to mfr_coef) */
/* ∫liquid_ice_collisions — :449-489 (outer integrand) + :527-562; rates[10] = (QCFRZ, QCSHD, NCCOL, QRFRZ, QRSHD, NRCOL, ∫M_col,
 * BCCOL, BRCOL, ∫𝟙_wet M_col) */
static inline void FN(o_p3_collision_integrals)(const TY(cmxo_p3col) * k, const TY(cmx_quadrature) * quad, FT rates[10], FT cond[2]) {
    const TY(cmx_p3_params) *pr = &k->ip->scheme;
    for (int q = 0; q < 10; ++q) rates[q] = 0;
    if (cond) cond[0] = cond[1] = 0;
    for (int sg = 0; sg < 4; ++sg) {
        FT a = k->ice_bnd[sg], b = k->ice_bnd[sg + 1];
        if (!(a < b)) continue;
        FT scale = (b - a) / 2, shift = (a + b) / 2, r[10] = {0};
        for (int i = 0; i < Q_N(quad); ++i) {
            FT Di = scale * Q_NODE(quad, i) + shift, w = Q_WEIGHT(quad, i);
            FT v_i = FN(o_p3_particle_velocity)(pr, &k->s, &k->vt, Di);
            FT ri = M_SQRT(FN(o_p3_ice_area)(pr, &k->s, Di) / (FT)M_PI);
            FT K[3] = {(FT)M_PI * (ri * ri), (FT)M_PI * ri, (FT)(M_PI / 4)};
            FT Nc, Mc, Bc, Nr = 0, Mr = 0, Br = 0;
            FN(o_liquid_integrals)(k, quad, 0, v_i, K, &Nc, &Mc, &Bc);
            /* get_liquid_integrals_rain_closed — :381-418 */
            if (!(k->N0r == 0 || !(k->r_hi > k->r_lo))) {
                FN(o_closed_rain_inner_NM)(k, v_i, K, &Nr, &Mr);
                if (!(isfinite(Nr) && isfinite(Mr))) { Nr = 0; Mr = 0; Br = 0; }
                else { FT n_, m_; FN(o_liquid_integrals)(k, quad, 1, v_i, K, &n_, &m_, &Br); }
            }
            FT M_col = Mc + Mr;
            FT M_frz = FN(o_min)(M_col, FN(o_max_freeze_rate)(k, Di, v_i));
            FT f_frz = M_col == 0 ? (FT)0 : M_frz / M_col;
            FT wet = M_col > M_frz ? (FT)1 : (FT)0;
            FT n = M_EXP(k->logN0 + k->mu * M_LOG(Di) - k->lam * Di);
            r[0] += n * Mc * f_frz * w;       r[1] += n * Mc * (1 - f_frz) * w;   r[2] += n * Nc * w;
            r[3] += n * Mr * f_frz * w;       r[4] += n * Mr * (1 - f_frz) * w;   r[5] += n * Nr * w;
            r[6] += n * M_col * w;            r[7] += n * Bc * f_frz * w;         r[8] += n * Br * f_frz * w;
            r[9] += n * wet * M_col * w;
            if (cond) { cond[0] += scale * n * wet * M_frz * w; cond[1] += scale * n * wet * (Bc + Br) * f_frz * w; }
        }
        for (int q = 0; q < 10; ++q) rates[q] += scale * r[q];
    }
}
/* bulk_liquid_ice_collision_sources — :600-655: src[7] = (∂ₜq_c, ∂ₜq_r, ∂ₜN_c, ∂ₜN_r, ∂ₜL_rim, ∂ₜL_ice, ∂ₜB_rim) */
static inline void FN(o_p3_collision_sources)(const TY(cmxo_p3col) * k, const FT rates[10], FT rho_a, FT src[7]) {
    const TY(cmx_p3_params) *pr = &k->ip->scheme;
    FT QCFRZ = rates[0], QCSHD = rates[1], NCCOL = rates[2], QRFRZ = rates[3], QRSHD = rates[4], NRCOL = rates[5], M_col = rates[6],
       BCCOL = rates[7], BRCOL = rates[8], wetM = rates[9];
    FT f_wet = M_col == 0 ? (FT)0 : wetM / M_col;
    FT D_shd = (FT)1e-3;
    FT NRSHD = QRSHD / (k->rho_w * (D_shd * D_shd * D_shd * (FT)M_PI / 6));
    FT B_rim = k->s.rho_rim == 0 ? (FT)0 : (k->s.rho_q_ice * k->s.F_rim) / k->s.rho_rim;
    FT QIWET = f_wet * k->s.rho_q_ice * (1 - k->s.F_rim) / pr->tau_wet;
    FT BIWET = f_wet * (k->s.rho_q_ice / pr->rho_i - B_rim) / pr->tau_wet;
    src[0] = (-QCFRZ - QCSHD) / rho_a;
    src[1] = (-QRFRZ + QCSHD) / rho_a;
    src[2] = -NCCOL;
    src[3] = -NRCOL + NRSHD;
    src[4] = QCFRZ + QRFRZ + QIWET;
    src[5] = QCFRZ + QRFRZ;
    src[6] = BCCOL + BRCOL + BIWET;
}
/* oracle twin of cmx_p3_liquid_ice_collisions_*: sources[7][n] and/or rates[10][n] (either may be NULL) */
void FN(cmxo_p3_liquid_ice_collisions)(const TY(cmx_p3_ice_params) * ip, const TY(cmx_air_properties) * aps, const TY(cmx_thermo) * tps,
                                      const TY(cmx_quadrature) * quad, uint32_t flags, const TY(cmxo_thresholds) * th, int64_t n,
                                      const FT *rho_q_ice, const FT *rho_n_ice, const FT *x3, const FT *x4, const FT *L_c, const FT *N_c,
                                      const FT *L_r, const FT *N_r, const FT *rho_a, const FT *T, const FT *loglam, FT *sources, FT *rates,
                                      int32_t nthreads) {
    (void)nthreads;
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads > 0 ? nthreads : 1)
    for (int64_t i = 0; i < n; ++i) {
        const TY(cmx_p3_params) *pr = &ip->scheme;
        TY(cmxo_p3_state) s = (flags & CMX_P3_INPUT_IS_STATE) ? FN(o_p3_state)(pr, rho_q_ice[i], rho_n_ice[i], x3[i], x4[i], th->eps_ft)
                                                              : FN(o_p3_state_from_prognostic)(pr, rho_q_ice[i], rho_n_ice[i], x3[i], x4[i], th->eps_ft);
        FT r[10] = {0}, src[7] = {0};
        if (!(s.rho_n_ice < s.eps || s.rho_q_ice < s.eps)) {
            TY(cmxo_p3col) k;
            FN(o_p3col_setup)(&k, ip, aps, tps, flags, th, &s, L_c[i], N_c[i], L_r[i], N_r[i], rho_a[i], T[i], loglam[i]);
            FN(o_p3_collision_integrals)(&k, quad, r, (FT *)0);
            FN(o_p3_collision_sources)(&k, r, rho_a[i], src);
        }
        if (rates) for (int q = 0; q < 10; ++q) rates[(int64_t)q * n + i] = r[q];
        if (sources) for (int q = 0; q < 7; ++q) sources[(int64_t)q * n + i] = src[q];
    }
}
/* probes for the KATs of test/p3_tests.jl:695-726: compute_max_freeze_rate(aps, tps, vel, ρₐ, Tₐ, state)(D) and
 * compute_local_rime_density(vel, ρₐ, T, state)(Dᵢ, Dₗ) for a state built from (L, N, F_rim, ρ_rim) */
void FN(cmxo_p3_collision_probes)(const TY(cmx_p3_ice_params) * ip, const TY(cmx_air_properties) * aps, const TY(cmx_thermo) * tps,
                                 uint32_t flags, FT L, FT N, FT F_rim, FT rho_rim, FT rho_a, FT T, FT loglam, FT Di, FT Dl, FT out[2]) {
    TY(cmxo_thresholds) th = {M_EPS, M_EPS, (FT)0, M_EPS};
    TY(cmxo_p3_state) s = FN(o_p3_state)(&ip->scheme, L, N, F_rim, rho_rim, M_EPS);
    TY(cmxo_p3col) k;
    FN(o_p3col_setup)(&k, ip, aps, tps, flags, &th, &s, (FT)0, (FT)0, (FT)0, (FT)0, rho_a, T, loglam);
    FT v_i = FN(o_p3_particle_velocity)(&ip->scheme, &s, &k.vt, Di);
    out[0] = FN(o_max_freeze_rate)(&k, Di, v_i);
    FT dv = M_ABS(v_i - FN(o_chen_rain_particle_velocity)(k.ra, k.rb, k.rc, Dl));
    out[1] = FN(o_local_rime_density)(&ip->rho_rim_local, (Dl * 1000000 * dv) / (2 * k.T_C));
}

/* ---- IceNucleation.jl: the rates of the 2M+P3 entry ------------------------------------------------------------------ */
/* P3_deposition_N_i — src/IceNucleation.jl:165-169;  P3_het_N_i — :204-207 */
static inline FT FN(o_P3_deposition_N_i)(const TY(cmx_morrison_milbrandt2014) * ip, FT T) {
    FT Tp = FN(o_max)(ip->T_dep_thres, T);
    FT Ni = 1000 * ip->c1 * M_EXP(ip->c2 * (ip->T0 - Tp));
    return T < ip->T0 ? Ni : (FT)0;
}
static inline FT FN(o_P3_het_N_i)(const TY(cmx_morrison_milbrandt2014) * ip, FT T, FT N_l, FT V_l, FT dt) {
    FT Ts = ip->T0 - T;
    return N_l * (1 - M_EXP(-ip->het_B * V_l * dt * M_EXP(ip->het_a * Ts)));
}
/* INP_concentration_mean — :250-253: restated in cmx_oracle_1m_impl.h (first user);  INP_concentration_frequency — :221-226 */
static inline FT FN(o_INP_concentration_frequency)(const TY(cmx_frostenberg2023) * ip, FT INPC, FT T) {
    if (T >= ip->T_freeze) return 0;
    FT mu = FN(o_INP_concentration_mean)(ip, T);
    FT d = M_LOG(INPC) - mu, two_s2 = 2 * (ip->sigma * ip->sigma);
    return M_EXP(-(d * d) / two_s2) / M_SQRT((FT)M_PI * two_s2);
}
/* liquid_freezing_rate(::RainFreezing, pdf_r, …) — :274-311 (rain, exponential PSD): (∂ₜn_frz, ∂ₜq_frz) */
static inline void FN(o_liquid_freezing_rate_rain)(const TY(cmx_rain_freezing) * rf, const TY(cmx_rain_pdf_sb2006) * pdf, int limited,
                                                  const TY(cmx_thermo) * tps, const TY(cmxo_thresholds) * th, FT q, FT rho, FT N, FT T,
                                                  FT *dn, FT *dq) {
    FT T_freeze = tps->T_freeze, n = N / rho;
    TY(cmxo_rain_pdf) rp = FN(o_pdf_rain_parameters)(pdf, limited, q, rho, N, th);
    FT J = rf->het_B * M_EXP(rf->het_a * (T_freeze - T));
    FT M3 = n * 6 * M_POW(rp.Dr_mean, (FT)3), M6 = n * 720 * M_POW(rp.Dr_mean, (FT)6);      /* exponential_Mⁿ DistributionTools.jl:189-191 */
    FT V1 = (FT)M_PI / 6;
    FT rn = J * V1 * M3, rq = J * pdf->rho_w * (V1 * V1) * M6;
    int cond = (n > th->eps_n) && (q > th->eps_m) && (T < T_freeze - 4);
    *dn = cond ? rn : (FT)0; *dq = cond ? rq : (FT)0;
}
/* liquid_freezing_rate(::RainFreezing, ::CloudParticlePDF_SB2006, …) — :355-389 (cloud, generalized gamma PSD) */
static inline void FN(o_liquid_freezing_rate_cloud)(const TY(cmx_rain_freezing) * rf, const TY(cmx_cloud_pdf_sb2006) * pdf,
                                                   const TY(cmx_thermo) * tps, const TY(cmxo_thresholds) * th, FT q, FT rho, FT N, FT T,
                                                   FT *dn, FT *dq) {
    FT T_freeze = tps->T_freeze, n = N / rho, logN0c, lam_c, nu, mu;
    FN(o_pdf_cloud_parameters)(pdf, q, rho, N, th, &logN0c, &lam_c, &nu, &mu);
    FT J = rf->het_B * M_EXP(rf->het_a * (T_freeze - T));
    /* generalized_gamma_Mⁿ DistributionTools.jl:109-112 */
    FT M3 = n * M_POW(lam_c, -3 / mu) * M_TGAMMA((nu + 1 + 3) / mu) / M_TGAMMA((nu + 1) / mu);
    FT M6 = n * M_POW(lam_c, -6 / mu) * M_TGAMMA((nu + 1 + 6) / mu) / M_TGAMMA((nu + 1) / mu);
    FT V1 = (FT)M_PI / 6;
    FT rn = J * V1 * M3, rq = J * pdf->rho_w * (V1 * V1) * M6;
    int cond = (n > th->eps_n) && (q > th->eps_m) && (T < T_freeze - 4);
    *dn = cond ? rn : (FT)0; *dq = cond ? rq : (FT)0;
}
/* immersion_limit_rate — :425-435 */
static inline FT FN(o_immersion_limit_rate)(const TY(cmx_frostenberg2023) * ip, FT T, FT rho, FT tau, FT inpc_log_shift, FT n_active) {
    if (T >= ip->T_freeze) return 0;
    FT log_inpc = FN(o_INP_concentration_mean)(ip, T) + inpc_log_shift;
    FT per_kg = M_EXP(log_inpc) / rho;
    return FN(o_max)((FT)0, per_kg - n_active) / tau;
}
/* deposition_rate — :491-511 (T_thresh = T_freeze − 15, S_i_thresh = 0.05 defaults) */
static inline void FN(o_deposition_rate)(const TY(cmx_frostenberg2023) * ip, const TY(cmx_thermo) * tps, FT T, FT rho, FT q_tot, FT q_liq,
                                        FT q_ice, FT n_ice, FT m_nuc, FT tau_act, FT inpc_log_shift, FT *dn, FT *dq) {
    FT T_thresh = ip->T_freeze - 15, S_thresh = (FT)0.05;
    FT q_sat_ice = FN(o_qsat_ice)(tps, T, rho);
    FT q_vap = FN(o_q_vap)(q_tot, q_liq, q_ice);
    FT S_i = q_vap / q_sat_ice - 1;
    int cond = (T < T_thresh) && (S_i > S_thresh);
    FT log_inpc = FN(o_INP_concentration_mean)(ip, T) + inpc_log_shift;
    FT per_kg = M_EXP(log_inpc) / rho;
    FT rn = FN(o_max)((FT)0, per_kg - n_ice) / tau_act;
    rn = cond ? rn : (FT)0;
    FT q_excess = FN(o_max)((FT)0, q_vap - q_sat_ice);
    *dn = rn;
    *dq = FN(o_min)(m_nuc * rn, q_excess / (2 * tau_act));
}

/* ---- bulk_microphysics_tendencies(::Microphysics2Moment, mp{WR, P3IceParams}, …) — BMT:898-1083 ------------------------------- */
/* out[8] = (dq_lcl, dn_lcl, dq_rai, dn_rai, dq_ice, dn_ice, dq_rim, db_rim); scale[8] = Σ|terms| of each output (parity scale) */
static inline void FN(o_bulk_tendencies_2m_p3)(const TY(cmx_warm_rain_2m) * wr, const TY(cmx_p3_ice_params) * ip, const TY(cmx_thermo) * tps,
                                              uint32_t flags, const TY(cmxo_thresholds) * th, FT rho, FT T, FT q_tot, FT q_lcl, FT n_lcl,
                                              FT q_rai, FT n_rai, FT q_ice, FT n_ice, FT q_rim, FT b_rim, FT loglam, FT inpc_log_shift,
                                              FT out[8], FT scale[8]) {
    const TY(cmx_p3_params) *pr = &ip->scheme;
    const TY(cmx_air_properties) *aps = &wr->air_properties;
    const int limited = (flags & CMX_P3_RAIN_PDF_LIMITED) != 0;
    rho = FN(o_max)((FT)0, rho); q_tot = FN(o_max)((FT)0, q_tot); q_lcl = FN(o_max)((FT)0, q_lcl); q_rai = FN(o_max)((FT)0, q_rai);
    n_lcl = FN(o_max)((FT)0, n_lcl); n_rai = FN(o_max)((FT)0, n_rai); q_ice = FN(o_max)((FT)0, q_ice); n_ice = FN(o_max)((FT)0, n_ice);
    q_rim = FN(o_max)((FT)0, q_rim); b_rim = FN(o_max)((FT)0, b_rim);
    FT L_lcl = q_lcl * rho, L_rai = q_rai * rho, N_lcl = n_lcl * rho, N_rai = n_rai * rho;
    FT L_ice = q_ice * rho, N_ice = n_ice * rho, L_rim = q_rim * rho, B_rim = b_rim * rho;
    TY(cmxo_p3_state) s = FN(o_p3_state_from_prognostic)(pr, L_ice, N_ice, L_rim, B_rim, th->eps_ft);
    /* warm rain — BMT:942 (the warm flags: limited PSD, no velocities) */
    TY(cmxo_warm_rain_out) w = FN(o_bulk_tendencies_2m_warm)(wr, tps, (const TY(cmx_rain_vel) *)0, limited ? CMX_SB2006_LIMITED : 0u, th,
                                                            (FT)0, rho, T, q_tot, q_lcl, n_lcl, q_rai, n_rai, q_ice);
    FT dq_lcl = w.dq_lcl_dt, dn_lcl = w.dn_lcl_dt, dq_rai = w.dq_rai_dt, dn_rai = w.dn_rai_dt;
    FT dq_ice = 0, dn_ice = 0, dq_rim = 0, db_rim = 0;
    FT sc[8] = {w.scale[0], w.scale[1], w.scale[2], w.scale[3], 0, 0, 0, 0};
    uint32_t p3flags = flags & (CMX_P3_SLOPE_CONSTANT | CMX_P3_NO_ASPECT_RATIO | CMX_P3_RAIN_PDF_LIMITED);
    const int gi_iters = sizeof(FT) == 4 ? 20 : 30;
    /* BMT:959; the P3 integrals additionally need ρq, ρn ≥ eps (the gate of every cmx_p3_* entry: below it the size distribution
     * is not defined and get_distribution_logλ returns −Inf) */
    if (q_ice > th->eps_m && n_ice > th->eps_n && !(s.rho_n_ice < s.eps || s.rho_q_ice < s.eps)) {
        TY(cmxo_p3col) k;
        FT r[10], src[7];
        FN(o_p3col_setup)(&k, ip, aps, tps, p3flags, th, &s, L_lcl, N_lcl, L_rai, N_rai, rho, T, loglam);
        FT cond[2];
        FN(o_p3_collision_integrals)(&k, &ip->quad, r, cond);
        FN(o_p3_collision_sources)(&k, r, rho, src);
        dq_lcl += src[0]; dq_rai += src[1]; dn_lcl += src[2] / rho; dn_rai += src[3] / rho;
        dq_ice += src[5] / rho; dq_rim += src[4] / rho; db_rim += src[6] / rho;
        sc[0] += M_ABS(src[0]); sc[2] += (r[3] + r[1]) / rho; sc[1] += M_ABS(src[2] / rho); sc[3] += (r[5] + M_ABS(src[3] + r[5])) / rho;
        sc[4] += M_ABS(src[5] / rho); sc[6] += M_ABS(src[4] / rho); sc[7] += (M_ABS(r[7]) + M_ABS(r[8]) + M_ABS(src[6] - r[7] - r[8])) / rho;
        {   /* The freeze / shed split of the collected liquid is limited by compute_max_freeze_rate (P3_processes.jl:167-201), which is
             * proportional to mfr_coef = K (T_frz − T) + L_v D ρ (q_sat(T_frz) − q_sat(T)): two differences whose operands are mfr_amp
             * times larger than the result near T_frz.  Wherever that limit is active the frozen mass (∂ₜq_ice, ∂ₜq_rim, −∂ₜq_rai through
             * QRFRZ and QCSHD, ∂ₜn_rai through the shed drops) carries that conditioning; the rime volume also that of T − T_frz in the
             * local rime density (compute_local_rime_density :281-299). */
            const FT D_shd = (FT)1e-3;
            const FT S_frz = k.mfr_amp * cond[0] / rho, S_B = (k.mfr_amp + k.dT_amp) * cond[1] / rho;
            sc[2] += S_frz; sc[4] += S_frz; sc[6] += S_frz;
            sc[3] += S_frz / (k.rho_w * (D_shd * D_shd * D_shd * (FT)M_PI / 6));
            sc[7] += S_B;
        }
        /* aggregation — BMT:976-977 */
        FT agg = FN(o_p3_ice_self_collection)(pr, &ip->vel_ice, &ip->quad, p3flags, &s, rho, loglam, gi_iters);
        dn_ice -= agg / rho; sc[5] += M_ABS(agg / rho);
        /* melting — BMT:980-994 (p = 1e-6 default of ice_melt) */
        FT mN = 0, mL = 0;
        if (T > tps->T_freeze)
            FN(o_p3_ice_melt)(pr, &ip->vel_ice, aps, tps, &ip->vent, &ip->quad, p3flags, &s, rho, T, loglam, (FT)1e-6, gi_iters, &mN, &mL);
        FT mq = mL / rho, mn = mN / rho;
        dq_rai += mq; dn_rai += mn; dq_ice -= mq; dn_ice -= mn;
        dq_rim -= mq * s.F_rim;
        db_rim -= s.rho_rim > 0 ? mq * s.F_rim / s.rho_rim : (FT)0;
        {   /* ice_melt ∝ K (T − T_frz) (P3_processes.jl:64-94): operands (|T| + T_frz)/|T − T_frz| times the result */
            const FT am = T > tps->T_freeze ? k.dT_amp : (FT)1;
            sc[2] += am * M_ABS(mq); sc[3] += am * M_ABS(mn); sc[4] += am * M_ABS(mq); sc[5] += am * M_ABS(mn); sc[6] += am * M_ABS(mq * s.F_rim);
            sc[7] += s.rho_rim > 0 ? am * M_ABS(mq * s.F_rim / s.rho_rim) : (FT)0;
        }
    }
    /* ice nucleation (F23 + Bigg) — BMT:997-1034 */
    FT tau_act = ip->tau_act;
    FT D_nuc = (FT)10e-6;
    FT m_nuc = pr->rho_i * (D_nuc * D_nuc * D_nuc * (FT)M_PI / 6);
    FT n_active = n_ice;                                                                       /* NIceProxyDepletion — IceNucleation.jl:527 */
    FT dep_n, dep_q;
    FN(o_deposition_rate)(&ip->ice_nucleation, tps, T, rho, q_tot, q_lcl + q_rai, q_ice, n_active, m_nuc, tau_act, inpc_log_shift, &dep_n, &dep_q);
    dn_ice += dep_n; dq_ice += dep_q; sc[5] += M_ABS(dep_n); sc[4] += M_ABS(dep_q);
    FT bn, bq;
    FN(o_liquid_freezing_rate_cloud)(&ip->rain_freezing, &ip->cloud_pdf, tps, th, q_lcl, rho, N_lcl, T, &bn, &bq);
    FT cap = FN(o_immersion_limit_rate)(&ip->ice_nucleation, T, rho, tau_act, inpc_log_shift, n_active);
    FT imm_n = FN(o_min)(bn, cap);
    FT imm_q = bn > 0 ? bq * imm_n / bn : (FT)0;
    dq_lcl -= imm_q; dn_lcl -= imm_n; dq_ice += imm_q; dn_ice += imm_n; dq_rim += imm_q; db_rim += imm_q / pr->rho_i;
    sc[0] += M_ABS(imm_q); sc[1] += M_ABS(imm_n); sc[4] += M_ABS(imm_q); sc[5] += M_ABS(imm_n); sc[6] += M_ABS(imm_q); sc[7] += M_ABS(imm_q / pr->rho_i);
    /* sublimation / deposition — BMT:1037-1054 */
    FT n_per_q = q_ice > th->eps_m ? n_ice / q_ice : (FT)0;
    FT sd_scale;
    FT sd = FN(o_conv_q_vap_to_q_icl_const)(wr->subdep_tau_relax, tps, q_tot, q_lcl, q_ice, q_rai, (FT)0, rho, T, &sd_scale);
    if (T > tps->T_freeze) sd = FN(o_min)(sd, (FT)0);
    FT sd_n = sd < 0 ? n_per_q * sd : (FT)0;
    dq_ice += sd; dn_ice += sd_n;
    FT sub = FN(o_min)(sd, (FT)0);
    dq_rim += sub * s.F_rim;
    db_rim += s.rho_rim > 0 ? sub * s.F_rim / s.rho_rim : (FT)0;
    sc[4] += sd_scale; sc[5] += n_per_q * (sd < 0 ? sd_scale : 0); sc[6] += sub != 0 ? sd_scale * s.F_rim : 0;
    sc[7] += (sub != 0 && s.rho_rim > 0) ? sd_scale * s.F_rim / s.rho_rim : (FT)0;
    /* ice number adjustment — BMT:1057-1064 */
    FT na = FN(o_number_tendency_from_mass_limits)((FT)1e-12, (FT)1e-5, (FT)100, q_ice, n_ice, th);
    dn_ice += na; sc[5] += (M_ABS(na) + 2 * M_ABS(n_ice)) / 100 * (na != 0);
    /* rain heterogeneous freezing — BMT:1067-1075 */
    FT rn, rq;
    FN(o_liquid_freezing_rate_rain)(&ip->rain_freezing, &ip->rain_pdf, limited, tps, th, q_rai, rho, N_rai, T, &rn, &rq);
    dq_rai -= rq; dn_rai -= rn; dq_ice += rq; dn_ice += rn; dq_rim += rq; db_rim += rq / pr->rho_i;
    sc[2] += M_ABS(rq); sc[3] += M_ABS(rn); sc[4] += M_ABS(rq); sc[5] += M_ABS(rn); sc[6] += M_ABS(rq); sc[7] += M_ABS(rq / pr->rho_i);
    out[0] = dq_lcl; out[1] = dn_lcl; out[2] = dq_rai; out[3] = dn_rai; out[4] = dq_ice; out[5] = dn_ice; out[6] = dq_rim; out[7] = db_rim;
    if (scale) for (int q = 0; q < 8; ++q) scale[q] = sc[q];
}
/* oracle twin of cmx_microphysics_2m_p3_tendencies_*: out[8][n] (+ scale[8][n], nullable) */
void FN(cmxo_microphysics_2m_p3_tendencies)(const TY(cmx_warm_rain_2m) * wr, const TY(cmx_p3_ice_params) * ip, const TY(cmx_thermo) * tps,
                                           uint32_t flags, const TY(cmxo_thresholds) * th, int64_t n, const FT *rho, const FT *T,
                                           const FT *q_tot, const FT *q_lcl, const FT *n_lcl, const FT *q_rai, const FT *n_rai,
                                           const FT *q_ice, const FT *n_ice, const FT *q_rim, const FT *b_rim, const FT *loglam,
                                           const FT *inpc_log_shift, FT *out, FT *scale, int32_t nthreads) {
    (void)nthreads;
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads > 0 ? nthreads : 1)
    for (int64_t i = 0; i < n; ++i) {
        FT o[8], sc[8];
        FN(o_bulk_tendencies_2m_p3)(wr, ip, tps, flags, th, rho[i], T[i], q_tot[i], q_lcl[i], n_lcl[i], q_rai[i], n_rai[i], q_ice[i], n_ice[i],
                                   q_rim[i], b_rim[i], loglam[i], inpc_log_shift ? inpc_log_shift[i] : (FT)0, o, sc);
        for (int q = 0; q < 8; ++q) { out[(int64_t)q * n + i] = o[q]; if (scale) scale[(int64_t)q * n + i] = sc[q]; }
    }
}
/* pointwise probes / twins of the het-nucleation entry points */
void FN(cmxo_liquid_freezing_rate)(const TY(cmx_rain_freezing) * rf, const TY(cmx_cloud_pdf_sb2006) * pdf_c, const TY(cmx_rain_pdf_sb2006) * pdf_r,
                                  int32_t limited, const TY(cmx_thermo) * tps, const TY(cmxo_thresholds) * th, int64_t n, const FT *q, const FT *rho,
                                  const FT *N, const FT *T, FT *dn, FT *dq) {
    for (int64_t i = 0; i < n; ++i) {
        if (pdf_c) FN(o_liquid_freezing_rate_cloud)(rf, pdf_c, tps, th, q[i], rho[i], N[i], T[i], &dn[i], &dq[i]);
        else FN(o_liquid_freezing_rate_rain)(rf, pdf_r, limited, tps, th, q[i], rho[i], N[i], T[i], &dn[i], &dq[i]);
    }
}
void FN(cmxo_f23_deposition_rate)(const TY(cmx_frostenberg2023) * ip, const TY(cmx_thermo) * tps, FT m_nuc, FT tau_act, int64_t n, const FT *T,
                                 const FT *rho, const FT *q_tot, const FT *q_liq, const FT *q_ice, const FT *n_ice, const FT *shift, FT *dn, FT *dq) {
    for (int64_t i = 0; i < n; ++i)
        FN(o_deposition_rate)(ip, tps, T[i], rho[i], q_tot[i], q_liq[i], q_ice[i], n_ice[i], m_nuc, tau_act, shift ? shift[i] : (FT)0, &dn[i], &dq[i]);
}
void FN(cmxo_f23_immersion_limit_rate)(const TY(cmx_frostenberg2023) * ip, FT tau, int64_t n, const FT *T, const FT *rho, const FT *n_active,
                                      const FT *shift, FT *dn) {
    for (int64_t i = 0; i < n; ++i) dn[i] = FN(o_immersion_limit_rate)(ip, T[i], rho[i], tau, shift ? shift[i] : (FT)0, n_active ? n_active[i] : (FT)0);
}
FT FN(cmxo_INP_concentration_mean)(const TY(cmx_frostenberg2023) * ip, FT T) { return FN(o_INP_concentration_mean)(ip, T); }
FT FN(cmxo_INP_concentration_frequency)(const TY(cmx_frostenberg2023) * ip, FT INPC, FT T) { return FN(o_INP_concentration_frequency)(ip, INPC, T); }
FT FN(cmxo_P3_deposition_N_i)(const TY(cmx_morrison_milbrandt2014) * ip, FT T) { return FN(o_P3_deposition_N_i)(ip, T); }
FT FN(cmxo_P3_het_N_i)(const TY(cmx_morrison_milbrandt2014) * ip, FT T, FT N_l, FT V_l, FT dt) { return FN(o_P3_het_N_i)(ip, T, N_l, V_l, dt); }
/* P3.het_ice_nucleation — src/P3_processes.jl:20-46 (ABIFM_J: src/IceNucleation.jl:124-134) */
void FN(cmxo_p3_het_ice_nucleation)(const TY(cmx_abifm_dust) * dust, const TY(cmx_thermo) * tps, int64_t n, const FT *q_lcl, const FT *N_lcl,
                                   const FT *RH, const FT *T, const FT *rho, FT *dNdt, FT *dLdt) {
    for (int64_t i = 0; i < n; ++i) {
        FT J = FN(o_ABIFM_J)(dust, RH[i] - FN(o_a_w_ice)(tps, T[i]));
        FT A_aer = (FT)1e-10;
        FT JA = isfinite(J) ? J * A_aer : (FT)0;
        dNdt[i] = FN(o_max)((FT)0, JA * N_lcl[i]);
        dLdt[i] = FN(o_max)((FT)0, JA * q_lcl[i] * rho[i]);
    }
}
/* probe for the closed-form rain inner integral (test/p3_tests.jl:919-985): for P3State(L, N, F_rim, ρ_rim) at ρₐ and an outer diameter Dᵢ
 * out = (∂ₜN_col, ∂ₜM_col, D*, v_i(Dᵢ), r_i(Dᵢ), D_min, D_max, N₀r, D̄_r) of closed_rain_inner_NM — src/P3_processes.jl:343-369 */
void FN(cmxo_p3_closed_rain_probe)(const TY(cmx_p3_ice_params) * ip, const TY(cmx_air_properties) * aps, const TY(cmx_thermo) * tps, uint32_t flags,
                                  FT L, FT N, FT F_rim, FT rho_rim, FT rho_a, FT loglam, FT L_r, FT N_r, FT Di, FT out[9]) {
    TY(cmxo_thresholds) th = {M_EPS, M_EPS, (FT)0, M_EPS};
    TY(cmxo_p3_state) s = FN(o_p3_state)(&ip->scheme, L, N, F_rim, rho_rim, M_EPS);
    TY(cmxo_p3col) k;
    FN(o_p3col_setup)(&k, ip, aps, tps, flags, &th, &s, (FT)0, (FT)0, L_r, N_r, rho_a, (FT)270, loglam);
    FT v_i = FN(o_p3_particle_velocity)(&ip->scheme, &s, &k.vt, Di);
    FT ri = M_SQRT(FN(o_p3_ice_area)(&ip->scheme, &s, Di) / (FT)M_PI);
    FT K[3] = {(FT)M_PI * (ri * ri), (FT)M_PI * ri, (FT)(M_PI / 4)};
    FT Nc = 0, Mc = 0;
    if (!(k.N0r == 0 || !(k.r_hi > k.r_lo))) FN(o_closed_rain_inner_NM)(&k, v_i, K, &Nc, &Mc);
    out[0] = Nc; out[1] = Mc; out[2] = FN(o_crossover_diameter)(v_i, k.ra, k.rb, k.rc, k.r_lo, k.r_hi, k.brent_iters);
    out[3] = v_i; out[4] = ri; out[5] = k.r_lo; out[6] = k.r_hi; out[7] = k.N0r; out[8] = k.Dr_mean;
}
/* probe: P3.crossover_diameter for Chen2022 rain at ρₐ (src/P3_processes.jl:304-317); out = (D*, v_r(D*), v_r(D_min), v_r(D_max)) */
void FN(cmxo_p3_crossover_probe)(const TY(cmx_chen2022_rain_vel) * c, FT rho_a, FT v_target, FT D_min, FT D_max, int maxiters, FT out[4]) {
    FT a[3], b[3], cc[3];
    FN(o_chen2022_rain_coeffs)(c, rho_a, a, b, cc);
    out[0] = FN(o_crossover_diameter)(v_target, a, b, cc, D_min, D_max, maxiters);
    out[1] = FN(o_chen_rain_particle_velocity)(a, b, cc, out[0]);
    out[2] = FN(o_chen_rain_particle_velocity)(a, b, cc, D_min);
    out[3] = FN(o_chen_rain_particle_velocity)(a, b, cc, D_max);
}
