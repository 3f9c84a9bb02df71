/*
 * cmx_oracle_impl.h — type-generic body of the CPU oracle (TEST INFRASTRUCTURE).
 *
 * Included twice by cmx_oracle.c with (FT, SFX) = (double, f64) and (float, f32).
 * Every function restates, operation by operation and in the same order, one
 * scalar function of CliMA/CloudMicrophysics.jl v0.38.1; the file:line it
 * follows (relative to /root/reference) is cited on each.  Thresholds
 * (eps(FT), cbrt(floatmin(FT))) are EXPLICIT inputs (cmxo_thresholds) so that
 * "Float64 arithmetic with Float32 gates" — the fair comparison target for the
 * Float32 device kernel, SURVEY §7 H2 — can be evaluated as well.
 *
 * Arithmetic that lives in un-vendored dependencies is restated from the
 * published formulas and pinned by the reference's own known-answer tests:
 *   Thermodynamics.jl (compat "0.15.4, 1", Project.toml:40): Rankine–Kirchhoff
 *   saturation vapour pressure, q_sat, supersaturation, latent heats, cp_m —
 *   call sites src/ThermodynamicsInterface.jl:9-33,82-90,118-125.
 */

#define CAT_(a, b) a##_##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SFX)          /* cmxo_foo  -> cmxo_foo_f64 */
#define TY(name) CAT(name, SFX)          /* cmx_sb2006 -> cmx_sb2006_f64 */

/* ---- thresholds: src/Utilities.jl:318-340 -------------------------------- */
typedef struct TY(cmxo_thresholds) {
    FT eps_m;   /* ϵ_numerics_2M_M(FT) = eps(FT)            Utilities.jl:325 */
    FT eps_n;   /* ϵ_numerics_2M_N(FT) = eps(FT)            Utilities.jl:332 */
    FT eps_1m;  /* ϵ_numerics(FT) = cbrt(floatmin(FT))      Utilities.jl:318 */
    FT eps_ft;  /* eps(FT) used directly (CM2:824)                            */
} TY(cmxo_thresholds);

static inline FT FN(o_max)(FT a, FT b) { return a > b ? a : b; }   /* Julia max for non-NaN */
static inline FT FN(o_min)(FT a, FT b) { return a < b ? a : b; }
static inline FT FN(o_clamp)(FT x, FT lo, FT hi) { return x < lo ? lo : (x > hi ? hi : x); } /* Base.clamp */

/* ---- Thermodynamics.jl restatement ---------------------------------------- */
/* TD.latent_heat_vapor: L_v(T) = LH_v0 + (cp_v - cp_l)(T - T_0)   (TDI:17) */
static inline FT FN(o_latent_heat_vapor)(const TY(cmx_thermo) * p, FT T) {
    return p->LH_v0 + (p->cp_v - p->cp_l) * (T - p->T_0);
}
/* TD.latent_heat_sublim (TDI:18) */
static inline FT FN(o_latent_heat_sublim)(const TY(cmx_thermo) * p, FT T) {
    return p->LH_s0 + (p->cp_v - p->cp_i) * (T - p->T_0);
}
/* TD.saturation_vapor_pressure(tps, T, Liquid()/Ice()) (TDI:82-85):
 * p_sat = p_tr (T/T_tr)^(Δcp/R_v) exp[(LH_0 - Δcp T_0)/R_v (1/T_tr - 1/T)] */
static inline FT FN(o_psat)(const TY(cmx_thermo) * p, FT T, FT LH_0, FT dcp) {
    return p->press_triple * M_POW(T / p->T_triple, dcp / p->R_v) *
           M_EXP((LH_0 - dcp * p->T_0) / p->R_v * (1 / p->T_triple - 1 / T));
}
static inline FT FN(o_psat_liquid)(const TY(cmx_thermo) * p, FT T) {
    return FN(o_psat)(p, T, p->LH_v0, p->cp_v - p->cp_l);
}
static inline FT FN(o_psat_ice)(const TY(cmx_thermo) * p, FT T) {
    return FN(o_psat)(p, T, p->LH_s0, p->cp_v - p->cp_i);
}
/* TD.q_vap_saturation(tps, T, ρ, Liquid()) = p_sat / (ρ R_v T)  (TDI:87-90) */
static inline FT FN(o_qsat_liquid)(const TY(cmx_thermo) * p, FT T, FT rho) {
    return FN(o_psat_liquid)(p, T) / (rho * p->R_v * T);
}
static inline FT FN(o_qsat_ice)(const TY(cmx_thermo) * p, FT T, FT rho) {
    return FN(o_psat_ice)(p, T) / (rho * p->R_v * T);
}
/* TD.cp_m(tps, q_tot, q_liq, q_ice) (TDI:21) */
static inline FT FN(o_cp_m)(const TY(cmx_thermo) * p, FT q_tot, FT q_liq, FT q_ice) {
    return p->cp_d + (p->cp_v - p->cp_d) * q_tot + (p->cp_l - p->cp_v) * q_liq +
           (p->cp_i - p->cp_v) * q_ice;
}
/* TDI.q_vap(q_tot, q_liq, q_ice) = clamp_to_nonneg(q_tot - q_liq - q_ice) (TDI:60) */
static inline FT FN(o_q_vap)(FT q_tot, FT q_liq, FT q_ice) {
    return FN(o_max)((FT)0, q_tot - q_liq - q_ice);
}
/* TDI.supersaturation_over_liquid (TDI:118-121): S = q_v ρ R_v T / p_sat - 1 */
static inline FT FN(o_supersaturation_over_liquid)(const TY(cmx_thermo) * p, FT q_tot, FT q_liq,
                                                  FT q_ice, FT rho, FT T) {
    FT q_v = FN(o_q_vap)(q_tot, q_liq, q_ice);
    FT p_v = q_v * (rho * p->R_v * T);
    return p_v / FN(o_psat_liquid)(p, T) - 1;
}

/* ---- Common.jl ------------------------------------------------------------- */
/* CO.G_func_liquid — src/Common.jl:47-63 */
static inline FT FN(o_G_func_liquid)(const TY(cmx_air_properties) * aps, const TY(cmx_thermo) * tps,
                                    FT T, const TY(cmxo_thresholds) * th) {
    FT R_v = tps->R_v;
    FT L = FN(o_latent_heat_vapor)(tps, T);
    FT p_vs = FN(o_psat_liquid)(tps, T);
    FT p_vs_safe = FN(o_max)(p_vs, th->eps_1m);
    FT D_safe = FN(o_max)(aps->D_vapor, th->eps_1m);
    FT K_safe = FN(o_max)(aps->K_therm, th->eps_1m);
    return 1 / (L / K_safe / T * (L / R_v / T - 1) + R_v * T / D_safe / p_vs_safe);
}

/* ---- MicrophysicsNonEq.jl -------------------------------------------------- */
/* CMNonEq._conv_q_vap_to_q_lcl_const — src/MicrophysicsNonEq.jl:117-140
 * (dqcld_dT :74-76, gamma_helper :88-90).  *scale receives |q_v| + |q_sat|
 * over the timescale: the size of the two terms whose difference is returned. */
static inline FT FN(o_conv_q_vap_to_q_lcl_const)(FT tau, const TY(cmx_thermo) * tps, FT q_tot,
                                                FT q_lcl, FT q_icl, FT q_rai, FT q_sno, FT rho,
                                                FT T, FT *scale) {
    FT R_v = tps->R_v;
    FT L_v = FN(o_latent_heat_vapor)(tps, T);
    FT cp_air = FN(o_cp_m)(tps, q_tot, q_lcl + q_rai, q_icl + q_sno);
    FT q_v = FN(o_q_vap)(q_tot, q_lcl + q_rai, q_icl + q_sno);
    FT q_sat = FN(o_qsat_liquid)(tps, T, rho);
    FT dqsl_dT = q_sat * (L_v / (R_v * (T * T)) - 1 / T);
    FT Gamma_l = 1 + (L_v / cp_air) * dqsl_dT;
    FT sat_excess = q_v - q_sat;
    FT timescale = tau * Gamma_l;
    if (scale) *scale = (M_ABS(q_v) + M_ABS(q_sat)) / M_ABS(timescale);
    if (sat_excess < 0)
        return -FN(o_min)(-sat_excess, FN(o_max)((FT)0, q_lcl)) / timescale;
    return sat_excess / timescale;
}

/* ---- Microphysics2M.jl ----------------------------------------------------- */
typedef struct TY(cmxo_rain_pdf) { FT N0r, Dr_mean, xr_mean; } TY(cmxo_rain_pdf);

/* CM2.pdf_rain_parameters — not-limited src/Microphysics2M.jl:67-89, limited :90-110 */
static inline TY(cmxo_rain_pdf) FN(o_pdf_rain_parameters)(const TY(cmx_rain_pdf_sb2006) * pdf,
                                                         int limited, FT q, FT rho, FT N,
                                                         const TY(cmxo_thresholds) * th) {
    TY(cmxo_rain_pdf) r;
    const FT pi = (FT)M_PI;
    FT safe_q = FN(o_max)(q, th->eps_m);
    FT safe_N = FN(o_max)(N, th->eps_n);
    FT L = rho * safe_q;
    int cond;
    if (!limited) {
        FT xr_mean = L / safe_N;
        FT lam = M_CBRT(pi * pdf->rho_w / xr_mean);
        r.N0r = lam * safe_N;
        r.Dr_mean = 1 / lam;
        r.xr_mean = xr_mean;
        cond = (N < th->eps_n) || (q < th->eps_m);
    } else {
        FT xt = FN(o_clamp)(L / safe_N, pdf->xr_min, pdf->xr_max);                 /* Eq. 94 */
        FT N0 = FN(o_clamp)(safe_N * M_CBRT(pi * pdf->rho_w / xt), pdf->N0_min, pdf->N0_max); /* Eq. 95 */
        FT lam = FN(o_clamp)(M_SQRT(M_SQRT(pi * pdf->rho_w * N0 / L)), pdf->lambda_min,
                             pdf->lambda_max);                                   /* Eq. 96 */
        r.xr_mean = FN(o_clamp)(L * lam / N0, pdf->xr_min, pdf->xr_max);           /* Eq. 97 */
        r.N0r = N0;
        r.Dr_mean = 1 / lam;
        cond = (N < th->eps_n) && (q < th->eps_m);
    }
    if (cond) { r.N0r = 0; r.Dr_mean = 0; r.xr_mean = 0; }
    return r;
}

/* CM2.pdf_rain_parameters_mass — src/Microphysics2M.jl:141-146: returns Br (Ar = N Br / 3) */
static inline FT FN(o_pdf_rain_Br)(const TY(cmx_rain_pdf_sb2006) * pdf, int limited, FT q, FT rho,
                                  FT N, const TY(cmxo_thresholds) * th) {
    TY(cmxo_rain_pdf) r = FN(o_pdf_rain_parameters)(pdf, limited, q, rho, N, th);
    return M_CBRT(6 / r.xr_mean);
}

typedef struct TY(cmxo_lclrai_rates) { FT dq_lcl_dt, dN_lcl_dt, dq_rai_dt, dN_rai_dt; }
    TY(cmxo_lclrai_rates);

/* CM2.autoconversion — src/Microphysics2M.jl:396-427 */
static inline TY(cmxo_lclrai_rates) FN(o_autoconversion)(const TY(cmx_acnv_sb2006) * acnv,
                                                        const TY(cmx_cloud_pdf_sb2006) * pdf_c,
                                                        FT q_lcl, FT q_rai, FT rho, FT N_lcl,
                                                        const TY(cmxo_thresholds) * th) {
    TY(cmxo_lclrai_rates) r = {0, 0, 0, 0};
    FT nu_c = pdf_c->nu_c;
    FT safe_q_lcl = FN(o_max)(q_lcl, th->eps_m);
    FT safe_N_lcl = FN(o_max)(N_lcl, th->eps_n);
    FT L_lcl = rho * safe_q_lcl;
    FT x_lcl = FN(o_min)(acnv->x_star, L_lcl / safe_N_lcl);
    FT safe_q_rai = FN(o_max)((FT)0, q_rai);
    FT tau = 1 - safe_q_lcl / (safe_q_lcl + safe_q_rai);                          /* Eq. 5 */
    FT phi_au = 0;
    if (!(q_rai < th->eps_m)) {
        FT ta = M_POW(tau, acnv->a);
        phi_au = acnv->A * ta * M_POW(1 - ta, acnv->b);
    }
    FT dL_rai_dt = acnv->kcc / 20 / acnv->x_star * (nu_c + 2) * (nu_c + 4) /
                   ((nu_c + 1) * (nu_c + 1)) * (L_lcl * L_lcl) * (x_lcl * x_lcl) *
                   (1 + phi_au / ((1 - tau) * (1 - tau))) * acnv->rho_0 / rho;    /* Eq. 4 */
    FT dN_rai_dt = dL_rai_dt / acnv->x_star;
    FT dL_lcl_dt = -dL_rai_dt;
    FT dN_lcl_dt = -2 * dN_rai_dt;
    if (q_lcl < th->eps_m || N_lcl < th->eps_n) return r;
    r.dq_lcl_dt = dL_lcl_dt / rho;
    r.dN_lcl_dt = dN_lcl_dt;
    r.dq_rai_dt = dL_rai_dt / rho;
    r.dN_rai_dt = dN_rai_dt;
    return r;
}

/* CM2.accretion(::SB2006, …) — src/Microphysics2M.jl:445-470 */
static inline TY(cmxo_lclrai_rates) FN(o_accretion)(const TY(cmx_accr_sb2006) * accr, FT q_lcl,
                                                   FT q_rai, FT rho, FT N_lcl,
                                                   const TY(cmxo_thresholds) * th) {
    TY(cmxo_lclrai_rates) r = {0, 0, 0, 0};
    FT safe_q_lcl = FN(o_max)(q_lcl, th->eps_m);
    FT safe_q_rai = FN(o_max)(q_rai, th->eps_m);
    FT safe_N_lcl = FN(o_max)(N_lcl, th->eps_n);
    FT L_lcl = rho * safe_q_lcl;
    FT L_rai = rho * safe_q_rai;
    FT x_lcl = L_lcl / safe_N_lcl;
    FT tau = 1 - safe_q_lcl / (safe_q_lcl + safe_q_rai);
    FT phi_ac = M_POW(tau / (tau + accr->tau_0), accr->c);                         /* Eq. 8 */
    FT dL_rai_dt = accr->kcr * L_lcl * L_rai * phi_ac * M_SQRT(accr->rho_0 / rho); /* Eq. 7 */
    FT dL_lcl_dt = -dL_rai_dt;
    FT dN_lcl_dt = dL_lcl_dt / x_lcl;
    if (q_lcl < th->eps_m || q_rai < th->eps_m || N_lcl < th->eps_n) return r;
    r.dq_lcl_dt = dL_lcl_dt / rho;
    r.dN_lcl_dt = dN_lcl_dt;
    r.dq_rai_dt = dL_rai_dt / rho;
    r.dN_rai_dt = 0;
    return r;
}

/* CM2.cloud_liquid_self_collection — src/Microphysics2M.jl:488-501 */
static inline FT FN(o_cloud_liquid_self_collection)(const TY(cmx_acnv_sb2006) * acnv,
                                                   const TY(cmx_cloud_pdf_sb2006) * pdf_c,
                                                   FT q_lcl, FT rho, FT dN_lcl_dt_au,
                                                   const TY(cmxo_thresholds) * th) {
    FT nu_c = pdf_c->nu_c;
    FT L_lcl = rho * q_lcl;
    FT sc = -acnv->kcc * (nu_c + 2) / (nu_c + 1) * (acnv->rho_0 / rho) * (L_lcl * L_lcl) -
            dN_lcl_dt_au;
    return (q_lcl < th->eps_m) ? (FT)0 : sc;
}

/* CM2.rain_self_collection — src/Microphysics2M.jl:545-560 */
static inline FT FN(o_rain_self_collection)(const TY(cmx_rain_pdf_sb2006) * pdf, int limited,
                                           const TY(cmx_selfcol_sb2006) * self, FT q_rai, FT rho,
                                           FT N_rai, const TY(cmxo_thresholds) * th) {
    FT safe_q = FN(o_max)(q_rai, th->eps_m);
    FT safe_N = FN(o_max)(N_rai, th->eps_n);
    FT L_rai = rho * safe_q;
    FT Br = FN(o_pdf_rain_Br)(pdf, limited, safe_q, rho, safe_N, th);
    FT sc = -self->krr * N_rai * L_rai * M_SQRT(pdf->rho_0 / rho) *
            M_POW(1 + self->kappa_rr / Br, self->d);                              /* Eq. 11 */
    return (q_rai < th->eps_m || N_rai < th->eps_n) ? (FT)0 : sc;
}

/* CM2.rain_breakup — src/Microphysics2M.jl:579-601 (no factor 2 in the exponential
 * branch: the code, not docs/src/Microphysics2M.md:474, is followed — SURVEY App. A.9) */
static inline FT FN(o_rain_breakup)(const TY(cmx_rain_pdf_sb2006) * pdf, int limited,
                                   const TY(cmx_breakup_sb2006) * brek, FT q_rai, FT rho,
                                   FT N_rai, FT dN_rai_dt_sc, const TY(cmxo_thresholds) * th) {
    const FT pi = (FT)M_PI;
    FT safe_q = FN(o_max)(q_rai, th->eps_m);
    FT safe_N = FN(o_max)(N_rai, th->eps_n);
    TY(cmxo_rain_pdf) r = FN(o_pdf_rain_parameters)(pdf, limited, safe_q, rho, safe_N, th);
    FT Dr = M_CBRT(r.xr_mean * 6 / (pi * pdf->rho_w));
    FT dD = Dr - brek->Deq;
    FT phi_br = (Dr < brek->Dr_th) ? (FT)-1
                                  : ((Dr <= brek->Deq) ? brek->kbr * dD
                                                       : M_EXP(brek->kappa_br * dD) - 1);
    FT br = -(phi_br + 1) * dN_rai_dt_sc;                                         /* Eq. 13 */
    return (q_rai < th->eps_m || N_rai < th->eps_n) ? (FT)0 : br;
}

/* CM2.rain_terminal_velocity(::SB2006, ::SB2006VelType, …) — src/Microphysics2M.jl:685-702,
 * helper :720-739 */
static inline void FN(o_rain_terminal_velocity_sb)(const TY(cmx_rain_pdf_sb2006) * pdf, int limited,
                                                  const TY(cmx_sb2006_vel) * v, FT q_rai, FT rho,
                                                  FT N_rai, const TY(cmxo_thresholds) * th,
                                                  FT *vt0_out, FT *vt1_out, FT *scale) {
    FT safe_q = FN(o_max)(q_rai, th->eps_m);
    FT safe_N = FN(o_max)(N_rai, th->eps_n);
    TY(cmxo_rain_pdf) r = FN(o_pdf_rain_parameters)(pdf, limited, safe_q, rho, safe_N, th);
    FT Dr_mean = r.Dr_mean;
    FT pa0 = 1, pb0 = 1, pa1 = 1, pb1 = 1;
    if (!limited) {
        FT lam = 1 / Dr_mean;
        FT rc = -1 / (2 * v->cR) * M_LOG(v->aR / v->bR);
        FT ta = 2 * rc * lam, tb = 2 * rc * (lam + v->cR);
        pa0 = M_EXP(-ta);
        pb0 = M_EXP(-tb);
        pa1 = (ta * ta * ta + 3 * (ta * ta) + 6 * ta + 6) * M_EXP(-ta) / 6;
        pb1 = (tb * tb * tb + 3 * (tb * tb) + 6 * tb + 6) * M_EXP(-tb) / 6;
    }
    FT s = M_SQRT(v->rho_0 / rho);
    FT d1 = 1 + v->cR * Dr_mean;
    FT vt0 = FN(o_max)((FT)0, s * (v->aR * pa0 - v->bR * pb0 / d1));
    FT vt1 = FN(o_max)((FT)0, s * (v->aR * pa1 - v->bR * pb1 / ((d1 * d1) * (d1 * d1))));
    *vt0_out = (N_rai < th->eps_n) ? (FT)0 : vt0;
    *vt1_out = (q_rai < th->eps_m) ? (FT)0 : vt1;
    if (scale) {   /* aR·pa − bR·pb/(…) cancels near its zero crossing */
        scale[0] = s * (M_ABS(v->aR * pa0) + M_ABS(v->bR * pb0 / d1));
        scale[1] = s * (M_ABS(v->aR * pa1) + M_ABS(v->bR * pb1 / ((d1 * d1) * (d1 * d1))));
    }
}

/* CO.Chen2022_vel_coeffs(::Chen2022VelTypeRain, ρ) — src/Common.jl:290-302 */
static inline void FN(o_chen2022_rain_coeffs)(const TY(cmx_chen2022_rain_vel) * c, FT rho, FT aiu[3],
                                             FT bi[3], FT ciu[3]) {
    rho = FN(o_max)(rho, (FT)0);
    FT q = M_EXP(c->rho_0 * rho);
    FT ai[3] = {c->a[0] * q, c->a[1] * q, c->a[2] * q * M_POW(rho, c->a3_pow)};
    for (int i = 0; i < 3; ++i) {
        bi[i] = c->b[i] - c->b_rho * rho;
        aiu[i] = ai[i] * M_POW((FT)1000, bi[i]);
        ciu[i] = c->c[i] * 1000;
    }
}
/* CO.Chen2022_exponential_pdf — src/Common.jl:414-422 */
static inline FT FN(o_chen2022_exponential_pdf)(FT a, FT b, FT c, FT lam_inv, int k) {
    FT delta = (FT)(k + 1);
    FT fac = 1;
    for (int i = 2; i <= k; ++i) fac *= (FT)i;
    return a * M_EXP(-delta * M_LOG(lam_inv) - (b + delta) * M_LOG(1 / lam_inv + c)) *
           M_TGAMMA(b + delta) / fac;
}
/* CM2.rain_terminal_velocity(::SB2006, ::Chen2022VelTypeRain, …) — src/Microphysics2M.jl:703-719 */
static inline void FN(o_rain_terminal_velocity_chen)(const TY(cmx_rain_pdf_sb2006) * pdf,
                                                    int limited,
                                                    const TY(cmx_chen2022_rain_vel) * c, FT q_rai,
                                                    FT rho, FT N_rai,
                                                    const TY(cmxo_thresholds) * th, FT *vt0_out,
                                                    FT *vt3_out, FT *scale) {
    FT aiu[3], bi[3], ciu[3];
    FN(o_chen2022_rain_coeffs)(c, rho, aiu, bi, ciu);
    FT safe_q = FN(o_max)(q_rai, th->eps_m);
    FT safe_N = FN(o_max)(N_rai, th->eps_n);
    TY(cmxo_rain_pdf) r = FN(o_pdf_rain_parameters)(pdf, limited, safe_q, rho, safe_N, th);
    FT vt0 = 0, vt3 = 0, s0 = 0, s3 = 0;
    for (int i = 0; i < 3; ++i) {
        FT t0 = FN(o_chen2022_exponential_pdf)(aiu[i], bi[i], ciu[i], r.Dr_mean, 0);
        FT t3 = FN(o_chen2022_exponential_pdf)(aiu[i], bi[i], ciu[i], r.Dr_mean, 3);
        vt0 += t0; vt3 += t3; s0 += M_ABS(t0); s3 += M_ABS(t3);
    }
    if (scale) { scale[0] = s0; scale[1] = s3; }
    *vt0_out = (N_rai < th->eps_n) ? (FT)0 : FN(o_max)((FT)0, vt0);
    *vt3_out = (q_rai < th->eps_m) ? (FT)0 : FN(o_max)((FT)0, vt3);
}

/* CM2.Γ_incl — src/Microphysics2M.jl:746-753 */
static inline FT FN(o_gamma_incl)(FT a, FT x) {
    return M_EXP(-x) / (((FT)0.33 - (FT)0.7 * a) * M_POW(x, (FT)0.08 - (FT)0.93 * a) +
                        ((FT)1.34 - (FT)0.1 * a) * M_POW(x, (FT)0.8 - a));
}

/* CM2.rain_evaporation — src/Microphysics2M.jl:780-828 */
static inline void FN(o_rain_evaporation)(const TY(cmx_sb2006) * sb, int limited,
                                         const TY(cmx_air_properties) * aps,
                                         const TY(cmx_thermo) * tps, FT q_tot, FT q_lcl, FT q_icl,
                                         FT q_rai, FT q_sno, FT rho, FT N_rai, FT T,
                                         const TY(cmxo_thresholds) * th, FT *dN_out, FT *dq_out,
                                         FT *sN_out, FT *sq_out) {
    const FT pi = (FT)M_PI;
    const TY(cmx_evap_sb2006) *evap = &sb->evap;
    const TY(cmx_rain_pdf_sb2006) *pdf = &sb->pdf_r;
    FT S = FN(o_supersaturation_over_liquid)(tps, q_tot, q_lcl + q_rai, q_icl + q_sno, rho, T);
    FT x_star = pdf->xr_min;
    FT G = FN(o_G_func_liquid)(aps, tps, T, th);
    FT safe_q = FN(o_max)(q_rai, th->eps_m);
    FT safe_N = FN(o_max)(N_rai, th->eps_n);
    TY(cmxo_rain_pdf) r = FN(o_pdf_rain_parameters)(pdf, limited, safe_q, rho, safe_N, th);
    FT xr_mean = r.xr_mean;
    FT Dr = M_CBRT(6 * xr_mean / (pi * pdf->rho_w));
    FT t_star = M_CBRT((FT)6 * x_star / xr_mean);
    FT a_vent_0 = evap->a_vent_0_coeff * FN(o_gamma_incl)((FT)-1, t_star);
    FT b_vent_0 = evap->b_vent_0_coeff * FN(o_gamma_incl)(evap->beta_vent_0, t_star);
    FT a_vent_1 = evap->a_vent_1;
    FT b_vent_1 = evap->b_vent_1;
    FT N_Re = evap->alpha * M_POW(xr_mean, evap->beta) * M_SQRT(evap->rho_0 / rho) * Dr / aps->nu_air;
    FT cbrt_Sc = M_CBRT(aps->nu_air / FN(o_max)(aps->D_vapor, th->eps_1m));
    FT sqrt_N_Re = M_SQRT(N_Re);
    FT Fv0 = a_vent_0 + b_vent_0 * cbrt_Sc * sqrt_N_Re;
    FT Fv1 = a_vent_1 + b_vent_1 * cbrt_Sc * sqrt_N_Re;
    FT dN = FN(o_min)((FT)0, 2 * pi * G * S * N_rai * Dr * Fv0 / xr_mean);
    FT dq = FN(o_min)((FT)0, 2 * pi * G * S * N_rai * Dr * Fv1 / rho);
    if (q_rai < th->eps_m || xr_mean / x_star < th->eps_ft || N_rai <= th->eps_n || S >= 0) dN = 0;
    if (q_rai < th->eps_m || N_rai <= th->eps_n || S >= 0) dq = 0;
    *dN_out = dN;
    *dq_out = dq;
    /* S = p_v/p_sat − 1 cancels near saturation: the size of the two cancelling terms, carried
     * through the same prefactor, is the scale of the evaporation tendencies */
    if (sN_out) *sN_out = M_ABS(2 * pi * G * (S + 2) * N_rai * Dr * Fv0 / xr_mean);
    if (sq_out) *sq_out = M_ABS(2 * pi * G * (S + 2) * N_rai * Dr * Fv1 / rho);
}

/* CM2.number_tendency_from_mass_limits — src/Microphysics2M.jl:882-891 */
static inline FT FN(o_number_tendency_from_mass_limits)(FT x_min, FT x_max, FT tau, FT q, FT n,
                                                       const TY(cmxo_thresholds) * th) {
    FT n_target = (q < th->eps_m) ? (FT)0 : FN(o_clamp)(n, q / x_max, q / x_min);
    return (n_target - n) / tau;
}

/* DT.generalized_gamma_Mⁿ + CM2.cloud_terminal_velocity — src/Microphysics2M.jl:647-664,
 * src/DistributionTools.jl:109-112, log_pdf_cloud_parameters_mass CM2:176-191 */
static inline void FN(o_cloud_terminal_velocity)(const TY(cmx_cloud_pdf_sb2006) * pdf_c, FT rho_w,
                                                FT grav, FT nu_air, FT q_liq, FT rho, FT N_liq,
                                                const TY(cmxo_thresholds) * th, FT *vt0_out,
                                                FT *vt1_out) {
    const FT pi = (FT)M_PI;
    FT nu_c = pdf_c->nu_c, mu_c = pdf_c->mu_c;
    FT safe_q = FN(o_max)(q_liq, th->eps_m);
    FT safe_N = FN(o_max)(N_liq, th->eps_n);
    FT L = rho * safe_q;
    FT logx = M_LOG(L / safe_N);
    FT logB = -mu_c * (logx + pdf_c->loggamma_z1 - pdf_c->loggamma_z2);
    FT Bc = M_EXP(logB);
    FT pref = (FT)(1.0 / 18.0) * M_CBRT(((FT)6 / rho_w / pi) * ((FT)6 / rho_w / pi)) *
              (rho_w / rho - 1) * grav / nu_air;
    /* Mⁿ = N B^(-n/μ) Γ((ν+1+n)/μ)/Γ((ν+1)/μ)   (DistributionTools.jl:109-112) */
    FT z1 = (nu_c + 1) / mu_c;
    FT n0 = (FT)(2.0 / 3.0), n1 = (FT)(5.0 / 3.0);
    FT M0 = safe_N * M_EXP(-n0 / mu_c * logB + M_LGAMMA(z1 + n0 / mu_c) - M_LGAMMA(z1));
    FT M1 = safe_N * M_EXP(-n1 / mu_c * logB + M_LGAMMA(z1 + n1 / mu_c) - M_LGAMMA(z1));
    (void)Bc;
    FT vt0 = pref * M0 / safe_N;
    FT vt1 = pref * M1 / rho / safe_q;
    int cond = (N_liq < th->eps_n) || (q_liq < th->eps_m);
    *vt0_out = cond ? (FT)0 : vt0;
    *vt1_out = cond ? (FT)0 : vt1;
}

/* ---- BulkMicrophysicsTendencies.jl ---------------------------------------- */
typedef struct TY(cmxo_warm_rain_out) {
    FT dq_lcl_dt, dn_lcl_dt, dq_rai_dt, dn_rai_dt;  /* BMT:852-853 (non-zero fields) */
    FT vt_rai_n, vt_rai_m;                          /* CM2.rain_terminal_velocity     */
    FT scale[6];                                    /* Σ|terms| of each output */
    int near_branch;                                /* 1 if within `branch_margin` of a discontinuity */
} TY(cmxo_warm_rain_out);

/* bulk_microphysics_tendencies(::Microphysics2Moment, mp{WR,Nothing}, …) — BMT:820-854,
 * warm_rain_tendencies_2m — BMT:707-782 */
static inline TY(cmxo_warm_rain_out) FN(o_bulk_tendencies_2m_warm)(
    const TY(cmx_warm_rain_2m) * wr, const TY(cmx_thermo) * tps, const TY(cmx_rain_vel) * vel,
    uint32_t flags, const TY(cmxo_thresholds) * th, FT branch_margin, FT rho, FT T, FT q_tot,
    FT q_lcl, FT n_lcl, FT q_rai, FT n_rai, FT q_ice) {   /* q_ice = 0 for the warm-only entry, the P3 ice content for BMT:942 */
    TY(cmxo_warm_rain_out) o;
    const TY(cmx_sb2006) *sb = &wr->seifert_beheng;
    const TY(cmx_air_properties) *aps = &wr->air_properties;
    const int limited = (flags & CMX_SB2006_LIMITED) != 0;
    /* clamp_to_nonneg — BMT:828-837 (T is not clamped) */
    rho = FN(o_max)((FT)0, rho);
    q_tot = FN(o_max)((FT)0, q_tot);
    q_lcl = FN(o_max)((FT)0, q_lcl);
    q_rai = FN(o_max)((FT)0, q_rai);
    n_lcl = FN(o_max)((FT)0, n_lcl);
    n_rai = FN(o_max)((FT)0, n_rai);
    q_ice = FN(o_max)((FT)0, q_ice);
    FT N_lcl = rho * n_lcl;                                                       /* BMT:718-719 */
    FT N_rai = rho * n_rai;
    FT dq_lcl = 0, dq_rai = 0, dn_lcl = 0, dn_rai = 0;
    FT s_ql = 0, s_qr = 0, s_nl = 0, s_nr = 0;
    /* condensation / evaporation of cloud liquid — BMT:731-738 */
    FT sc_cond;
    FT cond = FN(o_conv_q_vap_to_q_lcl_const)(wr->condevap_tau_relax, tps, q_tot, q_lcl, q_ice,
                                             q_rai, (FT)0, rho, T, &sc_cond);
    dq_lcl += cond;
    dn_lcl += 0;
    s_ql += sc_cond;
    /* rain evaporation — BMT:741-744 */
    FT evN, evq, evNs, evqs;
    FN(o_rain_evaporation)(sb, limited, aps, tps, q_tot, q_lcl, q_ice, q_rai, (FT)0, rho, N_rai, T,
                          th, &evN, &evq, &evNs, &evqs);
    dq_rai += evq;
    dn_rai += evN / rho;
    s_qr += evqs;
    s_nr += evNs / rho;
    /* autoconversion — BMT:747-751 */
    TY(cmxo_lclrai_rates) au = FN(o_autoconversion)(&sb->acnv, &sb->pdf_c, q_lcl, q_rai, rho, N_lcl, th);
    dq_lcl += au.dq_lcl_dt;
    dq_rai += au.dq_rai_dt;
    dn_lcl += au.dN_lcl_dt / rho;
    dn_rai += au.dN_rai_dt / rho;
    s_ql += M_ABS(au.dq_lcl_dt);
    s_qr += M_ABS(au.dq_rai_dt);
    s_nl += M_ABS(au.dN_lcl_dt / rho);
    s_nr += M_ABS(au.dN_rai_dt / rho);
    /* cloud liquid self-collection — BMT:754-755 */
    FT lsc = FN(o_cloud_liquid_self_collection)(&sb->acnv, &sb->pdf_c, q_lcl, rho, au.dN_lcl_dt, th);
    dn_lcl += lsc / rho;
    s_nl += M_ABS(lsc / rho);
    /* accretion — BMT:758-761 */
    TY(cmxo_lclrai_rates) ac = FN(o_accretion)(&sb->accr, q_lcl, q_rai, rho, N_lcl, th);
    dq_lcl += ac.dq_lcl_dt;
    dq_rai += ac.dq_rai_dt;
    dn_lcl += ac.dN_lcl_dt / rho;
    s_ql += M_ABS(ac.dq_lcl_dt);
    s_qr += M_ABS(ac.dq_rai_dt);
    s_nl += M_ABS(ac.dN_lcl_dt / rho);
    /* rain self-collection — BMT:764-765 */
    FT rsc = FN(o_rain_self_collection)(&sb->pdf_r, limited, &sb->self, q_rai, rho, N_rai, th);
    dn_rai += rsc / rho;
    s_nr += M_ABS(rsc / rho);
    /* rain breakup — BMT:768-769 */
    FT rbr = FN(o_rain_breakup)(&sb->pdf_r, limited, &sb->brek, q_rai, rho, N_rai, rsc, th);
    dn_rai += rbr / rho;
    s_nr += M_ABS(rbr / rho);
    /* number adjustment for mass limits — BMT:773-779.  Its own two terms
     * (n_target, n) cancel, so both enter the scale. */
    FT na_l = FN(o_number_tendency_from_mass_limits)(sb->pdf_c.xc_min, sb->pdf_c.xc_max,
                                                    sb->numadj.tau, q_lcl, n_lcl, th);
    dn_lcl += na_l;
    s_nl += (M_ABS(na_l) + 2 * M_ABS(n_lcl)) / M_ABS(sb->numadj.tau) * (na_l != 0);
    FT na_r = FN(o_number_tendency_from_mass_limits)(sb->pdf_r.xr_min, sb->pdf_r.xr_max,
                                                    sb->numadj.tau, q_rai, n_rai, th);
    dn_rai += na_r;
    s_nr += (M_ABS(na_r) + 2 * M_ABS(n_rai)) / M_ABS(sb->numadj.tau) * (na_r != 0);

    o.dq_lcl_dt = dq_lcl;
    o.dn_lcl_dt = dn_lcl;
    o.dq_rai_dt = dq_rai;
    o.dn_rai_dt = dn_rai;
    o.scale[0] = s_ql;
    o.scale[1] = s_nl;
    o.scale[2] = s_qr;
    o.scale[3] = s_nr;
    o.vt_rai_n = 0;
    o.vt_rai_m = 0;
    o.scale[4] = o.scale[5] = 0;
    if (vel && (flags & CMX_VEL_SB2006))
        FN(o_rain_terminal_velocity_sb)(&sb->pdf_r, limited, &vel->sb2006, q_rai, rho, N_rai, th,
                                       &o.vt_rai_n, &o.vt_rai_m, &o.scale[4]);
    else if (vel && (flags & CMX_VEL_CHEN2022))
        FN(o_rain_terminal_velocity_chen)(&sb->pdf_r, limited, &vel->chen2022, q_rai, rho, N_rai,
                                         th, &o.vt_rai_n, &o.vt_rai_m, &o.scale[4]);
    /* the breakup function Φ_br jumps at Dr = Dr_th (−1 → kbr(Dr_th − Deq), CM2:596): points whose
     * mean-volume diameter lies within `branch_margin` (relative) of the threshold may legitimately
     * land on either branch in another precision. */
    o.near_branch = 0;
    if (!(q_rai < th->eps_m || N_rai < th->eps_n)) {
        TY(cmxo_rain_pdf) pr = FN(o_pdf_rain_parameters)(&sb->pdf_r, limited, FN(o_max)(q_rai, th->eps_m), rho,
                                                        FN(o_max)(N_rai, th->eps_n), th);
        FT Dr = M_CBRT(pr.xr_mean * 6 / ((FT)M_PI * sb->pdf_r.rho_w));
        o.near_branch = M_ABS(Dr - sb->brek.Dr_th) <= branch_margin * sb->brek.Dr_th;
    }
    return o;
}

/* ---- exported array drivers ------------------------------------------------ */
void FN(cmxo_default_thresholds)(TY(cmxo_thresholds) * th, int float32_gates) {
    if (float32_gates) {
        th->eps_m = th->eps_n = th->eps_ft = (FT)FLT_EPSILON;
        th->eps_1m = (FT)cbrtf(FLT_MIN);
    } else {
        th->eps_m = th->eps_n = th->eps_ft = (FT)DBL_EPSILON;
        th->eps_1m = (FT)cbrt(DBL_MIN);
    }
}

/* oracle twin of cmx_sb2006_warm_rain_tendencies_* (include/cmx.h); host pointers;
 * `scale` = 6 optional columns (may be NULL / hold NULLs) */
void FN(cmxo_sb2006_warm_rain_tendencies)(
    const TY(cmx_warm_rain_2m) * wr, const TY(cmx_thermo) * tps, const TY(cmx_rain_vel) * vel,
    uint32_t flags, const TY(cmxo_thresholds) * th, int64_t n, const FT *rho, const FT *T,
    const FT *q_tot, const FT *q_lcl, const FT *n_lcl, const FT *q_rai, const FT *n_rai,
    FT *dq_lcl_dt, FT *dn_lcl_dt, FT *dq_rai_dt, FT *dn_rai_dt, FT *vt_rai_n, FT *vt_rai_m,
    FT *const *scale, uint8_t *near_branch, FT branch_margin, int32_t nthreads) {
    (void)nthreads;
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
    for (int64_t i = 0; i < n; ++i) {
        TY(cmxo_warm_rain_out) o = FN(o_bulk_tendencies_2m_warm)(
            wr, tps, vel, flags, th, branch_margin, rho[i], T[i], q_tot[i], q_lcl[i], n_lcl[i], q_rai[i],
            n_rai[i], (FT)0);
        dq_lcl_dt[i] = o.dq_lcl_dt;
        dn_lcl_dt[i] = o.dn_lcl_dt;
        dq_rai_dt[i] = o.dq_rai_dt;
        dn_rai_dt[i] = o.dn_rai_dt;
        if (vt_rai_n) vt_rai_n[i] = o.vt_rai_n;
        if (vt_rai_m) vt_rai_m[i] = o.vt_rai_m;
        if (scale)
            for (int k = 0; k < 6; ++k)
                if (scale[k]) scale[k][i] = o.scale[k];
        if (near_branch) near_branch[i] = (uint8_t)o.near_branch;
    }
}

/* oracle twin of cmx_sb2006_process_rates_* — SB2006_2M_kernel, test/gpu_tests.jl:220-235 */
void FN(cmxo_sb2006_process_rates)(const TY(cmx_warm_rain_2m) * wr, const TY(cmx_thermo) * tps,
                                  const TY(cmx_rain_vel) * vel, uint32_t flags,
                                  const TY(cmxo_thresholds) * th, int64_t n, const FT *q_tot,
                                  const FT *q_lcl, const FT *q_rai, const FT *N_lcl,
                                  const FT *N_rai, const FT *rho, const FT *T,
                                  FT *const out[CMX_SB2006_NPROC]) {
    const TY(cmx_sb2006) *sb = &wr->seifert_beheng;
    const int limited = (flags & CMX_SB2006_LIMITED) != 0;
#define PUT(col, v) do { if (out[col]) out[col][i] = (v); } while (0)
    for (int64_t i = 0; i < n; ++i) {
        TY(cmxo_lclrai_rates) au =
            FN(o_autoconversion)(&sb->acnv, &sb->pdf_c, q_lcl[i], q_rai[i], rho[i], N_lcl[i], th);
        FT lsc = FN(o_cloud_liquid_self_collection)(&sb->acnv, &sb->pdf_c, q_lcl[i], rho[i],
                                                   au.dN_lcl_dt, th);
        TY(cmxo_lclrai_rates) ac = FN(o_accretion)(&sb->accr, q_lcl[i], q_rai[i], rho[i], N_lcl[i], th);
        FT rsc = FN(o_rain_self_collection)(&sb->pdf_r, limited, &sb->self, q_rai[i], rho[i], N_rai[i], th);
        FT rbr = FN(o_rain_breakup)(&sb->pdf_r, limited, &sb->brek, q_rai[i], rho[i], N_rai[i], rsc, th);
        FT v0 = 0, v1 = 0;
        if (vel && (flags & CMX_VEL_SB2006))
            FN(o_rain_terminal_velocity_sb)(&sb->pdf_r, limited, &vel->sb2006, q_rai[i], rho[i],
                                           N_rai[i], th, &v0, &v1, NULL);
        else if (vel && (flags & CMX_VEL_CHEN2022))
            FN(o_rain_terminal_velocity_chen)(&sb->pdf_r, limited, &vel->chen2022, q_rai[i], rho[i],
                                             N_rai[i], th, &v0, &v1, NULL);
        FT evN, evq;
        FN(o_rain_evaporation)(sb, limited, &wr->air_properties, tps, q_tot[i], q_lcl[i], (FT)0,
                              q_rai[i], (FT)0, rho[i], N_rai[i], T[i], th, &evN, &evq, NULL, NULL);
        FT na_r = FN(o_number_tendency_from_mass_limits)(sb->pdf_r.xr_min, sb->pdf_r.xr_max,
                                                        sb->numadj.tau, q_rai[i], N_rai[i] / rho[i], th);
        FT na_l = FN(o_number_tendency_from_mass_limits)(sb->pdf_c.xc_min, sb->pdf_c.xc_max,
                                                        sb->numadj.tau, q_lcl[i], N_lcl[i] / rho[i], th);
        FT ce = FN(o_conv_q_vap_to_q_lcl_const)(wr->condevap_tau_relax, tps, q_tot[i], q_lcl[i],
                                               (FT)0, q_rai[i], (FT)0, rho[i], T[i], NULL);
        PUT(CMX_SB_ACNV_DQ_LCL, au.dq_lcl_dt);
        PUT(CMX_SB_ACNV_DN_LCL, au.dN_lcl_dt);
        PUT(CMX_SB_ACNV_DQ_RAI, au.dq_rai_dt);
        PUT(CMX_SB_ACNV_DN_RAI, au.dN_rai_dt);
        PUT(CMX_SB_LCL_SELFCOL, lsc);
        PUT(CMX_SB_ACCR_DQ_LCL, ac.dq_lcl_dt);
        PUT(CMX_SB_ACCR_DN_LCL, ac.dN_lcl_dt);
        PUT(CMX_SB_ACCR_DQ_RAI, ac.dq_rai_dt);
        PUT(CMX_SB_RAI_SELFCOL, rsc);
        PUT(CMX_SB_RAI_BREAKUP, rbr);
        PUT(CMX_SB_RAI_VEL_N, v0);
        PUT(CMX_SB_RAI_VEL_M, v1);
        PUT(CMX_SB_EVAP_DN_RAI, evN);
        PUT(CMX_SB_EVAP_DQ_RAI, evq);
        PUT(CMX_SB_NUMADJ_RAI, na_r);
        PUT(CMX_SB_NUMADJ_LCL, na_l);
        PUT(CMX_SB_CONDEVAP, ce);
        /* ∂rain_evaporation_∂N_rai_∂q_rai — src/Microphysics2M.jl:844-855 */
        PUT(CMX_SB_DEVAP_DN_RAI, N_rai[i] > th->eps_n ? evN / N_rai[i] : (FT)0);
        PUT(CMX_SB_DEVAP_DQ_RAI, q_rai[i] > th->eps_m ? evq / q_rai[i] : (FT)0);
    }
#undef PUT
}

/* ---- Common.jl water activities + IceNucleation.jl ------------------------ */
/* CO.a_w_ice — src/Common.jl:267-271 */
static inline FT FN(o_a_w_ice)(const TY(cmx_thermo) * tps, FT T) {
    return FN(o_psat_ice)(tps, T) / FN(o_psat_liquid)(tps, T);
}
/* CO.a_w_eT — src/Common.jl:250-253 */
static inline FT FN(o_a_w_eT)(const TY(cmx_thermo) * tps, FT e, FT T) {
    return e / FN(o_psat_liquid)(tps, T);
}
/* CMI_het.ABIFM_J — src/IceNucleation.jl:124-134 */
static inline FT FN(o_ABIFM_J)(const TY(cmx_abifm_dust) * dust, FT delta_a_w) {
    FT logJ = dust->ABIFM_m * delta_a_w + dust->ABIFM_c;
    return M_POW((FT)10, logJ + 4);
}
/* CMI_hom.homogeneous_J_cubic — src/IceNucleation.jl:557-565; the DomainError becomes NaN + *err = 1 */
static inline FT FN(o_homogeneous_J_cubic)(const TY(cmx_koop2000) * ip, FT d, int *err) {
    if (!(ip->delta_a_w_min <= d && d <= ip->delta_a_w_max)) {
        *err = 1;
        return (FT)NAN;
    }
    FT logJ = ip->c1 + ip->c2 * d - ip->c3 * (d * d) + ip->c4 * (d * d * d);
    return M_POW((FT)10, logJ + 6);
}
/* CMI_hom.homogeneous_J_linear — src/IceNucleation.jl:581-584 */
static inline FT FN(o_homogeneous_J_linear)(const TY(cmx_koop2000) * ip, FT d) {
    FT logJ = ip->linear_c2 * d + ip->linear_c1;
    return M_POW((FT)10, logJ + 6);
}

/* oracle twin of cmx_ice_nucleation_rates_* (include/cmx.h); returns the number of domain errors */
int64_t FN(cmxo_ice_nucleation_rates)(const TY(cmx_thermo) * tps, const TY(cmx_abifm_dust) * dust,
                                     const TY(cmx_koop2000) * koop, uint32_t flags, int64_t n,
                                     const FT *T, const FT *a_w, const FT *r, FT *delta_a_w, FT *J_het,
                                     FT *J_hom, FT *rate_het, FT *rate_hom) {
    const FT pi = (FT)M_PI;
    int64_t nerr = 0;
    for (int64_t i = 0; i < n; ++i) {
        FT d = a_w[i] - FN(o_a_w_ice)(tps, T[i]);
        FT jh = FN(o_ABIFM_J)(dust, d);
        int err = 0;
        FT jo = (flags & CMX_ICENUC_HOM_LINEAR) ? FN(o_homogeneous_J_linear)(koop, d)
                                                : FN(o_homogeneous_J_cubic)(koop, d, &err);
        nerr += err;
        if (delta_a_w) delta_a_w[i] = d;
        if (J_het) J_het[i] = jh;
        if (J_hom) J_hom[i] = jo;
        if (rate_het) rate_het[i] = jh * (4 * pi * (r[i] * r[i]));          /* J·A_aer   parcel/ParcelTendencies.jl:120-133 */
        if (rate_hom) rate_hom[i] = jo * ((FT)4 / 3 * pi * (r[i] * r[i] * r[i])); /* J·V   parcel/ParcelTendencies.jl:194-205 */
    }
    return nerr;
}
void FN(cmxo_water_activity)(const TY(cmx_thermo) * tps, int64_t n, const FT *T, const FT *e, FT *a_w_ice,
                            FT *a_w_eT) {
    for (int64_t i = 0; i < n; ++i) {
        if (a_w_ice) a_w_ice[i] = FN(o_a_w_ice)(tps, T[i]);
        if (a_w_eT) a_w_eT[i] = FN(o_a_w_eT)(tps, e[i], T[i]);
    }
}

/* scalar probes used by the known-answer tests */
FT FN(cmxo_psat_liquid)(const TY(cmx_thermo) * p, FT T) { return FN(o_psat_liquid)(p, T); }
FT FN(cmxo_psat_ice)(const TY(cmx_thermo) * p, FT T) { return FN(o_psat_ice)(p, T); }
FT FN(cmxo_gamma_incl)(FT a, FT x) { return FN(o_gamma_incl)(a, x); }
void FN(cmxo_pdf_rain_parameters)(const TY(cmx_rain_pdf_sb2006) * pdf, int limited, FT q, FT rho,
                                 FT N, const TY(cmxo_thresholds) * th, FT out[3]) {
    TY(cmxo_rain_pdf) r = FN(o_pdf_rain_parameters)(pdf, limited, q, rho, N, th);
    out[0] = r.N0r; out[1] = r.Dr_mean; out[2] = r.xr_mean;
}
void FN(cmxo_cloud_terminal_velocity)(const TY(cmx_cloud_pdf_sb2006) * pdf_c, FT rho_w, FT grav,
                                     FT nu_air, FT q_liq, FT rho, FT N_liq,
                                     const TY(cmxo_thresholds) * th, FT out[2]) {
    FN(o_cloud_terminal_velocity)(pdf_c, rho_w, grav, nu_air, q_liq, rho, N_liq, th, &out[0], &out[1]);
}
void FN(cmxo_chen2022_rain_coeffs)(const TY(cmx_chen2022_rain_vel) * c, FT rho, FT out[9]) {
    FN(o_chen2022_rain_coeffs)(c, rho, out, out + 3, out + 6);
}

/* oracle twin of cmx_sb2006_cloud_terminal_velocity_* (CM2.cloud_terminal_velocity, src/Microphysics2M.jl:647-664) */
void FN(cmxo_sb2006_cloud_terminal_velocity)(const TY(cmx_cloud_pdf_sb2006) * pdf, const TY(cmx_stokes_vel) * v, const TY(cmxo_thresholds) * th,
                                            int64_t n, const FT *q_liq, const FT *rho, const FT *N_liq, FT *vt0, FT *vt1) {
    for (int64_t i = 0; i < n; ++i)
        FN(o_cloud_terminal_velocity)(pdf, v->rho_w, v->grav, v->nu_air, q_liq[i], rho[i], N_liq[i], th, &vt0[i], &vt1[i]);
}

/* ---- bulk 2M cloud → rain conversion of KK2000 / B1994 / TC1980 / LD2004 — src/Microphysics2M.jl:920-1003 -------- */
/* CO.logistic_function — src/Common.jl:125-139 (log1pexp: LogExpFunctions, stable form) */
static inline FT FN(o_logistic_function)(FT x, FT x_0, FT k, FT eps) {
    x = FN(o_max)((FT)0, x);
    FT xs = FN(o_max)(x, eps), x0s = FN(o_max)(x_0, eps);
    FT z = k * (xs / x0s - x0s / xs);
    FT mz = -z;
    FT l1pe = mz > 0 ? mz + M_LOG1P(M_EXP(-mz)) : M_LOG1P(M_EXP(mz));
    FT r = M_EXP(-l1pe);
    return x < eps ? (FT)0 : (x_0 < eps ? (FT)1 : r);
}
static inline FT FN(o_acnv_2m)(const TY(cmx_bulk_2m_schemes) * p, uint32_t scheme, const TY(cmxo_thresholds) * th, FT q_lcl, FT rho, FT N_d) {
    const int smooth = (scheme & CMX_2M_SMOOTH_TRANSITION) != 0;
    const FT eps = th->eps_1m;
    switch (scheme & 0xffu) {
    case CMX_2M_KK2000: {
        FT q = FN(o_max)((FT)0, q_lcl);
        return p->kk2000.acnv_A * M_POW(q, p->kk2000.acnv_a) * M_POW(N_d, p->kk2000.acnv_b) * M_POW(rho, p->kk2000.acnv_c);
    }
    case CMX_2M_B1994: {
        const TY(cmx_b1994) *b = &p->b1994;
        FT q = FN(o_max)((FT)0, q_lcl), d;
        if (smooth) {
            FT f_low = FN(o_logistic_function)(N_d, b->acnv_N_0, b->acnv_k, eps);
            d = f_low * b->acnv_d_low + (1 - f_low) * b->acnv_d_high;
        } else {
            d = N_d >= b->acnv_N_0 ? b->acnv_d_low : b->acnv_d_high;
        }
        return b->acnv_C * M_POW(d, b->acnv_a) * M_POW(q * rho, b->acnv_b) * M_POW(N_d, b->acnv_c) / rho;
    }
    case CMX_2M_TC1980: {
        const TY(cmx_tc1980) *t = &p->tc1980;
        FT q = FN(o_max)((FT)0, q_lcl);
        FT thr = t->acnv_m0_liq_coeff * N_d / rho * M_POW(t->acnv_r_0, t->acnv_me_liq);
        FT out = smooth ? FN(o_logistic_function)(q, thr, t->acnv_k, eps) : (FT)(q - thr > 0);
        return t->acnv_D * M_POW(q, t->acnv_a) * M_POW(N_d, t->acnv_b) * out;
    }
    default: {   /* LD2004 */
        const TY(cmx_ld2004) *l = &p->ld2004;
        if (q_lcl <= th->eps_m) return 0;
        FT r_vol = M_CBRT(3 * q_lcl * rho / 4 / (FT)M_PI / l->rho_w / N_d) * 1000000;
        FT beta_6 = M_CBRT((r_vol + 3) / r_vol);
        FT E = l->E_0 * M_POW(beta_6, (FT)6);
        FT R_6 = beta_6 * r_vol;
        FT R_6C = l->R_6C_0 / M_CBRT(M_SQRT(q_lcl * rho)) / M_SQRT(R_6);
        FT out = smooth ? FN(o_logistic_function)(R_6, R_6C, l->k, eps) : (FT)(R_6 - R_6C > 0);
        return E * M_POW(q_lcl * rho, (FT)3) / N_d / rho * out;
    }
    }
}
static inline FT FN(o_accr_2m)(const TY(cmx_bulk_2m_schemes) * p, uint32_t scheme, FT q_lcl, FT q_rai, FT rho) {
    FT ql = FN(o_max)((FT)0, q_lcl), qr = FN(o_max)((FT)0, q_rai);
    switch (scheme & 0xffu) {
    case CMX_2M_KK2000: return p->kk2000.accr_A * M_POW(ql * qr, p->kk2000.accr_a) * M_POW(rho, p->kk2000.accr_b);
    case CMX_2M_B1994: return p->b1994.accr_A * ql * rho * qr;
    default: return p->tc1980.accr_A * ql * qr;
    }
}
void FN(cmxo_bulk_2m_cloud_to_rain)(const TY(cmx_bulk_2m_schemes) * p, uint32_t scheme, const TY(cmxo_thresholds) * th, int64_t n,
                                   const FT *q_lcl, const FT *q_rai, const FT *rho, const FT *N_d, FT *acnv, FT *accr) {
    for (int64_t i = 0; i < n; ++i) {
        if (acnv) acnv[i] = FN(o_acnv_2m)(p, scheme, th, q_lcl[i], rho[i], N_d[i]);
        if (accr) accr[i] = FN(o_accr_2m)(p, scheme, q_lcl[i], q_rai[i], rho[i]);
    }
}

#include "cmx_oracle_1m_impl.h"
#include "cmx_oracle_arg_impl.h"
#include "cmx_oracle_p3_impl.h"
#include "cmx_oracle_sed_impl.h"
#include "cmx_oracle_p3col_impl.h"
#include "cmx_oracle_column_impl.h"
#include "cmx_oracle_extra_impl.h"
#include "cmx_oracle_dist_impl.h"
#include "cmx_oracle_diag_impl.h"

#undef CAT_
#undef CAT
#undef FN
#undef TY
