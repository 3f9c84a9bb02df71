/*
 * cmx_oracle_column_impl.h — oracle (TEST INFRASTRUCTURE) of the fused column step cmx_sb2006_column_tendencies_sedimentation_*
 * (SURVEY.md §8f-4).  Included from cmx_oracle_impl.h.
 *
 * PARITY UNPINNED FOR THE FLUX STEP: the sedimentation flux divergence is the HOST MODEL's operator (ClimaAtmos advects
 * precipitation with a first-order upwind, "right-biased" flux), it is not part of CloudMicrophysics.jl and the reference holds no
 * test vector for it.  What this file restates is the formula documented in include/cmx.h,
 *     F_k = ρ_k χ_k w_k,      ∂χ_k/∂t |sed = (F_{k+1} − F_k) / (ρ_k Δz_k),      F_{n_lev} = 0,
 * evaluated in the straightforward order, per column, on top of the PINNED pieces it consumes:
 *   the tendencies          o_bulk_tendencies_2m_warm      (BMT:820-854, :707-782; KATs test/gpu_tests.jl:821-872)
 *   the rain fall speeds    o_rain_terminal_velocity_sb / _chen   (CM2:685-719; KATs :864-866, test/microphysics2M_tests.jl:484-498)
 *   the cloud fall speeds   o_cloud_terminal_velocity      (CM2:647-664)
 * with the clamped state of BMT:828-837.  scale[k] = the tendency's own Σ|terms| + (S_{k+1} + S_k)/(ρ_k Δz_k), S = ρ χ · (the fall
 * speed's own Σ|terms|: aR·pa − bR·pb/(1+cR·D̄) cancels towards small drops, CM2:698-700) ≥ |F|.
 */
void FN(cmxo_sb2006_column_tendencies_sedimentation)(
    const TY(cmx_warm_rain_2m) * wr, const TY(cmx_thermo) * tps, const TY(cmx_rain_vel) * vel, const TY(cmx_stokes_vel) * cloud_vel,
    uint32_t flags, const TY(cmxo_thresholds) * th, int64_t n_col, int32_t n_lev, const FT *inv_dz, const FT *rho, const FT *T,
    const FT *q_tot, const FT *q_lcl, const FT *n_lcl, const FT *q_rai, const FT *n_rai, FT *dq_lcl_dt, FT *dn_lcl_dt, FT *dq_rai_dt,
    FT *dn_rai_dt, FT *precip_flux, FT *const *scale, uint8_t *near_branch, FT branch_margin, int32_t nthreads) {
    const TY(cmx_sb2006) *sb = &wr->seifert_beheng;
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
    for (int64_t col = 0; col < n_col; ++col) {
        FT Fqr_up = 0, Fnr_up = 0, Fql_up = 0, Fnl_up = 0;            /* F_{n_lev} = 0: nothing enters through the model top */
        FT Sqr_up = 0, Snr_up = 0;
        for (int32_t k = n_lev - 1; k >= 0; --k) {
            const int64_t i = col * (int64_t)n_lev + k;
            TY(cmxo_warm_rain_out) o = FN(o_bulk_tendencies_2m_warm)(wr, tps, vel, flags, th, branch_margin, rho[i], T[i], q_tot[i],
                                                                     q_lcl[i], n_lcl[i], q_rai[i], n_rai[i], (FT)0);
            /* clamp_to_nonneg — BMT:828-837 */
            const FT r = FN(o_max)((FT)0, rho[i]), ql = FN(o_max)((FT)0, q_lcl[i]), nl = FN(o_max)((FT)0, n_lcl[i]);
            const FT qr = FN(o_max)((FT)0, q_rai[i]), nr = FN(o_max)((FT)0, n_rai[i]);
            const FT Fqr = r * qr * o.vt_rai_m, Fnr = r * nr * o.vt_rai_n;
            const FT Sqr = r * qr * FN(o_max)(o.scale[5], o.vt_rai_m), Snr = r * nr * FN(o_max)(o.scale[4], o.vt_rai_n);
            FT Fql = 0, Fnl = 0;
            if (cloud_vel) {
                FT c_n, c_m;
                FN(o_cloud_terminal_velocity)(&sb->pdf_c, cloud_vel->rho_w, cloud_vel->grav, cloud_vel->nu_air, ql, r, r * nl, th, &c_n, &c_m);
                Fql = r * ql * c_m;
                Fnl = r * nl * c_n;
            }
            const FT w = inv_dz[k] / r;
            dq_lcl_dt[i] = o.dq_lcl_dt + (Fql_up - Fql) * w;
            dn_lcl_dt[i] = o.dn_lcl_dt + (Fnl_up - Fnl) * w;
            dq_rai_dt[i] = o.dq_rai_dt + (Fqr_up - Fqr) * w;
            dn_rai_dt[i] = o.dn_rai_dt + (Fnr_up - Fnr) * w;
            if (scale) {
                if (scale[0]) scale[0][i] = o.scale[0] + (M_ABS(Fql_up) + M_ABS(Fql)) * w;
                if (scale[1]) scale[1][i] = o.scale[1] + (M_ABS(Fnl_up) + M_ABS(Fnl)) * w;
                if (scale[2]) scale[2][i] = o.scale[2] + (Sqr_up + Sqr) * w;
                if (scale[3]) scale[3][i] = o.scale[3] + (Snr_up + Snr) * w;
            }
            if (near_branch) near_branch[i] = (uint8_t)o.near_branch;
            if (k == 0 && precip_flux) precip_flux[col] = Fqr;
            Fqr_up = Fqr; Fnr_up = Fnr; Fql_up = Fql; Fnl_up = Fnl;
            Sqr_up = Sqr; Snr_up = Snr;
        }
    }
}

/*
 * Oracle of cmx_mp1m_column_tendencies_sedimentation_* — the operational 1-moment column step: the 1-moment tendencies (Instantaneous,
 * or LinearizedAverage when nsub > 0) + the four sedimentation velocities + the host model's upwind flux divergence (formula above,
 * applied to q_lcl, q_icl, q_rai, q_sno).  PARITY UNPINNED FOR THE FLUX STEP, as for the 2-moment column step; pinned pieces:
 *   the tendencies     o_source_terms_1m / o_aggregate_1m / o_linearized_implicit_step_1m   (BMT:141-252, 269-465, 572-632; KATs
 *                      test/gpu_tests.jl:737-778)
 *   the fall speeds    o_sed_velocity_cloud_liquid / _cloud_ice / _snow_chen, o_terminal_velocity_rain_chen  (NonEq:250-281,
 *                      CM1:251-297; KATs test/gpu_tests.jl:627-630), on the clamped state of BMT:147-152.
 * scale = the Instantaneous tendency's own Σ|terms| + (S_{k+1} + S_k)/(ρ_k Δz_k), S = ρ χ w⁺ with w⁺ the fall speed evaluated with
 * `chen_ice_scale` (the Chen-2022 ice curves are E + F e^{−cD} with E ≈ −F near their zero crossing: the caller passes the table with
 * the negative amplitudes switched off; NULL: S = |F|).
 */
void FN(cmxo_mp1m_column_tendencies_sedimentation)(
    const TY(cmx_microphysics_1m) * mp, const TY(cmx_thermo) * tps, const TY(cmx_stokes_vel) * stokes, const TY(cmx_chen2022_rain_vel) * chen_rain,
    const TY(cmx_chen2022_ice_vel) * chen_ice, const TY(cmx_chen2022_ice_vel) * chen_ice_scale, uint32_t flags, const TY(cmxo_thresholds) * th,
    FT q_min, FT dt, int32_t nsub, int64_t n_col, int32_t n_lev, const FT *inv_dz, const FT *rho, const FT *T, const FT *q_tot, const FT *q_lcl,
    const FT *q_icl, const FT *q_rai, const FT *q_sno, FT *const *tend, FT *precip_rai, FT *precip_sno, FT *const *scale, int32_t nthreads) {
    const FT eps = th->eps_1m;
    const FT dt_sub = nsub > 0 ? dt / (FT)nsub : (FT)0;
    const FT Lv_over_cp = tps->LH_v0 / tps->cp_d, Ls_over_cp = tps->LH_s0 / tps->cp_d;
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
    for (int64_t col = 0; col < n_col; ++col) {
        FT F_up[4] = {0, 0, 0, 0}, S_up[4] = {0, 0, 0, 0};            /* F_{n_lev} = 0: nothing enters through the model top */
        for (int32_t k = n_lev - 1; k >= 0; --k) {
            const int64_t i = col * (int64_t)n_lev + k;
            /* (1) tendencies */
            TY(cmxo_src_1m) s = FN(o_source_terms_1m)(mp, tps, flags, th, rho[i], T[i], q_tot[i], q_lcl[i], q_icl[i], q_rai[i], q_sno[i]);
            FT t[4], sc[4];
            FN(o_aggregate_1m)(&s, t, sc);
            if (nsub > 0) {   /* BMT:572-632 */
                FT Ti = T[i], ql = q_lcl[i], qi = q_icl[i], qr = q_rai[i], qs = q_sno[i];
                for (int m = 0; m < nsub; ++m) {
                    FT r[4];
                    FN(o_linearized_implicit_step_1m)(mp, tps, flags, th, q_min, rho[i], Ti, q_tot[i], ql, qi, qr, qs, dt_sub, r);
                    ql += r[0] * dt_sub; qi += r[1] * dt_sub; qr += r[2] * dt_sub; qs += r[3] * dt_sub;
                    Ti += (Lv_over_cp * (r[0] + r[2]) + Ls_over_cp * (r[1] + r[3])) * dt_sub;
                }
                t[0] = (ql - q_lcl[i]) / dt; t[1] = (qi - q_icl[i]) / dt; t[2] = (qr - q_rai[i]) / dt; t[3] = (qs - q_sno[i]) / dt;
            }
            /* (2) fall speeds of the clamped state, (3) fluxes */
            const FT r = FN(o_max)((FT)0, rho[i]);
            const FT q[4] = {FN(o_max)((FT)0, q_lcl[i]), FN(o_max)((FT)0, q_icl[i]), FN(o_max)((FT)0, q_rai[i]), FN(o_max)((FT)0, q_sno[i])};
            FT w[4], wp[4];
            w[0] = FN(o_sed_velocity_cloud_liquid)(&mp->cloud_liquid, stokes, r, q[0], eps);
            w[1] = FN(o_sed_velocity_cloud_ice)(&mp->cloud_ice, &chen_ice->small_ice, r, q[1], eps);
            w[2] = FN(o_terminal_velocity_rain_chen)(&mp->rain, chen_rain, r, q[2], eps);
            w[3] = FN(o_sed_velocity_snow_chen)(&mp->snow, &chen_ice->large_ice, r, q[3], eps);
            wp[0] = w[0]; wp[2] = w[2];
            wp[1] = chen_ice_scale ? FN(o_sed_velocity_cloud_ice)(&mp->cloud_ice, &chen_ice_scale->small_ice, r, q[1], eps) : w[1];
            wp[3] = chen_ice_scale ? FN(o_sed_velocity_snow_chen)(&mp->snow, &chen_ice_scale->large_ice, r, q[3], eps) : w[3];
            const FT g = inv_dz[k] / r;
            for (int m = 0; m < 4; ++m) {
                const FT F = r * q[m] * w[m], S = r * q[m] * FN(o_max)(M_ABS(wp[m]), M_ABS(w[m]));
                tend[m][i] = t[m] + (F_up[m] - F) * g;
                if (scale && scale[m]) scale[m][i] = sc[m] + (S_up[m] + S) * g;
                if (k == 0 && m == 2 && precip_rai) precip_rai[col] = F;
                if (k == 0 && m == 3 && precip_sno) precip_sno[col] = F;
                F_up[m] = F; S_up[m] = S;
            }
        }
    }
}
