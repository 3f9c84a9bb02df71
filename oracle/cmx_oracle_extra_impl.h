/*
 * cmx_oracle_extra_impl.h — oracle (TEST INFRASTRUCTURE) for the remaining public functions of HetIceNucleation / Common that round 3
 * gave device entries (VERDICT r02 row g).  Included from cmx_oracle_impl.h.  Restates, operation by operation:
 *   CO.H2SO4_soln_saturation_vapor_pressure, CO.a_w_xT                         src/Common.jl:188-246
 *   CMI_het.dust_activated_number_fraction, MohlerDepositionRate, deposition_J  src/IceNucleation.jl:44-102
 *   (INP_concentration_frequency lives in cmx_oracle_p3col_impl.h)
 * Pinning: tests/golden/ice_nucleation_kats.json (test/gpu_tests.jl:876-984); the parameter values that no reference number fixes are
 * listed as "parity unpinned" in cmx/parameters.py.
 */
static inline FT FN(o_H2SO4_soln_saturation_vapor_pressure)(const TY(cmx_h2so4_solution_params) * p, FT x, FT T) {
    FT w_h = p->w_2 * x;
    return M_EXP(p->c1 - p->c2 * x + p->c3 * x * w_h - p->c4 * x * (w_h * w_h) + (p->c5 + p->c6 * x - p->c7 * x * w_h) / T) * 100;
}
void FN(cmxo_h2so4_solution)(const TY(cmx_h2so4_solution_params) * prs, const TY(cmx_thermo) * tps, int64_t n, const FT *x, const FT *T, FT *p_sol,
                            FT *a_w) {
    for (int64_t i = 0; i < n; ++i) {
        FT ps = FN(o_H2SO4_soln_saturation_vapor_pressure)(prs, x[i], T[i]);
        if (p_sol) p_sol[i] = ps;
        if (a_w) a_w[i] = ps / FN(o_psat_liquid)(tps, T[i]);          /* a_w_xT */
    }
}
/* the reference asserts S_i < Sᵢ_max: such points are NaN here (the device entry counts them) */
void FN(cmxo_mohler2006_deposition)(const TY(cmx_mohler_dust) * dust, const TY(cmx_mohler2006) * ip, int64_t n, const FT *S_i, const FT *T,
                                   const FT *dSi_dt, const FT *N_aer, FT *act_frac, FT *dep_rate) {
    for (int64_t i = 0; i < n; ++i) {
        int ok = S_i[i] < ip->S_i_max;
        FT S0 = T[i] > ip->T_thr ? dust->S0_warm : dust->S0_cold;
        FT a = T[i] > ip->T_thr ? dust->a_warm : dust->a_cold;
        if (act_frac) act_frac[i] = ok ? FN(o_max)((FT)0, M_EXP(a * (S_i[i] - S0)) - 1) : (FT)NAN;
        if (dep_rate) dep_rate[i] = ok ? FN(o_max)((FT)0, N_aer[i] * a * dSi_dt[i]) : (FT)NAN;
    }
}
void FN(cmxo_deposition_J)(const TY(cmx_deposition_dust) * dust, int64_t n, const FT *d, FT *J) {
    for (int64_t i = 0; i < n; ++i) J[i] = M_POW((FT)10, dust->deposition_m * d[i] + dust->deposition_c + 4);
}
