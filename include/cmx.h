/*
 * cmx.h — C ABI of the MI355X-native cloud-microphysics rate evaluator.
 *
 * This is the drop-in boundary.  The reference package
 * (CliMA/CloudMicrophysics.jl v0.38.1) has no FFI of its own: "array
 * evaluation" there is a Julia broadcast / KernelAbstractions kernel of the
 * form `output[i] = f(params, x[i]...)` over equal-length columns
 * (test/gpu_performance.jl:49-57, test/gpu_tests.jl:220-244,407-415,
 * test/type_stability_tests.jl:131-137).  Each entry point below replaces ONE
 * such broadcast with one fused HIP kernel for gfx950; the reference call it
 * replaces is cited next to it.  INTEGRATION.md shows the Julia `ccall`
 * binding a maintainer would add.
 *
 * Conventions
 *  - all array pointers are DEVICE pointers (HBM), structure-of-arrays, one
 *    column per state variable, `n` elements each, no aliasing between inputs
 *    and outputs; 16-byte alignment enables the 128-bit load/store path, any
 *    alignment is accepted;
 *  - parameter structs are plain host structs, passed by pointer, copied into
 *    the kernel argument segment (they end up in SGPRs: wave-uniform);
 *    field order == declaration order of the reference's immutable Julia
 *    structs (src/parameters/Microphysics2M.jl etc.) so an all-FT isbits
 *    Julia struct can be passed with `Ref(x)`;
 *  - `stream` is a `hipStream_t` passed as `void*` (NULL = the null stream);
 *    calls are asynchronous on that stream and never synchronise;
 *  - return value: 0 = ok, < 0 = error (cmx_status), the library never
 *    throws, never takes ownership of a pointer and keeps no device memory;
 *  - negative ρ, q and n inputs are clamped to 0 like UT.clamp_to_nonneg in the reference's entries (T is not clamped); a NaN in any
 *    input of a point gives NaN in every output of that point.  The air density must be positive: ρ ≤ 0 is outside the domain.  Such a
 *    point never yields a valid-looking result — the Float32 kernels reproduce the reference's own NaN / ±Inf pattern, the Float64 kernels
 *    return a non-finite value wherever the reference does and may return NaN for an output the reference still evaluates (their
 *    finite-argument elementary functions assume ρ > 0; the Float64 1-moment entries return NaN in EVERY output of such a point;
 *    tests/test_nan_inputs_gpu.py::test_zero_and_negative_air_density);
 *  - `_f32` entry points compute in float with the reference's Float32
 *    thresholds (eps(Float32), cbrt(floatmin(Float32)) — src/Utilities.jl:318-340),
 *    `_f64` ones in double with the Float64 thresholds.
 *
 * What this boundary deliberately does NOT have (SURVEY.md §8b proposed them; the decision is recorded here so
 * the header is the whole contract):
 *  - no `cmx_ctx` / `cmx_context_create`: the library is STATELESS.  The device is the calling thread's current HIP
 *    device, the stream is an argument, parameters are copied into the kernel arguments per call, nothing is
 *    cached between calls (the only process-wide state is a per-device properties cache filled once under
 *    std::call_once).  Every entry is therefore re-entrant, capturable into a HIP graph, and safe to call
 *    concurrently from several host threads on several streams or devices;
 *  - no `cmx_*_host` entry points taking host pointers: a host-pointer variant could only be a copy-in / copy-out
 *    wrapper (PCIe 63 GB/s against 52 B/point: ≈ 90× below the HBM-resident rate) or a CPU
 *    implementation, and a CPU path inside the product is ruled out (the CPU restatement lives in oracle/ as
 *    test infrastructure only).  A caller with host data does hipMemcpyAsync on its own stream;
 *  - no `cmx_multi_*` device-list entry points: the multi-GPU model is one PROCESS per GPU (torch.distributed /
 *    RCCL, cmx/sharding.py); each rank calls the single-device entries on its contiguous shard of every column
 *    and there is no data-path collective to hide behind an API.
 */
#ifndef CMX_H
#define CMX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* cmx_version() = (MAJOR << 16) | MINOR.  The minor number moves whenever a parameter struct changes its layout or an entry point is added, so a
 * binding compiled against another header can refuse to run:
 *   0.1  rounds 1–2
 *   0.3  round 3: cmx_process_params_1m gained the Frostenberg member (cmx_microphysics_1m: 85 → 90 fields); new structs cmx_mohler2006,
 *        cmx_mohler_dust, cmx_deposition_dust, cmx_h2so4_solution_params; new entries cmx_mp1m_column_tendencies_sedimentation,
 *        cmx_mp1m_linearized_average_fields, cmx_microphysics_2m_p3_tendencies_fields, cmx_ice_nucleation_rates_xT, cmx_h2so4_solution,
 *        cmx_mohler2006_deposition, cmx_deposition_J, cmx_inp_concentration_frequency, cmx_arg2000_total_activated, cmx_lean_eval_literal;
 *        process columns CMX_SB_DEVAP_DN_RAI / CMX_SB_DEVAP_DQ_RAI; CMX_1M_CLOUD_ICE_FORMATION_TDEP and every Chen-2022 table accepted
 *   0.4  round 4: no struct layout changed.  New entries cmx_gamma_inc, cmx_gamma_inc_inv (UT.gamma_inc / gamma_inc_inv over columns),
 *        cmx_generalized_gamma, cmx_exponential_distribution, cmx_sb2006_size_distribution (DistributionTools and the SB2006 PSD accessors); cmx_column_sums_* takes a caller-owned workspace and is deterministic (one launch
 *        for all columns + a one-workgroup-per-column finish, no floating-point atomics).
 *   0.5  round 5: no struct layout changed.  New entry cmx_cloud_diagnostics (CloudDiagnostics over columns); cmx_sb2006_size_distribution refuses the
 *        limited variant of a struct without limiters (CMX_ERR_BAD_ARG instead of silently clamping to 0).
 *        julia/CMXExt.jl (the reference-side binding) checks this number. */
#define CMX_VERSION_MAJOR 0
#define CMX_VERSION_MINOR 5

typedef enum cmx_status {
    CMX_OK = 0,
    CMX_ERR_BAD_ARG = -1,     /* null pointer, n < 0, unknown flag combination */
    CMX_ERR_HIP = -2,         /* a HIP runtime call failed: see cmx_last_hip_error() */
    CMX_ERR_UNSUPPORTED = -3  /* valid request this build has no kernel for */
} cmx_status;

/* ---- domain of the Chen-2022 RAIN fall speeds (every entry that takes a cmx_chen2022_rain_vel_* table: cmx_sb2006_warm_rain_tendencies_*,
 * cmx_sb2006_process_rates_*, cmx_sb2006_column_tendencies_sedimentation_* with CMX_VEL_CHEN2022; cmx_mp1m_terminal_velocity_*,
 * cmx_sedimentation_velocities_*, cmx_mp1m_column_tendencies_sedimentation_*; the vel_rain member of cmx_p3_ice_params_*) ---------------
 * The table's exponents depend on the state only through the air density, b_i(ρ) = b_i − b_ρ ρ (src/Common.jl:290-302), and the reference
 * evaluates Γ(b_i(ρ) + 1) with SpecialFunctions at any ρ.  The library fits those Γ per parameter set on 0 ≤ ρ ≤ 2 kg/m³ (csrc/cmx_math.hpp
 * make_chen_gamma; the fit is checked against tgamma on the host and a table it cannot represent takes a general-Γ instantiation, valid
 * for every ρ).  With a fitted table, a point with ρ > 2 kg/m³ gets NaN rain fall speeds — and, in the column entries, NaN sedimentation
 * tendencies of that cell and the cell below — rather than an extrapolated value.  ρ ≤ 2 kg/m³ covers every terrestrial state
 * (ρ = 1.4 kg/m³ at 1050 hPa and 260 K); callers with denser gases must not use the Chen-2022 rain table.  The SB2006 fall speeds
 * (CMX_VEL_SB2006) and the Chen-2022 ICE tables have no such restriction. */

/* ---- flags for the SB2006 entry points ------------------------------------ */
/* rain PSD with the SB2006 Eq. 94-97 limiters (RainParticlePDF_SB2006_limited,
 * src/parameters/Microphysics2M.jl:314-335) vs. without (…_notlimited, :362-375) */
#define CMX_SB2006_LIMITED      (1u << 0)
/* rain terminal-velocity scheme for the two optional velocity columns */
#define CMX_VEL_SB2006          (1u << 1)  /* CM2.rain_terminal_velocity(::SB2006VelType)  src/Microphysics2M.jl:685-702 */
#define CMX_VEL_CHEN2022        (1u << 2)  /* CM2.rain_terminal_velocity(::Chen2022VelTypeRain) :703-719 */

/* ---------------------------------------------------------------------------
 * Parameter structs.  One macro stamps the Float32 and the Float64 family.
 * Field names are ASCII transliterations of the Julia field names
 * (νc → nu_c, ρw → rho_w, λ_min → lambda_min, κrr → kappa_rr, τ → tau …).
 * ------------------------------------------------------------------------- */
#define CMX_DECLARE_PARAM_STRUCTS(FT, SFX)                                                     \
    /* CloudParticlePDF_SB2006 — src/parameters/Microphysics2M.jl:401-416 */                   \
    typedef struct cmx_cloud_pdf_sb2006_##SFX {                                                \
        FT nu_c, mu_c, xc_min, xc_max, rho_w, loggamma_z1, loggamma_z2;                        \
    } cmx_cloud_pdf_sb2006_##SFX;                                                              \
    /* RainParticlePDF_SB2006_limited — :314-335 (the not-limited variant, :362-375, uses   */ \
    /* nu_r, mu_r, xr_min, xr_max, rho_w, rho_0 only; its N0/lambda fields are ignored)      */ \
    typedef struct cmx_rain_pdf_sb2006_##SFX {                                                 \
        FT nu_r, mu_r, xr_min, xr_max, N0_min, N0_max, lambda_min, lambda_max, rho_w, rho_0;   \
    } cmx_rain_pdf_sb2006_##SFX;                                                               \
    /* AcnvSB2006 — :443-456 */                                                                \
    typedef struct cmx_acnv_sb2006_##SFX { FT kcc, x_star, rho_0, A, a, b; }                   \
        cmx_acnv_sb2006_##SFX;                                                                 \
    /* AccrSB2006 — :480-489 */                                                                \
    typedef struct cmx_accr_sb2006_##SFX { FT kcr, tau_0, rho_0, c; } cmx_accr_sb2006_##SFX;   \
    /* SelfColSB2006 — :510-517 */                                                             \
    typedef struct cmx_selfcol_sb2006_##SFX { FT krr, kappa_rr, d; } cmx_selfcol_sb2006_##SFX; \
    /* BreakupSB2006 — :537-546 */                                                             \
    typedef struct cmx_breakup_sb2006_##SFX { FT Deq, Dr_th, kbr, kappa_br; }                  \
        cmx_breakup_sb2006_##SFX;                                                              \
    /* EvaporationSB2006 — :567-588; the last five are host-derived (:599-606) */              \
    typedef struct cmx_evap_sb2006_##SFX {                                                     \
        FT av, bv, alpha, beta, rho_0, a_vent_1, b_vent_1, a_vent_0_coeff, b_vent_0_coeff,     \
            beta_vent_0;                                                                       \
    } cmx_evap_sb2006_##SFX;                                                                   \
    /* NumberAdjustmentHorn2012 — :617-620 */                                                  \
    typedef struct cmx_numadj_horn2012_##SFX { FT tau; } cmx_numadj_horn2012_##SFX;            \
    /* SB2006 — :642-659 */                                                                    \
    typedef struct cmx_sb2006_##SFX {                                                          \
        cmx_cloud_pdf_sb2006_##SFX pdf_c;                                                      \
        cmx_rain_pdf_sb2006_##SFX pdf_r;                                                       \
        cmx_acnv_sb2006_##SFX acnv;                                                            \
        cmx_accr_sb2006_##SFX accr;                                                            \
        cmx_selfcol_sb2006_##SFX self;                                                         \
        cmx_breakup_sb2006_##SFX brek;                                                         \
        cmx_evap_sb2006_##SFX evap;                                                            \
        cmx_numadj_horn2012_##SFX numadj;                                                      \
    } cmx_sb2006_##SFX;                                                                        \
    /* AirProperties — src/parameters/AirProperties.jl:11-18 */                                \
    typedef struct cmx_air_properties_##SFX { FT K_therm, D_vapor, nu_air; }                   \
        cmx_air_properties_##SFX;                                                              \
    /* WarmRainParams2M — src/parameters/Microphysics2MParams.jl:14-19                      */ \
    /* (seifert_beheng, air_properties, condevap.τ_relax, subdep.τ_relax)                   */ \
    typedef struct cmx_warm_rain_2m_##SFX {                                                    \
        cmx_sb2006_##SFX seifert_beheng;                                                       \
        cmx_air_properties_##SFX air_properties;                                               \
        FT condevap_tau_relax;                                                                 \
        FT subdep_tau_relax;                                                                   \
    } cmx_warm_rain_2m_##SFX;                                                                  \
    /* Thermodynamics.Parameters.ThermodynamicsParameters is NOT the reference's struct   */   \
    /* (un-vendored Thermodynamics.jl); the shim flattens it through the accessors the    */   \
    /* reference itself uses (src/ThermodynamicsInterface.jl:9-25).                        */  \
    typedef struct cmx_thermo_##SFX {                                                          \
        FT R_v, R_d, cp_d, cp_v, cp_l, cp_i, LH_v0, LH_s0, T_0, T_triple, press_triple,        \
            T_freeze, cv_l;                                                                    \
    } cmx_thermo_##SFX;                                                                        \
    /* KK2000 / B1994 / TC1980 / LD2004 bulk two-moment autoconversion and accretion parameter */ \
    /* sets — src/parameters/Microphysics2M.jl:11-75, 89-160, 172-240, 258-279                 */ \
    typedef struct cmx_kk2000_##SFX { FT acnv_A, acnv_a, acnv_b, acnv_c, accr_A, accr_a, accr_b; } cmx_kk2000_##SFX; \
    typedef struct cmx_b1994_##SFX {                                                           \
        FT acnv_C, acnv_a, acnv_b, acnv_c, acnv_N_0, acnv_d_low, acnv_d_high, acnv_k, accr_A;  \
    } cmx_b1994_##SFX;                                                                         \
    typedef struct cmx_tc1980_##SFX {                                                          \
        FT acnv_a, acnv_b, acnv_D, acnv_r_0, acnv_me_liq, acnv_m0_liq_coeff, acnv_k, accr_A;   \
    } cmx_tc1980_##SFX;                                                                        \
    typedef struct cmx_ld2004_##SFX { FT R_6C_0, E_0, rho_w, k; } cmx_ld2004_##SFX;            \
    typedef struct cmx_bulk_2m_schemes_##SFX {                                                 \
        cmx_kk2000_##SFX kk2000; cmx_b1994_##SFX b1994; cmx_tc1980_##SFX tc1980;               \
        cmx_ld2004_##SFX ld2004;                                                               \
    } cmx_bulk_2m_schemes_##SFX;                                                               \
    /* StokesRegimeVelType — src/parameters/TerminalVelocity.jl:150-154 */                     \
    typedef struct cmx_stokes_vel_##SFX { FT rho_w, nu_air, grav; } cmx_stokes_vel_##SFX;      \
    /* SB2006VelType — src/parameters/TerminalVelocity.jl:174-182 */                           \
    typedef struct cmx_sb2006_vel_##SFX { FT rho_0, aR, bR, cR, rho_w, nu_air, grav; }         \
        cmx_sb2006_vel_##SFX;                                                                  \
    /* Chen2022VelTypeRain — src/parameters/TerminalVelocity.jl:288-295 (Table B1) */          \
    typedef struct cmx_chen2022_rain_vel_##SFX {                                               \
        FT rho_0, a[3], a3_pow, b[3], b_rho, c[3];                                             \
    } cmx_chen2022_rain_vel_##SFX;                                                             \
    /* terminal-velocity parameters for the optional velocity columns: which member is    */   \
    /* read is selected by CMX_VEL_SB2006 / CMX_VEL_CHEN2022                              */   \
    typedef struct cmx_rain_vel_##SFX {                                                        \
        cmx_sb2006_vel_##SFX sb2006;                                                           \
        cmx_chen2022_rain_vel_##SFX chen2022;                                                  \
    } cmx_rain_vel_##SFX;                                                                      \
    /* Koop2000 — src/parameters/IceNucleation.jl:38-55 */                                     \
    typedef struct cmx_koop2000_##SFX {                                                        \
        FT delta_a_w_min, delta_a_w_max, c1, c2, c3, c4, linear_c1, linear_c2;                 \
    } cmx_koop2000_##SFX;                                                                      \
    /* the two ABIFM fields of a dust type (Kaolinite, Illite, DesertDust, …:                */ \
    /* src/parameters/AerosolKaolinite.jl:19-21 etc.): log10 J = m·Δa_w + c  [cm⁻² s⁻¹]      */ \
    typedef struct cmx_abifm_dust_##SFX { FT ABIFM_m, ABIFM_c; } cmx_abifm_dust_##SFX;          \
    /* Mohler2006 (Sᵢ_max, T_thr) — src/parameters/IceNucleation.jl:13-18; the four deposition fields of DesertDust /       */ \
    /* ArizonaTestDust (S₀_warm, S₀_cold, a_warm, a_cold) — src/parameters/AerosolDesertDust.jl:13-21, AerosolATD.jl:12-20   */ \
    typedef struct cmx_mohler2006_##SFX { FT S_i_max, T_thr; } cmx_mohler2006_##SFX;           \
    typedef struct cmx_mohler_dust_##SFX { FT S0_warm, S0_cold, a_warm, a_cold; } cmx_mohler_dust_##SFX; \
    /* the two deposition fields of a mineral (Kaolinite, Feldspar, Ferrihydrite, …: src/parameters/AerosolKaolinite.jl etc.): */ \
    /* log10 J = m·Δa_w + c  [cm⁻² s⁻¹]                                                                                   */ \
    typedef struct cmx_deposition_dust_##SFX { FT deposition_m, deposition_c; } cmx_deposition_dust_##SFX; \
    /* H2SO4SolutionParameters — src/parameters/Aerosol_H2SO4_Solution.jl (Luo et al. 1995) */  \
    typedef struct cmx_h2so4_solution_params_##SFX { FT T_max, T_min, w_2, c1, c2, c3, c4, c5, c6, c7; } cmx_h2so4_solution_params_##SFX; \
    /* ---- 1-moment scheme: src/parameters/Microphysics1M.jl ---------------------------- */   \
    /* ParticleMass — m(r) = m0 χm (r/r0)^(me+Δm); gamma_coeff = Γ(me+Δm+1) host-derived */     \
    typedef struct cmx_particle_mass_##SFX { FT r0, m0, me, delta_m, chi_m, gamma_coeff; }     \
        cmx_particle_mass_##SFX;                                                               \
    /* ParticleArea — a(r) = a0 χa (r/r0)^(ae+Δa) */                                           \
    typedef struct cmx_particle_area_##SFX { FT a0, ae, delta_a, chi_a; }                      \
        cmx_particle_area_##SFX;                                                               \
    typedef struct cmx_ventilation_##SFX { FT a, b; } cmx_ventilation_##SFX;                   \
    /* Acnv1M (τ, q_threshold, k) and VarTimescaleAcnv (τ, α, Nc) */                           \
    typedef struct cmx_acnv_1m_##SFX { FT tau, q_threshold, k; } cmx_acnv_1m_##SFX;            \
    typedef struct cmx_var_timescale_acnv_##SFX { FT tau, alpha, Nc; }                         \
        cmx_var_timescale_acnv_##SFX;                                                          \
    /* CloudLiquid{ρw, r_eff, N_0}; CloudIce{pdf(n0), mass, ρᵢ, r_eff, N_0} */                 \
    typedef struct cmx_cloud_liquid_##SFX { FT rho_w, r_eff, N_0; } cmx_cloud_liquid_##SFX;    \
    typedef struct cmx_cloud_ice_##SFX {                                                       \
        FT n0; cmx_particle_mass_##SFX mass; FT rho_i, r_eff, N_0;                             \
    } cmx_cloud_ice_##SFX;                                                                     \
    /* Rain{pdf(n0), mass, area, vent} */                                                      \
    typedef struct cmx_rain_##SFX {                                                            \
        FT n0; cmx_particle_mass_##SFX mass; cmx_particle_area_##SFX area;                     \
        cmx_ventilation_##SFX vent;                                                            \
    } cmx_rain_##SFX;                                                                          \
    /* Snow{pdf(μ, ν), mass, area, vent, aspr(ϕ, κ), ρᵢ, gamma_aspect_oblate/prolate} */       \
    typedef struct cmx_snow_##SFX {                                                            \
        FT mu, nu; cmx_particle_mass_##SFX mass; cmx_particle_area_##SFX area;                 \
        cmx_ventilation_##SFX vent; FT phi, kappa, rho_i, gamma_aspect_oblate,                 \
            gamma_aspect_prolate;                                                              \
    } cmx_snow_##SFX;                                                                          \
    /* Blk1MVelTypeRain / Blk1MVelTypeSnow — src/parameters/TerminalVelocity.jl:12-30,76-85 */ \
    typedef struct cmx_blk1m_vel_rain_##SFX {                                                  \
        FT r0, ve, delta_v, chi_v, rho_w, C_drag, grav, gamma_vent, gamma_term, gamma_accr,    \
            gamma_accr_rain_sink;                                                              \
    } cmx_blk1m_vel_rain_##SFX;                                                                \
    typedef struct cmx_blk1m_vel_snow_##SFX {                                                  \
        FT r0, ve, delta_v, chi_v, v0, gamma_vent, gamma_term, gamma_accr;                     \
    } cmx_blk1m_vel_snow_##SFX;                                                                \
    /* Frostenberg2023 — src/parameters/IceNucleation.jl:171-193 (log_a = log(a), host-derived) */ \
    typedef struct cmx_frostenberg2023_##SFX { FT sigma, a, b, T_freeze, log_a; } cmx_frostenberg2023_##SFX; \
    /* process_params of Microphysics1MParams (src/parameters/Microphysics1MOptions.jl:296-   */ \
    /* 395), flattened: every variant's parameters are present, `flags` says which are read */  \
    typedef struct cmx_process_params_1m_##SFX {                                               \
        FT cloud_liquid_formation_tau_relax, cloud_ice_formation_tau_relax;                    \
        cmx_frostenberg2023_##SFX cloud_ice_formation_frostenberg; /* TemperatureDependent :314-318 */ \
        cmx_acnv_1m_##SFX rain_autoconversion;             /* Kessler1M          */            \
        cmx_var_timescale_acnv_##SFX rain_autoconversion_nd; /* PrescribedNd     */            \
        cmx_acnv_1m_##SFX snow_autoconversion;             /* NoSupersaturation  */            \
        FT r_ice_snow;                                     /* WithSupersaturation */           \
        FT e_lcl_rai, e_lcl_sno, e_icl_rai, e_icl_sno, e_rai_sno, coeff_disp;                  \
    } cmx_process_params_1m_##SFX;                                                             \
    /* Microphysics1MParams — src/parameters/Microphysics1MParams.jl:63-70 */                  \
    typedef struct cmx_microphysics_1m_##SFX {                                                 \
        cmx_process_params_1m_##SFX process_params;                                            \
        cmx_cloud_liquid_##SFX cloud_liquid;                                                   \
        cmx_cloud_ice_##SFX cloud_ice;                                                         \
        cmx_rain_##SFX rain;                                                                   \
        cmx_snow_##SFX snow;                                                                   \
        cmx_air_properties_##SFX air_properties;                                               \
        cmx_blk1m_vel_rain_##SFX vel_rain;                                                     \
        cmx_blk1m_vel_snow_##SFX vel_snow;                                                     \
    } cmx_microphysics_1m_##SFX;                                                               \
    /* AerosolActivationParameters — src/parameters/AerosolActivation.jl:12-37 */              \
    typedef struct cmx_aerosol_activation_params_##SFX {                                       \
        FT M_w, R, rho_w, rho_i, sigma, g, f1, f2, g1, g2, p1, p2;                             \
    } cmx_aerosol_activation_params_##SFX;                                                     \
    /* One lognormal mode of an AerosolDistribution (src/AerosolModel.jl:26-100), reduced to */ \
    /* what the activation needs: the component tuples of Mode_B / Mode_κ enter only through */ \
    /* mean_hygroscopicity_parameter (src/AerosolActivation.jl:55-95) and Σ M_j·w_j (:313),  */ \
    /* both evaluated on the host.                                                           */ \
    typedef struct cmx_aerosol_mode_##SFX {                                                    \
        FT r_dry, stdev, N, hygroscopicity, molar_mass_mix;                                    \
    } cmx_aerosol_mode_##SFX;                                                                  \
    typedef struct cmx_aerosol_distribution_##SFX {                                            \
        int32_t n_modes; int32_t pad_;                                                         \
        cmx_aerosol_mode_##SFX modes[CMX_ARG_MAX_MODES];                                       \
    } cmx_aerosol_distribution_##SFX;                                                          \
    /* ParametersP3 (src/parameters/MicrophysicsP3.jl:267-286), the fields the shape solver  */ \
    /* reads: MassPowerLaw (α_va, β_va :26-31), AreaPowerLaw (γ, σ :60-65), SlopePowerLaw    */ \
    /* (a, b, c, μ_max :104-113) or SlopeConstant (μ :139-142), ρ_i, ρ_l                      */ \
    typedef struct cmx_p3_params_##SFX {                                                       \
        FT alpha_va, beta_va, gamma, sigma, slope_a, slope_b, slope_c, mu_max, mu_const,       \
            rho_i, rho_l, tau_wet, T_freeze;                                                   \
    } cmx_p3_params_##SFX;                                                                     \
    /* CMP.Chen2022VelTypeSmallIce / LargeIce — src/parameters/TerminalVelocity.jl:207-216,    */ \
    /* 247-257 (Chen et al. 2022 tables B3 / B5 + the small/large cutoff dimension)            */ \
    typedef struct cmx_chen2022_small_ice_vel_##SFX {                                          \
        FT A[3], B[3], C[4], E[3], F[3], G[3], cutoff;                                         \
    } cmx_chen2022_small_ice_vel_##SFX;                                                        \
    typedef struct cmx_chen2022_large_ice_vel_##SFX {                                          \
        FT A[3], B[3], C[3], E[3], F[3], G[3], H[3], cutoff;                                   \
    } cmx_chen2022_large_ice_vel_##SFX;                                                        \
    typedef struct cmx_chen2022_ice_vel_##SFX {                                                \
        cmx_chen2022_small_ice_vel_##SFX small_ice;                                            \
        cmx_chen2022_large_ice_vel_##SFX large_ice;                                            \
    } cmx_chen2022_ice_vel_##SFX;                                                              \
    /* Quadrature.ChebyshevGauss(n) / GaussLegendre(FT, n) — src/Quadrature.jl:168-175,226-252: */ \
    /* n nodes yᵢ on [-1, 1] and TOTAL weights inv_weight_fun(yᵢ)·weight(i) (host-built once,   */ \
    /* like the reference builds GaussLegendre host-side and ships it as an isbits struct)      */ \
    typedef struct cmx_quadrature_##SFX {                                                      \
        int32_t n, reserved;                                                                   \
        FT node[CMX_QUAD_MAX], weight[CMX_QUAD_MAX];                                           \
    } cmx_quadrature_##SFX;                                                                    \
    /* Parameters0M — src/parameters/Microphysics0M.jl:12-28 */                                \
    typedef struct cmx_parameters_0m_##SFX { FT tau_precip, qc_0, S_0; } cmx_parameters_0m_##SFX; \
    /* LocalRimeDensity — src/parameters/MicrophysicsP3.jl:202-239 (Cober & List 1993 Eq. 16-17) */ \
    typedef struct cmx_local_rime_density_##SFX { FT a, b, c, rho_ice; } cmx_local_rime_density_##SFX; \
    /* RainFreezing (Bigg 1953 / Barklie–Gokhale 1959) — src/parameters/IceNucleation.jl:129-146 */ \
    typedef struct cmx_rain_freezing_##SFX { FT het_a, het_B; } cmx_rain_freezing_##SFX;       \
    /* MorrisonMilbrandt2014 — src/parameters/IceNucleation.jl:80-107 */                       \
    typedef struct cmx_morrison_milbrandt2014_##SFX { FT T_dep_thres, c1, c2, T0, het_a, het_B; } \
        cmx_morrison_milbrandt2014_##SFX;                                                      \
    /* P3IceParams — src/parameters/Microphysics2MParams.jl:58-86: scheme (+ its vent and      */ \
    /* ρ_rim_local members, src/parameters/MicrophysicsP3.jl:267-288), terminal_velocity       */ \
    /* (Chen2022VelType: rain, small_ice, large_ice), cloud_pdf, rain_pdf, ice_nucleation,     */ \
    /* rain_freezing, inp_depletion_model.τ_act, quad                                          */ \
    typedef struct cmx_p3_ice_params_##SFX {                                                   \
        cmx_p3_params_##SFX scheme;                                                            \
        cmx_ventilation_##SFX vent;                                                            \
        cmx_local_rime_density_##SFX rho_rim_local;                                            \
        cmx_chen2022_rain_vel_##SFX vel_rain;                                                  \
        cmx_chen2022_ice_vel_##SFX vel_ice;                                                    \
        cmx_cloud_pdf_sb2006_##SFX cloud_pdf;                                                  \
        cmx_rain_pdf_sb2006_##SFX rain_pdf;                                                    \
        cmx_frostenberg2023_##SFX ice_nucleation;                                              \
        cmx_rain_freezing_##SFX rain_freezing;                                                 \
        FT tau_act;                                                                            \
        cmx_quadrature_##SFX quad;                                                             \
    } cmx_p3_ice_params_##SFX;

#define CMX_ARG_MAX_MODES 8
#define CMX_QUAD_MAX 128

CMX_DECLARE_PARAM_STRUCTS(float, f32)
CMX_DECLARE_PARAM_STRUCTS(double, f64)

/* ---------------------------------------------------------------------------
 * Size contract of every parameter struct: field count × sizeof(FT) (+ the 8-byte integer header of the two structs
 * that carry a count).  No struct has padding — every member is an FT or an array / struct of FT — so a binding that
 * declares the same fields in the same order (INTEGRATION.md §2 quotes these counts) has the same layout; a binding
 * with a missing field (e.g. the 13th member `cv_l` of cmx_thermo_*) has a different sizeof and must not compile.
 * tests/test_abi.py checks the same numbers from the Python side, tests/native/abi_caller.c from plain C.
 * ------------------------------------------------------------------------- */
#if defined(__cplusplus)
#define CMX_STATIC_ASSERT(cond, msg) static_assert(cond, msg)
#else
#define CMX_STATIC_ASSERT(cond, msg) _Static_assert(cond, msg)
#endif
#define CMX_ASSERT_PARAM_STRUCT_SIZES(FT, SFX) \
    CMX_STATIC_ASSERT(sizeof(cmx_abifm_dust_##SFX) == 2 * sizeof(FT), "cmx_abifm_dust");\
    CMX_STATIC_ASSERT(sizeof(cmx_mohler2006_##SFX) == 2 * sizeof(FT), "cmx_mohler2006");\
    CMX_STATIC_ASSERT(sizeof(cmx_mohler_dust_##SFX) == 4 * sizeof(FT), "cmx_mohler_dust");\
    CMX_STATIC_ASSERT(sizeof(cmx_deposition_dust_##SFX) == 2 * sizeof(FT), "cmx_deposition_dust");\
    CMX_STATIC_ASSERT(sizeof(cmx_h2so4_solution_params_##SFX) == 10 * sizeof(FT), "cmx_h2so4_solution_params");\
    CMX_STATIC_ASSERT(sizeof(cmx_accr_sb2006_##SFX) == 4 * sizeof(FT), "cmx_accr_sb2006");\
    CMX_STATIC_ASSERT(sizeof(cmx_acnv_1m_##SFX) == 3 * sizeof(FT), "cmx_acnv_1m");\
    CMX_STATIC_ASSERT(sizeof(cmx_acnv_sb2006_##SFX) == 6 * sizeof(FT), "cmx_acnv_sb2006");\
    CMX_STATIC_ASSERT(sizeof(cmx_aerosol_activation_params_##SFX) == 12 * sizeof(FT), "cmx_aerosol_activation_params");\
    CMX_STATIC_ASSERT(sizeof(cmx_aerosol_distribution_##SFX) == 8 + 40 * sizeof(FT), "cmx_aerosol_distribution");\
    CMX_STATIC_ASSERT(sizeof(cmx_aerosol_mode_##SFX) == 5 * sizeof(FT), "cmx_aerosol_mode");\
    CMX_STATIC_ASSERT(sizeof(cmx_air_properties_##SFX) == 3 * sizeof(FT), "cmx_air_properties");\
    CMX_STATIC_ASSERT(sizeof(cmx_b1994_##SFX) == 9 * sizeof(FT), "cmx_b1994");\
    CMX_STATIC_ASSERT(sizeof(cmx_blk1m_vel_rain_##SFX) == 11 * sizeof(FT), "cmx_blk1m_vel_rain");\
    CMX_STATIC_ASSERT(sizeof(cmx_blk1m_vel_snow_##SFX) == 8 * sizeof(FT), "cmx_blk1m_vel_snow");\
    CMX_STATIC_ASSERT(sizeof(cmx_breakup_sb2006_##SFX) == 4 * sizeof(FT), "cmx_breakup_sb2006");\
    CMX_STATIC_ASSERT(sizeof(cmx_bulk_2m_schemes_##SFX) == 28 * sizeof(FT), "cmx_bulk_2m_schemes");\
    CMX_STATIC_ASSERT(sizeof(cmx_chen2022_ice_vel_##SFX) == 42 * sizeof(FT), "cmx_chen2022_ice_vel");\
    CMX_STATIC_ASSERT(sizeof(cmx_chen2022_large_ice_vel_##SFX) == 22 * sizeof(FT), "cmx_chen2022_large_ice_vel");\
    CMX_STATIC_ASSERT(sizeof(cmx_chen2022_rain_vel_##SFX) == 12 * sizeof(FT), "cmx_chen2022_rain_vel");\
    CMX_STATIC_ASSERT(sizeof(cmx_chen2022_small_ice_vel_##SFX) == 20 * sizeof(FT), "cmx_chen2022_small_ice_vel");\
    CMX_STATIC_ASSERT(sizeof(cmx_cloud_ice_##SFX) == 10 * sizeof(FT), "cmx_cloud_ice");\
    CMX_STATIC_ASSERT(sizeof(cmx_cloud_liquid_##SFX) == 3 * sizeof(FT), "cmx_cloud_liquid");\
    CMX_STATIC_ASSERT(sizeof(cmx_cloud_pdf_sb2006_##SFX) == 7 * sizeof(FT), "cmx_cloud_pdf_sb2006");\
    CMX_STATIC_ASSERT(sizeof(cmx_evap_sb2006_##SFX) == 10 * sizeof(FT), "cmx_evap_sb2006");\
    CMX_STATIC_ASSERT(sizeof(cmx_frostenberg2023_##SFX) == 5 * sizeof(FT), "cmx_frostenberg2023");\
    CMX_STATIC_ASSERT(sizeof(cmx_kk2000_##SFX) == 7 * sizeof(FT), "cmx_kk2000");\
    CMX_STATIC_ASSERT(sizeof(cmx_koop2000_##SFX) == 8 * sizeof(FT), "cmx_koop2000");\
    CMX_STATIC_ASSERT(sizeof(cmx_ld2004_##SFX) == 4 * sizeof(FT), "cmx_ld2004");\
    CMX_STATIC_ASSERT(sizeof(cmx_local_rime_density_##SFX) == 4 * sizeof(FT), "cmx_local_rime_density");\
    CMX_STATIC_ASSERT(sizeof(cmx_microphysics_1m_##SFX) == 90 * sizeof(FT), "cmx_microphysics_1m");\
    CMX_STATIC_ASSERT(sizeof(cmx_morrison_milbrandt2014_##SFX) == 6 * sizeof(FT), "cmx_morrison_milbrandt2014");\
    CMX_STATIC_ASSERT(sizeof(cmx_numadj_horn2012_##SFX) == 1 * sizeof(FT), "cmx_numadj_horn2012");\
    CMX_STATIC_ASSERT(sizeof(cmx_p3_ice_params_##SFX) == 8 + 354 * sizeof(FT), "cmx_p3_ice_params");\
    CMX_STATIC_ASSERT(sizeof(cmx_p3_params_##SFX) == 13 * sizeof(FT), "cmx_p3_params");\
    CMX_STATIC_ASSERT(sizeof(cmx_parameters_0m_##SFX) == 3 * sizeof(FT), "cmx_parameters_0m");\
    CMX_STATIC_ASSERT(sizeof(cmx_particle_area_##SFX) == 4 * sizeof(FT), "cmx_particle_area");\
    CMX_STATIC_ASSERT(sizeof(cmx_particle_mass_##SFX) == 6 * sizeof(FT), "cmx_particle_mass");\
    CMX_STATIC_ASSERT(sizeof(cmx_process_params_1m_##SFX) == 23 * sizeof(FT), "cmx_process_params_1m");\
    CMX_STATIC_ASSERT(sizeof(cmx_quadrature_##SFX) == 8 + 256 * sizeof(FT), "cmx_quadrature");\
    CMX_STATIC_ASSERT(sizeof(cmx_rain_##SFX) == 13 * sizeof(FT), "cmx_rain");\
    CMX_STATIC_ASSERT(sizeof(cmx_rain_freezing_##SFX) == 2 * sizeof(FT), "cmx_rain_freezing");\
    CMX_STATIC_ASSERT(sizeof(cmx_rain_pdf_sb2006_##SFX) == 10 * sizeof(FT), "cmx_rain_pdf_sb2006");\
    CMX_STATIC_ASSERT(sizeof(cmx_rain_vel_##SFX) == 19 * sizeof(FT), "cmx_rain_vel");\
    CMX_STATIC_ASSERT(sizeof(cmx_sb2006_##SFX) == 45 * sizeof(FT), "cmx_sb2006");\
    CMX_STATIC_ASSERT(sizeof(cmx_sb2006_vel_##SFX) == 7 * sizeof(FT), "cmx_sb2006_vel");\
    CMX_STATIC_ASSERT(sizeof(cmx_selfcol_sb2006_##SFX) == 3 * sizeof(FT), "cmx_selfcol_sb2006");\
    CMX_STATIC_ASSERT(sizeof(cmx_snow_##SFX) == 19 * sizeof(FT), "cmx_snow");\
    CMX_STATIC_ASSERT(sizeof(cmx_stokes_vel_##SFX) == 3 * sizeof(FT), "cmx_stokes_vel");\
    CMX_STATIC_ASSERT(sizeof(cmx_tc1980_##SFX) == 8 * sizeof(FT), "cmx_tc1980");\
    CMX_STATIC_ASSERT(sizeof(cmx_thermo_##SFX) == 13 * sizeof(FT), "cmx_thermo");\
    CMX_STATIC_ASSERT(sizeof(cmx_var_timescale_acnv_##SFX) == 3 * sizeof(FT), "cmx_var_timescale_acnv");\
    CMX_STATIC_ASSERT(sizeof(cmx_ventilation_##SFX) == 2 * sizeof(FT), "cmx_ventilation");\
    CMX_STATIC_ASSERT(sizeof(cmx_warm_rain_2m_##SFX) == 50 * sizeof(FT), "cmx_warm_rain_2m");

CMX_ASSERT_PARAM_STRUCT_SIZES(float, f32)
CMX_ASSERT_PARAM_STRUCT_SIZES(double, f64)

/* ---------------------------------------------------------------------------
 * Library / device queries
 * ------------------------------------------------------------------------- */
/* (major << 16) | minor */
int32_t cmx_version(void);
/* text of the last HIP error seen by this thread ("" if none); static storage */
const char *cmx_last_hip_error(void);

/* Diagnostic entry (no reference counterpart): the library's own Float64 elementary functions (csrc/cmx_lean_f64.hpp — gfx950 has
 * no Float64 transcendental unit) evaluated over a device column, so that a test can measure them against libm in ulps ON the
 * device.  which: 0 exp2, 1 log2, 2 exp, 3 log, 4 rcp, 5 sqrt, 6 rsqrt, 7 expm1, 8 log1p, 9 erfc (the table-driven form of the ARG kernel),
 * 10 lgamma for z > 0 (the P3 shape solver's); 11-17 the finite-argument forms exp2_fin, exp_fin, rcp_finite, rcp_nz, sqrt_pos, rsqrt_pos,
 * pow_m34_pos and 18 log_pos (positive normal finite arguments only: DESIGN.md section 4.3); 19 the identity (the kernel's own instructions:
 * tools/f64_floor.py subtracts them from the per-dispatch instruction counters of the others). */
int32_t cmx_lean_eval_f64(int32_t which, int64_t n, const double *x, double *y, void *stream);
/* … and the same functions as compiled into the production Float64 kernels' translation units (polynomial coefficients as SGPR literals
 * instead of LDS reads; csrc/Makefile LITCOEF). */
int32_t cmx_lean_eval_literal_f64(int32_t which, int64_t n, const double *x, double *y, void *stream);

/* ---------------------------------------------------------------------------
 * (1) North star — SB2006 two-moment warm-rain fused tendencies.
 *
 * Replaces the broadcast
 *   BMT.bulk_microphysics_tendencies.(Ref(BMT.Microphysics2Moment()), Ref(mp), Ref(tps),
 *                                     ρ, T, q_tot, q_lcl, n_lcl, q_rai, n_rai)
 * src/BulkMicrophysicsTendencies.jl:820-854 → warm_rain_tendencies_2m :707-782
 * (KA wrapper: test/gpu_performance.jl:49-57), plus — when vt_rai_n / vt_rai_m
 * are non-NULL — CM2.rain_terminal_velocity(sb, vel, q_rai, ρ, ρ n_rai)
 * (src/Microphysics2M.jl:685-719) evaluated on the same clamped state.
 *
 * Inputs per point: ρ [kg/m3], T [K], q_tot, q_lcl, q_rai [kg/kg],
 * n_lcl, n_rai [1/kg] (per kg of air, as in BMT).  Outputs: dq_lcl_dt, dq_rai_dt
 * [kg/kg/s], dn_lcl_dt, dn_rai_dt [1/kg/s]; optional number- and mass-weighted
 * rain fall speeds [m/s].  The four identically-zero ice fields of the
 * reference's NamedTuple (BMT:840-842,852-853) are not materialised.
 * `vel` may be NULL iff both velocity columns are NULL.
 * ------------------------------------------------------------------------- */
int32_t cmx_sb2006_warm_rain_tendencies_f32(
    const cmx_warm_rain_2m_f32 *warm_rain, const cmx_thermo_f32 *tps,
    const cmx_rain_vel_f32 *vel, uint32_t flags, int64_t n,
    const float *rho, const float *T, const float *q_tot, const float *q_lcl,
    const float *n_lcl, const float *q_rai, const float *n_rai,
    float *dq_lcl_dt, float *dn_lcl_dt, float *dq_rai_dt, float *dn_rai_dt,
    float *vt_rai_n, float *vt_rai_m, void *stream);

int32_t cmx_sb2006_warm_rain_tendencies_f64(
    const cmx_warm_rain_2m_f64 *warm_rain, const cmx_thermo_f64 *tps,
    const cmx_rain_vel_f64 *vel, uint32_t flags, int64_t n,
    const double *rho, const double *T, const double *q_tot, const double *q_lcl,
    const double *n_lcl, const double *q_rai, const double *n_rai,
    double *dq_lcl_dt, double *dn_lcl_dt, double *dq_rai_dt, double *dn_rai_dt,
    double *vt_rai_n, double *vt_rai_m, void *stream);

/* (1b) The same tendencies behind the host model's own data layouts.
 * Input: 7 SEGMENTED columns in[k] (k = rho, T, q_tot, q_lcl, n_lcl, q_rai, n_rai): n_seg runs of seg_len contiguous
 * elements, run s of column k starting at in[k] + s·in_seg_stride[k] (elements).  That is a ClimaCore field in place: a
 * DataLayouts.VIJFH array (Nv, Ni, Nj, Nf, Nh) stores component f of element h as one run of Nv·Ni·Nj elements, i.e.
 * in[k] = pointer(parent(field)) + f·Nv·Ni·Nj, seg_len = Nv·Ni·Nj, stride = Nv·Ni·Nj·Nf, n_seg = Nh
 * (test/gpu_clima_core_test.jl:16-30,100-114); VF columns and VIJHF fields are the single-run case n_seg = 1 (strides may be NULL).
 * Output, exactly one of:
 *   out[4] + out_seg_stride[4]: dq_lcl_dt, dn_lcl_dt, dq_rai_dt, dn_rai_dt as segmented columns (each with its own stride);
 *   out_aos: n_seg·seg_len rows of 8 FT — the reference's result type, Vector{@NamedTuple{dq_lcl_dt, dn_lcl_dt, dq_rai_dt,
 *            dn_rai_dt, dq_ice_dt, dq_rim_dt, db_rim_dt, dn_lcl_activation_dt}} (BMT:852-853, test/gpu_performance.jl:212-216;
 *            the last four are 0), 16-byte aligned.
 * Values are bit-identical to cmx_sb2006_warm_rain_tendencies_* on the same points.  flags: CMX_SB2006_LIMITED. */
int32_t cmx_sb2006_warm_rain_tendencies_fields_f32(const cmx_warm_rain_2m_f32 *warm_rain, const cmx_thermo_f32 *tps, uint32_t flags, int64_t n_seg,
                                                   int64_t seg_len, const float *const *in, const int64_t *in_seg_stride, float *const *out,
                                                   const int64_t *out_seg_stride, float *out_aos, void *stream);
int32_t cmx_sb2006_warm_rain_tendencies_fields_f64(const cmx_warm_rain_2m_f64 *warm_rain, const cmx_thermo_f64 *tps, uint32_t flags, int64_t n_seg,
                                                   int64_t seg_len, const double *const *in, const int64_t *in_seg_stride, double *const *out,
                                                   const int64_t *out_seg_stride, double *out_aos, void *stream);

/* ---------------------------------------------------------------------------
 * (2) SB2006 per-process rates ("verbose" variant of the same kernel).
 *
 * Replaces the KA wrapper test_2_moment_SB2006_kernel! / SB2006_2M_kernel
 * (test/gpu_tests.jl:220-244): the individual CM2 process functions on
 * (q_tot, q_lcl, q_rai, N_lcl, N_rai, ρ, T) with N in [1/m3], plus the
 * cond/evap relaxation of src/MicrophysicsNonEq.jl:117-140.  `out` is a
 * host array of CMX_SB2006_NPROC device column pointers (any may be NULL to
 * skip that column), indexed by cmx_sb2006_process_column.
 * ------------------------------------------------------------------------- */
typedef enum cmx_sb2006_process_column {
    CMX_SB_ACNV_DQ_LCL = 0,  /* autoconversion(...).dq_lcl_dt    CM2:396-427 */
    CMX_SB_ACNV_DN_LCL,      /*                 .dN_lcl_dt [1/m3/s]          */
    CMX_SB_ACNV_DQ_RAI,      /*                 .dq_rai_dt                   */
    CMX_SB_ACNV_DN_RAI,      /*                 .dN_rai_dt [1/m3/s]          */
    CMX_SB_LCL_SELFCOL,      /* cloud_liquid_self_collection      CM2:488-501 */
    CMX_SB_ACCR_DQ_LCL,      /* accretion(...).dq_lcl_dt          CM2:445-470 */
    CMX_SB_ACCR_DN_LCL,      /*               .dN_lcl_dt [1/m3/s]            */
    CMX_SB_ACCR_DQ_RAI,      /*               .dq_rai_dt                     */
    CMX_SB_RAI_SELFCOL,      /* rain_self_collection [1/m3/s]     CM2:545-560 */
    CMX_SB_RAI_BREAKUP,      /* rain_breakup [1/m3/s]             CM2:579-601 */
    CMX_SB_RAI_VEL_N,        /* rain_terminal_velocity[1] (per `flags`) CM2:685-719 */
    CMX_SB_RAI_VEL_M,        /* rain_terminal_velocity[2]                    */
    CMX_SB_EVAP_DN_RAI,      /* rain_evaporation.∂ₜρn_rai [1/m3/s] CM2:780-828 */
    CMX_SB_EVAP_DQ_RAI,      /* rain_evaporation.∂ₜq_rai                     */
    CMX_SB_NUMADJ_RAI,       /* number_tendency_from_mass_limits(rain; q_rai, N_rai/ρ) CM2:882-891 */
    CMX_SB_NUMADJ_LCL,       /* same for cloud (xc_min, xc_max; q_lcl, N_lcl/ρ)     */
    CMX_SB_CONDEVAP,         /* _conv_q_vap_to_q_lcl_const        NonEq:117-140 */
    CMX_SB_DEVAP_DN_RAI,     /* ∂rain_evaporation_∂N_rai_∂q_rai(…).∂N_rai = ∂ₜρn_rai / N_rai  [1/s]  CM2:844-855 */
    CMX_SB_DEVAP_DQ_RAI,     /*                                   .∂q_rai = ∂ₜq_rai / q_rai   [1/s]              */
    CMX_SB2006_NPROC
} cmx_sb2006_process_column;

int32_t cmx_sb2006_process_rates_f32(
    const cmx_warm_rain_2m_f32 *warm_rain, const cmx_thermo_f32 *tps,
    const cmx_rain_vel_f32 *vel, uint32_t flags, int64_t n,
    const float *q_tot, const float *q_lcl, const float *q_rai, const float *N_lcl,
    const float *N_rai, const float *rho, const float *T,
    float *const out[CMX_SB2006_NPROC], void *stream);

int32_t cmx_sb2006_process_rates_f64(
    const cmx_warm_rain_2m_f64 *warm_rain, const cmx_thermo_f64 *tps,
    const cmx_rain_vel_f64 *vel, uint32_t flags, int64_t n,
    const double *q_tot, const double *q_lcl, const double *q_rai, const double *N_lcl,
    const double *N_rai, const double *rho, const double *T,
    double *const out[CMX_SB2006_NPROC], void *stream);

/* CM2.cloud_terminal_velocity(pdf_c, vel::StokesRegimeVelType, q_liq, ρₐ, N_liq) — src/Microphysics2M.jl:647-664:
 * number- and mass-weighted mean fall speeds of the cloud droplets (Stokes regime, generalized-gamma PSD moments
 * M^{2/3}, M^{5/3}; DistributionTools.jl:109-112).  N_liq is per m³ as in the reference.  Either output may be NULL. */
int32_t cmx_sb2006_cloud_terminal_velocity_f32(const cmx_cloud_pdf_sb2006_f32 *pdf_c, const cmx_stokes_vel_f32 *vel, int64_t n,
                                               const float *q_liq, const float *rho, const float *N_liq, float *vt_n, float *vt_m,
                                               void *stream);
int32_t cmx_sb2006_cloud_terminal_velocity_f64(const cmx_cloud_pdf_sb2006_f64 *pdf_c, const cmx_stokes_vel_f64 *vel, int64_t n,
                                               const double *q_liq, const double *rho, const double *N_liq, double *vt_n, double *vt_m,
                                               void *stream);

/* ---------------------------------------------------------------------------
 * (2b) Fused COLUMN kernel: the north-star tendencies + the sedimentation step a host model applies
 * right after them, in one pass: 7 columns in, 4 out (44 B/point f32) instead of 88 B/point for the unfused sequence.
 *
 * Replaces, per column of n_lev contiguous levels (level 0 = lowest; flat index i = col·n_lev + k — a ClimaCore VF /
 * VIJFH(Ni = Nj = 1) column field; test/gpu_clima_core_test.jl:16-30 builds such spaces):
 *   tend = BMT.bulk_microphysics_tendencies(Microphysics2Moment(), mp, tps, ρ, T, q_tot, q_lcl, n_lcl, q_rai, n_rai)   BMT:820-854
 *   (w_n, w_m) = CM2.rain_terminal_velocity(sb, vel, q_rai, ρ, ρ n_rai)                                                CM2:685-719
 *   (c_n, c_m) = CM2.cloud_terminal_velocity(sb.pdf_c, cloud_vel, q_lcl, ρ, ρ n_lcl)    [iff cloud_vel != NULL]        CM2:647-664
 * followed by the host model's first-order upwind ("right-biased": the value of the cell above) flux divergence
 *   F_k = ρ_k χ_k w_k,   ∂χ_k/∂t += (F_{k+1} − F_k) · inv_dz[k] / ρ_k,   F_{n_lev} = 0,
 * for χ = q_rai (w_m), n_rai (w_n) and — with cloud_vel — q_lcl (c_m), n_lcl (c_n).  The flux scheme is the host model's
 * (ClimaAtmos precipitation advection), NOT part of the reference package: its oracle is a restatement of the formula
 * above ("parity unpinned" for the flux step; the tendencies and fall speeds it consumes are pinned as in (1)).
 * flags: CMX_SB2006_LIMITED and exactly one of CMX_VEL_SB2006 / CMX_VEL_CHEN2022.  inv_dz: n_lev values 1/Δz_k [1/m].
 * precip_flux (optional, n_col values): F_0 of q_rai, the surface precipitation mass flux [kg m⁻² s⁻¹].
 * Results do not depend on tile boundaries or alignment (bit-identical to a one-point-per-lane evaluation).
 * ------------------------------------------------------------------------- */
int32_t cmx_sb2006_column_tendencies_sedimentation_f32(
    const cmx_warm_rain_2m_f32 *warm_rain, const cmx_thermo_f32 *tps, const cmx_rain_vel_f32 *vel,
    const cmx_stokes_vel_f32 *cloud_vel, uint32_t flags, int64_t n_col, int32_t n_lev, const float *inv_dz,
    const float *rho, const float *T, const float *q_tot, const float *q_lcl, const float *n_lcl, const float *q_rai,
    const float *n_rai, float *dq_lcl_dt, float *dn_lcl_dt, float *dq_rai_dt, float *dn_rai_dt, float *precip_flux,
    void *stream);
int32_t cmx_sb2006_column_tendencies_sedimentation_f64(
    const cmx_warm_rain_2m_f64 *warm_rain, const cmx_thermo_f64 *tps, const cmx_rain_vel_f64 *vel,
    const cmx_stokes_vel_f64 *cloud_vel, uint32_t flags, int64_t n_col, int32_t n_lev, const double *inv_dz,
    const double *rho, const double *T, const double *q_tot, const double *q_lcl, const double *n_lcl, const double *q_rai,
    const double *n_rai, double *dq_lcl_dt, double *dn_lcl_dt, double *dq_rai_dt, double *dn_rai_dt, double *precip_flux,
    void *stream);

/* Bulk cloud → rain conversion of the other two-moment schemes — src/Microphysics2M.jl:920-1003:
 *   acnv = CM2.conv_q_lcl_to_q_rai(scheme, q_lcl, ρ, N_d[, smooth_transition])     KK2000 | B1994 | TC1980 | LD2004
 *   accr = CM2.accretion(scheme, q_lcl, q_rai, ρ)                                   KK2000 | B1994 | TC1980
 * (KA kernels test_2_moment_acnv_kernel! / test_2_moment_accr_kernel!, test/gpu_tests.jl:782-818).  `scheme` is one of
 * CMX_2M_KK2000 … CMX_2M_LD2004; CMX_2M_SMOOTH_TRANSITION selects the logistic threshold (src/Common.jl:125-139) instead
 * of the step.  N_d per m³.  accr (and q_rai) may be NULL; LD2004 has no accretion (accr must be NULL). */
#define CMX_2M_KK2000 0u
#define CMX_2M_B1994 1u
#define CMX_2M_TC1980 2u
#define CMX_2M_LD2004 3u
#define CMX_2M_SMOOTH_TRANSITION (1u << 8)
int32_t cmx_bulk_2m_cloud_to_rain_f32(const cmx_bulk_2m_schemes_f32 *schemes, uint32_t scheme, int64_t n, const float *q_lcl,
                                      const float *q_rai, const float *rho, const float *N_d, float *acnv, float *accr, void *stream);
int32_t cmx_bulk_2m_cloud_to_rain_f64(const cmx_bulk_2m_schemes_f64 *schemes, uint32_t scheme, int64_t n, const double *q_lcl,
                                      const double *q_rai, const double *rho, const double *N_d, double *acnv, double *accr,
                                      void *stream);

/* ---------------------------------------------------------------------------
 * (4) Ice nucleation rates — ABIFM immersion freezing + Koop-2000 homogeneous freezing.
 *
 * Replaces the KA wrappers IceNucleation_ABIFM_J_kernel!, IceNucleation_homogeneous_J_kernel!,
 * Common_a_w_ice_kernel! (test/gpu_tests.jl:294-362) and the per-droplet products the parcel
 * model forms from them (parcel/ParcelTendencies.jl:120-133,194-205):
 *   Δa_w    = a_w − CO.a_w_ice(tps, T)                         src/Common.jl:267-271
 *   J_het   = CMI_het.ABIFM_J(dust, Δa_w)          [m⁻² s⁻¹]  src/IceNucleation.jl:124-134
 *   J_hom   = CMI_hom.homogeneous_J_cubic(ip, Δa_w) [m⁻³ s⁻¹]  src/IceNucleation.jl:557-565
 *             (or homogeneous_J_linear, :581-584, with CMX_ICENUC_HOM_LINEAR)
 *   rate_het = J_het · 4π r²    rate_hom = J_hom · 4/3 π r³    [s⁻¹ per droplet]
 * Inputs per point: T [K], a_w [-], r [m].  Any output column may be NULL.
 * homogeneous_J_cubic THROWS DomainError outside [Δa_w_min, Δa_w_max]; a device kernel cannot:
 * such points get J_hom = rate_hom = NaN and, if `n_domain_errors` is non-NULL, are counted.
 * `n_domain_errors` is a device array of CMX_ICENUC_ERR_WORDS int64, zeroed by the caller; the number of
 * domain-error points is the SUM of all its words (counts accumulate across calls until re-zeroed).
 * Only every (WORDS/SLOTS)-th word is written: one 128-byte line per slot, so that the ≈1e5 concurrent
 * workgroups of a 1e8-point launch do not serialise on a single L2 atomic unit.
 * ------------------------------------------------------------------------- */
#define CMX_ICENUC_HOM_LINEAR   (1u << 0)   /* homogeneous_J_linear instead of homogeneous_J_cubic */
#define CMX_ICENUC_ERR_SLOTS    64
#define CMX_ICENUC_ERR_WORDS    1024        /* int64 words in the n_domain_errors buffer (8 KiB) */

int32_t cmx_ice_nucleation_rates_f32(
    const cmx_thermo_f32 *tps, const cmx_abifm_dust_f32 *dust, const cmx_koop2000_f32 *koop,
    uint32_t flags, int64_t n, const float *T, const float *a_w, const float *r,
    float *delta_a_w, float *J_het, float *J_hom, float *rate_het, float *rate_hom,
    int64_t *n_domain_errors, void *stream);

int32_t cmx_ice_nucleation_rates_f64(
    const cmx_thermo_f64 *tps, const cmx_abifm_dust_f64 *dust, const cmx_koop2000_f64 *koop,
    uint32_t flags, int64_t n, const double *T, const double *a_w, const double *r,
    double *delta_a_w, double *J_het, double *J_hom, double *rate_het, double *rate_hom,
    int64_t *n_domain_errors, void *stream);

/* The same with the water activity of a sulphuric-acid solution droplet computed in the kernel, as the parcel model drives it
 * (parcel/ParcelTendencies.jl:120-133: a_w = CO.a_w_xT(H2SO4_prs, tps, x_sulph, T) when T < T_max of the solution fit): the second
 * input column is x, the weight fraction of H2SO4 [-], instead of a_w;  a_w = p_sol(x, T) / p_sat,liq(T), src/Common.jl:188-246. */
int32_t cmx_ice_nucleation_rates_xT_f32(const cmx_thermo_f32 *tps, const cmx_abifm_dust_f32 *dust, const cmx_koop2000_f32 *koop,
                                        const cmx_h2so4_solution_params_f32 *h2so4, uint32_t flags, int64_t n, const float *T, const float *x_sulph,
                                        const float *r, float *delta_a_w, float *J_het, float *J_hom, float *rate_het, float *rate_hom,
                                        int64_t *n_domain_errors, void *stream);
int32_t cmx_ice_nucleation_rates_xT_f64(const cmx_thermo_f64 *tps, const cmx_abifm_dust_f64 *dust, const cmx_koop2000_f64 *koop,
                                        const cmx_h2so4_solution_params_f64 *h2so4, uint32_t flags, int64_t n, const double *T, const double *x_sulph,
                                        const double *r, double *delta_a_w, double *J_het, double *J_hom, double *rate_het, double *rate_hom,
                                        int64_t *n_domain_errors, void *stream);

/* CO.H2SO4_soln_saturation_vapor_pressure(prs, x, T) [Pa] and CO.a_w_xT(prs, tps, x, T) over columns — src/Common.jl:188-246
 * (KA wrapper Common_H2SO4_kernel!, test/gpu_tests.jl:876-893).  Either output may be NULL. */
int32_t cmx_h2so4_solution_f32(const cmx_h2so4_solution_params_f32 *prs, const cmx_thermo_f32 *tps, int64_t n, const float *x_sulph, const float *T,
                               float *p_sol, float *a_w, void *stream);
int32_t cmx_h2so4_solution_f64(const cmx_h2so4_solution_params_f64 *prs, const cmx_thermo_f64 *tps, int64_t n, const double *x_sulph, const double *T,
                               double *p_sol, double *a_w, void *stream);

/* Mohler et al. (2006) deposition nucleation on dust — src/IceNucleation.jl:44-79 (KA wrappers
 * IceNucleation_dust_activated_number_fraction_kernel!, IceNucleation_MohlerDepositionRate_kernel!, test/gpu_tests.jl:930-966):
 *   act_frac = CMI_het.dust_activated_number_fraction(dust, ip, S_i, T) = max(0, exp(a (S_i − S₀)) − 1)
 *   dep_rate = CMI_het.MohlerDepositionRate(dust, ip, S_i, T, dSi_dt, N_aer) = max(0, N_aer a dSi_dt)        (a, S₀ by T ≷ T_thr)
 * Columns S_i, T; dSi_dt and N_aer (both needed for dep_rate, NULL otherwise); either output may be NULL.  The reference asserts
 * S_i < Sᵢ_max: such points get NaN and are counted in `n_domain_errors` (optional: ONE device int64, accumulating, zeroed by the caller). */
int32_t cmx_mohler2006_deposition_f32(const cmx_mohler_dust_f32 *dust, const cmx_mohler2006_f32 *ip, int64_t n, const float *S_i, const float *T,
                                      const float *dSi_dt, const float *N_aer, float *act_frac, float *dep_rate, int64_t *n_domain_errors,
                                      void *stream);
int32_t cmx_mohler2006_deposition_f64(const cmx_mohler_dust_f64 *dust, const cmx_mohler2006_f64 *ip, int64_t n, const double *S_i, const double *T,
                                      const double *dSi_dt, const double *N_aer, double *act_frac, double *dep_rate, int64_t *n_domain_errors,
                                      void *stream);

/* CMI_het.deposition_J(dust, Δa_w) = 10^(m Δa_w + c + 4) [m⁻² s⁻¹] — src/IceNucleation.jl:81-102 (water-activity based deposition
 * nucleation, China et al. 2017 / Alpert et al. 2022; KA wrapper IceNucleation_deposition_J_kernel!, test/gpu_tests.jl:968-984). */
int32_t cmx_deposition_J_f32(const cmx_deposition_dust_f32 *dust, int64_t n, const float *delta_a_w, float *J, void *stream);
int32_t cmx_deposition_J_f64(const cmx_deposition_dust_f64 *dust, int64_t n, const double *delta_a_w, double *J, void *stream);

/* CMI_het.INP_concentration_frequency(params, INPC, T) — src/IceNucleation.jl:219-226 (Frostenberg et al. 2023: log-normal relative
 * frequency of an INP concentration at temperature T; 0 at and above T_freeze; KA wrapper IceNucleation_INPC_frequency_kernel!,
 * test/gpu_tests.jl:1041-1055). */
int32_t cmx_inp_concentration_frequency_f32(const cmx_frostenberg2023_f32 *ip, int64_t n, const float *INPC, const float *T, float *freq, void *stream);
int32_t cmx_inp_concentration_frequency_f64(const cmx_frostenberg2023_f64 *ip, int64_t n, const double *INPC, const double *T, double *freq, void *stream);

/* CO.a_w_ice(tps, T) and CO.a_w_eT(tps, e, T) over columns — src/Common.jl:250-253,267-271
 * (KA wrappers Common_a_w_ice_kernel!, Common_a_w_eT_kernel!, test/gpu_tests.jl:340-362).
 * `e` may be NULL iff `a_w_eT` is NULL. */
int32_t cmx_water_activity_f32(const cmx_thermo_f32 *tps, int64_t n, const float *T, const float *e,
                               float *a_w_ice, float *a_w_eT, void *stream);
int32_t cmx_water_activity_f64(const cmx_thermo_f64 *tps, int64_t n, const double *T, const double *e,
                               double *a_w_ice, double *a_w_eT, void *stream);

/* ---------------------------------------------------------------------------
 * (5) One-moment (Marshall–Palmer) bulk scheme.
 *
 * `flags` = the reference's Microphysics1MOptions (src/parameters/Microphysics1MOptions.jl:257-286):
 * one bit per process / variant; a cleared bit is the reference's `nothing` (process disabled).
 * CMX_1M_DEFAULT_OPTIONS reproduces `Microphysics1MOptions()`.  Exactly one variant bit of a process may be set
 * (CMX_ERR_BAD_ARG otherwise).
 * ------------------------------------------------------------------------- */
#define CMX_1M_CLOUD_LIQUID_FORMATION     (1u << 0)   /* CloudLiquidFormation        NonEq:104-140 */
#define CMX_1M_CLOUD_ICE_FORMATION_CONST  (1u << 1)   /* ConstantTimescale           NonEq:163-193 */
#define CMX_1M_CLOUD_ICE_FORMATION_TDEP   (1u << 2)   /* TemperatureDependent          NonEq:32-50,194-224 */
#define CMX_1M_CLOUD_ICE_MELT             (1u << 3)   /* CloudIceMelt                CM1:1055-1077 */
#define CMX_1M_RAIN_ACNV_KESSLER          (1u << 4)   /* Kessler1M                   CM1:354-358   */
#define CMX_1M_RAIN_ACNV_PRESCRIBED_ND    (1u << 5)   /* PrescribedNd                CM1:359-364   */
#define CMX_1M_SNOW_ACNV_NO_SUPERSAT      (1u << 6)   /* NoSupersaturation           CM1:414-418   */
#define CMX_1M_SNOW_ACNV_WITH_SUPERSAT    (1u << 7)   /* WithSupersaturation         CM1:420-446   */
#define CMX_1M_RAIN_EVAPORATION           (1u << 8)   /* RainEvaporation             CM1:917-960   */
#define CMX_1M_SNOW_SUBLIMATION_ONLY      (1u << 9)   /* SublimationOnly             CM1:979-988   */
#define CMX_1M_SNOW_DEP_AND_SUBL          (1u << 10)  /* DepositionAndSublimation    CM1:990-999   */
#define CMX_1M_SNOW_MELT                  (1u << 11)  /* SnowMelt                    CM1:1094-1139 */
#define CMX_1M_ACCR_LCL_RAI               (1u << 12)  /* CloudLiquidRainAccretion    CM1:709-732   */
#define CMX_1M_ACCR_LCL_SNO               (1u << 13)  /* CloudLiquidSnowAccretion    CM1:734-760   */
#define CMX_1M_ACCR_ICL_RAI               (1u << 14)  /* CloudIceRainAccretion (+ rain sink) CM1:762-785,872-897 */
#define CMX_1M_ACCR_ICL_SNO               (1u << 15)  /* CloudIceSnowAccretion       CM1:787-810   */
#define CMX_1M_ACCR_RAI_SNO               (1u << 16)  /* RainSnowAccretion           CM1:815-867   */
#define CMX_1M_DEFAULT_OPTIONS                                                                            \
    (CMX_1M_CLOUD_LIQUID_FORMATION | CMX_1M_CLOUD_ICE_FORMATION_CONST | CMX_1M_CLOUD_ICE_MELT |           \
     CMX_1M_RAIN_ACNV_KESSLER | CMX_1M_SNOW_ACNV_NO_SUPERSAT | CMX_1M_RAIN_EVAPORATION |                  \
     CMX_1M_SNOW_DEP_AND_SUBL | CMX_1M_SNOW_MELT | CMX_1M_ACCR_LCL_RAI | CMX_1M_ACCR_LCL_SNO |            \
     CMX_1M_ACCR_ICL_RAI | CMX_1M_ACCR_ICL_SNO | CMX_1M_ACCR_RAI_SNO)

/* Fused 1M tendencies.  Replaces the broadcast
 *   BMT.bulk_microphysics_tendencies.(Ref(BMT.Instantaneous()), Ref(BMT.Microphysics1Moment()), Ref(mp), Ref(tps),
 *                                     ρ, T, q_tot, q_lcl, q_icl, q_rai, q_sno)
 * src/BulkMicrophysicsTendencies.jl:505-514 → _microphysics_source_terms :141-217 → _aggregate_tendencies
 * :227-252 (KA wrapper benchmark_1m_bulk_tendencies_kernel!, test/gpu_performance.jl:39-47).
 * 7 columns in, 4 out (dq_lcl_dt, dq_icl_dt, dq_rai_dt, dq_sno_dt [kg/kg/s]). */
int32_t cmx_mp1m_tendencies_f32(const cmx_microphysics_1m_f32 *mp, const cmx_thermo_f32 *tps, uint32_t flags,
                                int64_t n, const float *rho, const float *T, const float *q_tot, const float *q_lcl,
                                const float *q_icl, const float *q_rai, const float *q_sno, float *dq_lcl_dt,
                                float *dq_icl_dt, float *dq_rai_dt, float *dq_sno_dt, void *stream);
int32_t cmx_mp1m_tendencies_f64(const cmx_microphysics_1m_f64 *mp, const cmx_thermo_f64 *tps, uint32_t flags,
                                int64_t n, const double *rho, const double *T, const double *q_tot,
                                const double *q_lcl, const double *q_icl, const double *q_rai, const double *q_sno,
                                double *dq_lcl_dt, double *dq_icl_dt, double *dq_rai_dt, double *dq_sno_dt,
                                void *stream);

/* The 1-moment Instantaneous tendencies behind the host model's layouts — same conventions as
 * cmx_sb2006_warm_rain_tendencies_fields_*: in[7] = (rho, T, q_tot, q_lcl, q_icl, q_rai, q_sno) as segmented columns (n_seg runs of
 * seg_len elements, per-column run strides; a ClimaCore VIJFH field component in place), output either out[4] segmented columns or
 * out_aos = n rows of the reference's NamedTuple (dq_lcl_dt, dq_icl_dt, dq_rai_dt, dq_sno_dt) (BMT:246-251; the output type of
 * benchmark_1m_bulk_tendencies_kernel!, test/gpu_performance.jl:138-182).  Bit-identical to cmx_mp1m_tendencies_*. */
int32_t cmx_mp1m_tendencies_fields_f32(const cmx_microphysics_1m_f32 *mp, const cmx_thermo_f32 *tps, uint32_t flags, int64_t n_seg, int64_t seg_len,
                                       const float *const *in, const int64_t *in_seg_stride, float *const *out, const int64_t *out_seg_stride,
                                       float *out_aos, void *stream);
int32_t cmx_mp1m_tendencies_fields_f64(const cmx_microphysics_1m_f64 *mp, const cmx_thermo_f64 *tps, uint32_t flags, int64_t n_seg, int64_t seg_len,
                                       const double *const *in, const int64_t *in_seg_stride, double *const *out, const int64_t *out_seg_stride,
                                       double *out_aos, void *stream);

/* bulk_microphysics_tendencies(LinearizedAverage(), Microphysics1Moment(), mp, tps, ρ, T, q_tot, q_lcl, q_icl, q_rai, q_sno,
 * Δt, nsub) — src/BulkMicrophysicsTendencies.jl:572-632 (the mode ClimaAtmos runs operationally, :112-115): the average
 * tendencies over Δt from `nsub` linearized implicit substeps (:381-465) of the donor-based linearization dq/dt ≈ M q + e
 * (:269-379), with T updated from the latent heating of each substep.  q_min = TD.Parameters.q_min(tps) (the donor floor
 * of the linearization).  Same flags (Microphysics1MOptions) as cmx_mp1m_tendencies_*. */
int32_t cmx_mp1m_linearized_average_f32(const cmx_microphysics_1m_f32 *mp, const cmx_thermo_f32 *tps, uint32_t flags, float q_min,
                                        float dt, int32_t nsub, int64_t n, const float *rho, const float *T, const float *q_tot,
                                        const float *q_lcl, const float *q_icl, const float *q_rai, const float *q_sno,
                                        float *dq_lcl_dt, float *dq_icl_dt, float *dq_rai_dt, float *dq_sno_dt, void *stream);
int32_t cmx_mp1m_linearized_average_f64(const cmx_microphysics_1m_f64 *mp, const cmx_thermo_f64 *tps, uint32_t flags, double q_min,
                                        double dt, int32_t nsub, int64_t n, const double *rho, const double *T, const double *q_tot,
                                        const double *q_lcl, const double *q_icl, const double *q_rai, const double *q_sno,
                                        double *dq_lcl_dt, double *dq_icl_dt, double *dq_rai_dt, double *dq_sno_dt, void *stream);

/* … and behind the host model's layouts, exactly as cmx_mp1m_tendencies_fields_* (segmented columns in place, SoA or the reference's
 * array-of-NamedTuples output).  Bit-identical to cmx_mp1m_linearized_average_*. */
int32_t cmx_mp1m_linearized_average_fields_f32(const cmx_microphysics_1m_f32 *mp, const cmx_thermo_f32 *tps, uint32_t flags, float q_min, float dt,
                                               int32_t nsub, int64_t n_seg, int64_t seg_len, const float *const *in, const int64_t *in_seg_stride,
                                               float *const *out, const int64_t *out_seg_stride, float *out_aos, void *stream);
int32_t cmx_mp1m_linearized_average_fields_f64(const cmx_microphysics_1m_f64 *mp, const cmx_thermo_f64 *tps, uint32_t flags, double q_min, double dt,
                                               int32_t nsub, int64_t n_seg, int64_t seg_len, const double *const *in, const int64_t *in_seg_stride,
                                               double *const *out, const int64_t *out_seg_stride, double *out_aos, void *stream);

/* The OPERATIONAL 1-moment column step in one pass (SURVEY §8f-1 + §8f-4): per grid point of n_col columns × n_lev contiguous levels
 * (flat index col·n_lev + k, level 0 = lowest — `parent(field)` of a ClimaCore column field)
 *   tend  = BMT.bulk_microphysics_tendencies(mode, Microphysics1Moment(), mp, tps, ρ, T, q_tot, q_lcl, q_icl, q_rai, q_sno[, Δt, nsub])
 *           mode = Instantaneous() when nsub = 0 (BMT:505-514), LinearizedAverage() when nsub ≥ 1 (BMT:572-632; q_min, dt as there)
 *   w     = the four bulk fall speeds a host model precomputes — exactly cmx_sedimentation_velocities_* on the clamped state
 *           (CMNonEq.terminal_velocity cloud liquid / cloud ice, CM1.terminal_velocity Chen-2022 rain / snow; NonEq:250-281, CM1:251-297)
 * followed by the host model's first-order upwind ("right-biased") flux divergence of the four species
 *   F_k = ρ_k χ_k w_k,   ∂χ_k/∂t += (F_{k+1} − F_k) · inv_dz[k] / ρ_k,   F_{n_lev} = 0,    χ = q_lcl, q_icl, q_rai, q_sno.
 * The flux scheme is the host model's (ClimaAtmos precipitation advection; column spaces test/gpu_clima_core_test.jl:16-45), NOT part
 * of the reference package: "parity unpinned" for the flux step, as for cmx_sb2006_column_tendencies_sedimentation_*.
 * in = HOST array of the 7 device columns (rho, T, q_tot, q_lcl, q_icl, q_rai, q_sno), out = HOST array of the 4 device tendency
 * columns; precip_rai / precip_sno (optional, n_col values): the surface fluxes F_0 of rain and snow [kg m⁻² s⁻¹].  44 B/point (f32)
 * instead of the 148 B/point of the three unfused steps.  Results do not depend on tile boundaries or alignment; a NaN in ρ or in a
 * species' q gives a NaN flux of that species (and a NaN tendency of that species in the cell below). */
int32_t cmx_mp1m_column_tendencies_sedimentation_f32(const cmx_microphysics_1m_f32 *mp, const cmx_thermo_f32 *tps, const cmx_stokes_vel_f32 *stokes,
                                                     const cmx_chen2022_rain_vel_f32 *chen_rain, const cmx_chen2022_ice_vel_f32 *chen_ice, uint32_t flags,
                                                     float q_min, float dt, int32_t nsub, int64_t n_col, int32_t n_lev, const float *inv_dz,
                                                     const float *const *in, float *const *out, float *precip_rai, float *precip_sno, void *stream);
int32_t cmx_mp1m_column_tendencies_sedimentation_f64(const cmx_microphysics_1m_f64 *mp, const cmx_thermo_f64 *tps, const cmx_stokes_vel_f64 *stokes,
                                                     const cmx_chen2022_rain_vel_f64 *chen_rain, const cmx_chen2022_ice_vel_f64 *chen_ice, uint32_t flags,
                                                     double q_min, double dt, int32_t nsub, int64_t n_col, int32_t n_lev, const double *inv_dz,
                                                     const double *const *in, double *const *out, double *precip_rai, double *precip_sno, void *stream);

/* The individual 1M source terms — `_microphysics_source_terms` (BMT:141-217), same inputs (clamped the
 * same way), `out` = host array of CMX_MP1M_NSRC device column pointers (NULL = skip). */
typedef enum cmx_mp1m_source_column {
    CMX_1M_S_PHASE_CHANGE_VAP_LCL = 0, CMX_1M_S_PHASE_CHANGE_VAP_ICL,
    CMX_1M_S_ACNV_LCL_RAI, CMX_1M_S_ACNV_ICL_SNO,
    CMX_1M_S_ACCR_LCL_RAI, CMX_1M_S_ACCR_LCL_SNO_COLD, CMX_1M_S_ACCR_LCL_SNO_WARM, CMX_1M_S_ACCR_MELT_LCL_SNO,
    CMX_1M_S_ACCR_ICL_RAI, CMX_1M_S_ACCR_FREEZE_ICL_RAI, CMX_1M_S_ACCR_ICL_SNO,
    CMX_1M_S_ACCR_RAI_SNO_COLD, CMX_1M_S_ACCR_RAI_SNO_WARM, CMX_1M_S_ACCR_MELT_RAI_SNO,
    CMX_1M_S_PHASE_CHANGE_VAP_RAI, CMX_1M_S_PHASE_CHANGE_VAP_SNO,
    CMX_1M_S_MELT_ICL_LCL, CMX_1M_S_MELT_SNO_RAI,
    CMX_MP1M_NSRC
} cmx_mp1m_source_column;
int32_t cmx_mp1m_source_terms_f32(const cmx_microphysics_1m_f32 *mp, const cmx_thermo_f32 *tps, uint32_t flags,
                                  int64_t n, const float *rho, const float *T, const float *q_tot, const float *q_lcl,
                                  const float *q_icl, const float *q_rai, const float *q_sno,
                                  float *const out[CMX_MP1M_NSRC], void *stream);
int32_t cmx_mp1m_source_terms_f64(const cmx_microphysics_1m_f64 *mp, const cmx_thermo_f64 *tps, uint32_t flags,
                                  int64_t n, const double *rho, const double *T, const double *q_tot,
                                  const double *q_lcl, const double *q_icl, const double *q_rai, const double *q_sno,
                                  double *const out[CMX_MP1M_NSRC], void *stream);

/* Mass-weighted 1M fall speeds over (ρ, q) columns — BASELINE config 1 together with the autoconversion
 * source term above.  Replaces `@. w = CM1.terminal_velocity(rain, vel, ρ, q)` (test/gpu_clima_core_test.jl:39-42):
 *   vt_rai_blk1m = CM1.terminal_velocity(rain, Blk1MVelTypeRain, ρ, q_rai)       CM1:223-249
 *   vt_sno_blk1m = CM1.terminal_velocity(snow, Blk1MVelTypeSnow, ρ, q_sno)       CM1:223-249
 *   vt_rai_chen  = CM1.terminal_velocity(rain, Chen2022VelTypeRain, ρ, q_rai)    CM1:251-270
 * Output columns may be NULL; q_sno may be NULL iff vt_sno_blk1m is; chen may be NULL iff vt_rai_chen is. */
int32_t cmx_mp1m_terminal_velocity_f32(const cmx_microphysics_1m_f32 *mp, const cmx_chen2022_rain_vel_f32 *chen,
                                       int64_t n, const float *rho, const float *q_rai, const float *q_sno,
                                       float *vt_rai_blk1m, float *vt_sno_blk1m, float *vt_rai_chen, void *stream);
int32_t cmx_mp1m_terminal_velocity_f64(const cmx_microphysics_1m_f64 *mp, const cmx_chen2022_rain_vel_f64 *chen,
                                       int64_t n, const double *rho, const double *q_rai, const double *q_sno,
                                       double *vt_rai_blk1m, double *vt_sno_blk1m, double *vt_rai_chen, void *stream);

/* The bulk sedimentation velocities a host model precomputes per cell (ClimaAtmos set_sedimentation_precomputed_quantities;
 * test/gpu_clima_core_test.jl:36-45; KA kernel test_chen2022_terminal_velocity_kernel!, test/gpu_tests.jl:608-630):
 *   w_lcl = CMNonEq.terminal_velocity(liquid, ::StokesRegimeVelType, ρ, q_lcl)          src/MicrophysicsNonEq.jl:250-265
 *   w_icl = CMNonEq.terminal_velocity(ice, ::Chen2022VelTypeSmallIce, ρ, q_icl)         src/MicrophysicsNonEq.jl:267-281
 *   w_rai = CM1.terminal_velocity(rain, ::Chen2022VelTypeRain, ρ, q_rai)                src/Microphysics1M.jl:251-270
 *   w_sno = CM1.terminal_velocity(snow, ::Chen2022VelTypeLargeIce, ρ, q_sno)            src/Microphysics1M.jl:272-297
 * Any (q, w) pair may be NULL together with the parameter struct only it needs. */
int32_t cmx_sedimentation_velocities_f32(const cmx_microphysics_1m_f32 *mp, const cmx_stokes_vel_f32 *stokes,
                                         const cmx_chen2022_rain_vel_f32 *chen_rain, const cmx_chen2022_ice_vel_f32 *chen_ice, int64_t n,
                                         const float *rho, const float *q_lcl, const float *q_icl, const float *q_rai, const float *q_sno,
                                         float *w_lcl, float *w_icl, float *w_rai, float *w_sno, void *stream);
int32_t cmx_sedimentation_velocities_f64(const cmx_microphysics_1m_f64 *mp, const cmx_stokes_vel_f64 *stokes,
                                         const cmx_chen2022_rain_vel_f64 *chen_rain, const cmx_chen2022_ice_vel_f64 *chen_ice, int64_t n,
                                         const double *rho, const double *q_lcl, const double *q_icl, const double *q_rai,
                                         const double *q_sno, double *w_lcl, double *w_icl, double *w_rai, double *w_sno, void *stream);

/* ---------------------------------------------------------------------------
 * (6) Abdul-Razzak & Ghan (2000) aerosol activation.
 *
 * Replaces the broadcasts
 *   AA.N_activated_per_mode.(Ref(ap), Ref(ad), Ref(aip), Ref(tps), T, p, w, q_tot, q_liq, q_ice[, N_liq, N_ice])
 *   AA.M_activated_per_mode.(…)      AA.max_supersaturation.(…)
 * src/AerosolActivation.jl:138-200, 235-259, 294-321 (KA wrapper aerosol_activation_kernel!,
 * test/gpu_tests.jl:45-79) for an aerosol distribution shared by all states (BASELINE config 3).
 * Inputs per state: T [K], p [Pa], w [m/s], q_tot [kg/kg]; q_liq, q_ice [kg/kg] and N_liq, N_ice [1/m3] are
 * optional columns (NULL = 0, the reference's 10-argument methods).  Outputs (all optional): `N_act` / `M_act`
 * = host arrays of ad->n_modes device column pointers (activated number [1/m3] / mass per mode), `S_max`.
 * ------------------------------------------------------------------------- */
int32_t cmx_arg2000_activation_f32(
    const cmx_aerosol_activation_params_f32 *ap, const cmx_aerosol_distribution_f32 *ad,
    const cmx_air_properties_f32 *aip, const cmx_thermo_f32 *tps, int64_t n,
    const float *T, const float *p, const float *w, const float *q_tot, const float *q_liq, const float *q_ice,
    const float *N_liq, const float *N_ice, float *const *N_act, float *const *M_act, float *S_max, void *stream);
int32_t cmx_arg2000_activation_f64(
    const cmx_aerosol_activation_params_f64 *ap, const cmx_aerosol_distribution_f64 *ad,
    const cmx_air_properties_f64 *aip, const cmx_thermo_f64 *tps, int64_t n,
    const double *T, const double *p, const double *w, const double *q_tot, const double *q_liq, const double *q_ice,
    const double *N_liq, const double *N_ice, double *const *N_act, double *const *M_act, double *S_max, void *stream);

/* AA.total_N_activated(…) / AA.total_M_activated(…) — src/AerosolActivation.jl:355-433: the sums over the modes of N_activated_per_mode /
 * M_activated_per_mode (left to right, as Julia sums the tuple), formed in the activation kernel.  Either output may be NULL, not both. */
int32_t cmx_arg2000_total_activated_f32(const cmx_aerosol_activation_params_f32 *ap, const cmx_aerosol_distribution_f32 *ad,
                                        const cmx_air_properties_f32 *aip, const cmx_thermo_f32 *tps, int64_t n, const float *T, const float *p,
                                        const float *w, const float *q_tot, const float *q_liq, const float *q_ice, const float *N_liq,
                                        const float *N_ice, float *N_total, float *M_total, void *stream);
int32_t cmx_arg2000_total_activated_f64(const cmx_aerosol_activation_params_f64 *ap, const cmx_aerosol_distribution_f64 *ad,
                                        const cmx_air_properties_f64 *aip, const cmx_thermo_f64 *tps, int64_t n, const double *T, const double *p,
                                        const double *w, const double *q_tot, const double *q_liq, const double *q_ice, const double *N_liq,
                                        const double *N_ice, double *N_total, double *M_total, void *stream);

/* Same activation for aerosol that varies in space: every mode's (r_dry, stdev, N, hygroscopicity, molar_mass_mix) is a
 * device column (arrays of n_modes column pointers), as the reference's own KA kernel passes them per element
 * (aerosol_activation_kernel!, test/gpu_tests.jl:45-79).  hygroscopicity = the mode's mean B̄ or κ̄
 * (AA.mean_hygroscopicity_parameter, src/AerosolActivation.jl:61-97); molar_mass (nullable) = Σ w_j M_j, only needed
 * for M_act.  The mode-only factors the shared-distribution entry folds on the host are formed per state here. */
int32_t cmx_arg2000_activation_columns_f32(
    const cmx_aerosol_activation_params_f32 *ap, const cmx_air_properties_f32 *aip, const cmx_thermo_f32 *tps, int32_t n_modes,
    int64_t n, const float *T, const float *p, const float *w, const float *q_tot, const float *q_liq, const float *q_ice,
    const float *N_liq, const float *N_ice, const float *const *r_dry, const float *const *stdev, const float *const *N_mode,
    const float *const *hygroscopicity, const float *const *molar_mass, float *const *N_act, float *const *M_act, float *S_max,
    void *stream);
int32_t cmx_arg2000_activation_columns_f64(
    const cmx_aerosol_activation_params_f64 *ap, const cmx_air_properties_f64 *aip, const cmx_thermo_f64 *tps, int32_t n_modes,
    int64_t n, const double *T, const double *p, const double *w, const double *q_tot, const double *q_liq, const double *q_ice,
    const double *N_liq, const double *N_ice, const double *const *r_dry, const double *const *stdev, const double *const *N_mode,
    const double *const *hygroscopicity, const double *const *molar_mass, double *const *N_act, double *const *M_act, double *S_max,
    void *stream);

/* ---------------------------------------------------------------------------
 * (7) P3 ice scheme: state construction, size-distribution shape solver, mass-weighted mean diameter.
 *
 * Replaces, per point,
 *   state = P3.state_from_prognostic(params, ρq_ice, ρn_ice, ρq_rim, ρb_rim)    src/P3_particle_properties.jl:101-106
 *           (or P3.P3State(params, ρq_ice, ρn_ice, F_rim, ρ_rim) with CMX_P3_INPUT_IS_STATE, :43-56)
 *   logλ  = P3.get_distribution_logλ(state)                                      src/P3_size_distribution.jl:284-320
 *   D_m   = P3.D_m(state, logλ)                                                  src/P3_integral_properties.jl:56-61
 *   logN₀ = P3.get_logN₀(ρn_ice, μ(logλ), logλ)                                  src/P3_size_distribution.jl:233-237
 * (KA wrappers test_P3_get_distribution_logλ_kernel!, benchmark_p3_kernel!, test/gpu_tests.jl:436-451,
 * test/gpu_performance.jl:59-67).  Compute-bound: ≈12 evaluations of the shape residual per point, each 8 incomplete-
 * gamma evaluations of 20/30 fixed iterations (src/Utilities.jl:93-144).  The root is bracketed on logλ ∈ [2, 17] with
 * Brent's method like the reference (same end-point fallbacks, :295-297); logλ = −Inf when ρn_ice or ρq_ice < eps(FT).
 * The solver is Brent's zeroin under a fixed evaluation budget: RootSolvers.jl is not vendored, and zeroin is the restatement
 * that satisfies the reference's own warm-start suite (test/p3_shape_solver_warmstart_tests.jl).
 * brent_iters ≤ 0 selects the reference's fixed budget (8 Float32 / 10 Float64 iterations, :311); a larger value runs that many.
 * THE BUDGET DOES NOT ALWAYS CONVERGE: over 1e6 random states (L_ice 1e-6…1e-3 kg/m³, N_ice 1e2…1e6 m⁻³) the Float64 budget leaves
 * logλ more than 1e-6 from the root for 6.8 % of them and D_m more than 1 % off for 3.5 % — every one of them in 8 ≤ logλ < 11, where
 * the SlopePowerLaw's μ(λ) ramps (21 % of the states of that band).  brent_iters = 12 / 14 / 16 / 20 leave 2.2 % / 0.27 % / 0.02 % /
 * 0.001 % (profiles/r06_brent_exposure.json).  Whether the reference is equally unconverged there depends on RootSolvers' iterates;
 * on the reference's own state sweeps (warm-start, robustness, round trip) its budget converges, here too.
 * log_lambda_guess (nullable): the reference's optional warm start — the guess, when finite, strictly inside the
 * bracket and with a finite residual, replaces the bracket end on its side of the root (_narrow_bracket, :336-353).
 * Output columns may be NULL.
 * ------------------------------------------------------------------------- */
#define CMX_P3_INPUT_IS_STATE   (1u << 0)   /* columns 3, 4 are (F_rim, ρ_rim) instead of (ρq_rim, ρb_rim) */
#define CMX_P3_SLOPE_CONSTANT   (1u << 1)   /* SlopeConstant (μ = mu_const) instead of SlopePowerLaw */
#define CMX_P3_NO_ASPECT_RATIO  (1u << 2)   /* CMP.NoAspectRatio() instead of the default CMP.Oblate() (velocities) */
#define CMX_FREEZE_CLOUD_PSD     (1u << 4)   /* cmx_liquid_freezing_rate_*: cloud (generalized gamma) PSD instead of the rain PSD */
#define CMX_P3_RAIN_PDF_LIMITED (1u << 3)   /* rain_pdf is RainParticlePDF_SB2006_limited (collisions, rain freezing, 2M+P3 entry) */

int32_t cmx_p3_shape_f32(const cmx_p3_params_f32 *params, uint32_t flags, int32_t brent_iters, int64_t n, const float *rho_q_ice,
                         const float *rho_n_ice, const float *x3, const float *x4, const float *log_lambda_guess,
                         float *F_rim, float *rho_rim, float *log_lambda, float *D_m, float *log_N0, void *stream);
int32_t cmx_p3_shape_f64(const cmx_p3_params_f64 *params, uint32_t flags, int32_t brent_iters, int64_t n, const double *rho_q_ice,
                         const double *rho_n_ice, const double *x3, const double *x4, const double *log_lambda_guess,
                         double *F_rim, double *rho_rim, double *log_lambda, double *D_m, double *log_N0, void *stream);

/* P3 number- and mass-weighted ice fall speeds.  Replaces, per point,
 *   v_n = P3.ice_terminal_velocity_number_weighted(vel, ρₐ, state, logλ; p, quad)   src/P3_terminal_velocity.jl:72-91
 *   v_m = P3.ice_terminal_velocity_mass_weighted(vel, ρₐ, state, logλ; p, quad)     src/P3_terminal_velocity.jl:118-137
 * (and the *_from_prognostic wrappers :152-178 when CMX_P3_INPUT_IS_STATE is not set): the integrals
 * ∫ n(D) v(D) [m(D)] dD over the four mass-regime segments between the p and 1−p quantiles of the size distribution
 * (integral_bounds, src/P3_integral_properties.jl:34-46 → UT.gamma_inc_inv, src/Utilities.jl:205-252), with the
 * piecewise small/large-ice Chen-2022 particle velocity (src/Common.jl:304-350,381-382; ρᵢ = 916.7 as hard-wired at
 * src/P3_terminal_velocity.jl:41) times the aspect-ratio factor cbrt(ϕᵢ) (src/P3_particle_properties.jl:402-475) unless
 * CMX_P3_NO_ASPECT_RATIO.  log_lambda is an INPUT here, exactly as in the reference's signatures (use cmx_p3_shape_*
 * to produce it).  Either output may be NULL.  Points with ρn_ice or ρq_ice < eps(FT) give 0. */
int32_t cmx_p3_terminal_velocities_f32(const cmx_p3_params_f32 *params, const cmx_chen2022_ice_vel_f32 *vel,
                                       const cmx_quadrature_f32 *quad, uint32_t flags, float p, int64_t n,
                                       const float *rho_q_ice, const float *rho_n_ice, const float *x3, const float *x4,
                                       const float *rho_air, const float *log_lambda, float *v_n, float *v_m, void *stream);
int32_t cmx_p3_terminal_velocities_f64(const cmx_p3_params_f64 *params, const cmx_chen2022_ice_vel_f64 *vel,
                                       const cmx_quadrature_f64 *quad, uint32_t flags, double p, int64_t n,
                                       const double *rho_q_ice, const double *rho_n_ice, const double *x3, const double *x4,
                                       const double *rho_air, const double *log_lambda, double *v_n, double *v_m, void *stream);

/* BASELINE config 5 as ONE launch: cmx_p3_shape_* (log λ, D_m) followed by cmx_p3_terminal_velocities_* on the same columns, without the
 * log λ round trip through HBM and without reading the state twice (36 → 28 B/point of traffic in f32; the pass is compute-bound, so
 * the gain is the launch and the re-read, not bandwidth).  Same arithmetic as the two separate entries, bit for bit.  log_lambda_guess,
 * log_lambda and D_m may be NULL; brent_iters = 0 keeps the reference's fixed budget. */
int32_t cmx_p3_shape_terminal_velocities_f32(const cmx_p3_params_f32 *params, const cmx_chen2022_ice_vel_f32 *vel,
                                             const cmx_quadrature_f32 *quad, uint32_t flags, int32_t brent_iters, float p, int64_t n,
                                             const float *rho_q_ice, const float *rho_n_ice, const float *x3, const float *x4,
                                             const float *rho_air, const float *log_lambda_guess, float *log_lambda, float *D_m,
                                             float *v_n, float *v_m, void *stream);
int32_t cmx_p3_shape_terminal_velocities_f64(const cmx_p3_params_f64 *params, const cmx_chen2022_ice_vel_f64 *vel,
                                             const cmx_quadrature_f64 *quad, uint32_t flags, int32_t brent_iters, double p, int64_t n,
                                             const double *rho_q_ice, const double *rho_n_ice, const double *x3, const double *x4,
                                             const double *rho_air, const double *log_lambda_guess, double *log_lambda, double *D_m,
                                             double *v_n, double *v_m, void *stream);

/* P3 melting rate (QIMLT of Morrison & Milbrandt 2015): replaces, per point,
 *   (; dNdt, dLdt) = P3.ice_melt(vel, aps, tps, T, ρₐ, state, logλ; quad)                  src/P3_processes.jl:64-94
 * dL/dt = max(0, 4 K_therm / L_f(T) · (T − T_freeze) ∫ ∂m/∂D · F_v(D) · N′(D)/D dD) over the same bounds / segments as the fall
 * speeds, F_v = a_v + b_v ∛(ν/D_v) √(D v(D)/ν) (CO.ventilation_factor, src/Common.jl:506-514; vent = params.vent, the SB2006
 * ventilation coefficients); dN/dt = N/L · dL/dt.  Points with ρn_ice or ρq_ice < eps(FT) give 0 (the reference calls
 * ice_melt only where ice is present, BMT:966). */
int32_t cmx_p3_ice_melt_f32(const cmx_p3_params_f32 *params, const cmx_chen2022_ice_vel_f32 *vel, const cmx_air_properties_f32 *aps,
                            const cmx_thermo_f32 *tps, const cmx_ventilation_f32 *vent, const cmx_quadrature_f32 *quad, uint32_t flags,
                            float p, int64_t n, const float *rho_q_ice, const float *rho_n_ice, const float *x3, const float *x4,
                            const float *rho_air, const float *T, const float *log_lambda, float *dNdt, float *dLdt, void *stream);
int32_t cmx_p3_ice_melt_f64(const cmx_p3_params_f64 *params, const cmx_chen2022_ice_vel_f64 *vel, const cmx_air_properties_f64 *aps,
                            const cmx_thermo_f64 *tps, const cmx_ventilation_f64 *vent, const cmx_quadrature_f64 *quad, uint32_t flags,
                            double p, int64_t n, const double *rho_q_ice, const double *rho_n_ice, const double *x3, const double *x4,
                            const double *rho_air, const double *T, const double *log_lambda, double *dNdt, double *dLdt, void *stream);

/* P3 ice self-collection (aggregation): replaces, per point,
 *   (; dNdt) = P3.ice_self_collection(state, logλ, vel, ρₐ; quad)                          src/P3_processes.jl:676-712
 * (KA kernels test_P3_ice_self_collection_kernel!, test/gpu_tests.jl:444-451, and the reference's own P3 benchmark kernel
 * benchmark_p3_kernel!, test/gpu_performance.jl:59-67).  dN/dt ≥ 0 is the loss rate of ice number [1/m³/s].  8·quad.n²
 * integrand evaluations per point. */
int32_t cmx_p3_ice_self_collection_f32(const cmx_p3_params_f32 *params, const cmx_chen2022_ice_vel_f32 *vel, const cmx_quadrature_f32 *quad,
                                       uint32_t flags, int64_t n, const float *rho_q_ice, const float *rho_n_ice, const float *x3,
                                       const float *x4, const float *rho_air, const float *log_lambda, float *dNdt, void *stream);
int32_t cmx_p3_ice_self_collection_f64(const cmx_p3_params_f64 *params, const cmx_chen2022_ice_vel_f64 *vel, const cmx_quadrature_f64 *quad,
                                       uint32_t flags, int64_t n, const double *rho_q_ice, const double *rho_n_ice, const double *x3,
                                       const double *x4, const double *rho_air, const double *log_lambda, double *dNdt, void *stream);

/* P3 heterogeneous (immersion) freezing: replaces, per point,
 *   (; dNdt, dLdt) = P3.het_ice_nucleation(aerosol, tps, q_lcl, N_lcl, RH, T, ρₐ)              src/P3_processes.jl:20-46
 * J = CM_HetIce.ABIFM_J(aerosol, RH − a_w_ice(T)) on an assumed aerosol surface of 1e-10 m² per droplet; dNdt = max(0, J A N_lcl)
 * [1/m³/s], dLdt = max(0, J A q_lcl ρₐ) [kg/m³/s]; a non-finite J gives 0.  Either output may be NULL. */
int32_t cmx_p3_het_ice_nucleation_f32(const cmx_abifm_dust_f32 *dust, const cmx_thermo_f32 *tps, int64_t n, const float *q_lcl,
                                      const float *N_lcl, const float *RH, const float *T, const float *rho_air, float *dNdt, float *dLdt,
                                      void *stream);
int32_t cmx_p3_het_ice_nucleation_f64(const cmx_abifm_dust_f64 *dust, const cmx_thermo_f64 *tps, int64_t n, const double *q_lcl,
                                      const double *N_lcl, const double *RH, const double *T, const double *rho_air, double *dNdt,
                                      double *dLdt, void *stream);

/* Bigg (1953) immersion freezing of liquid drops: replaces, per point,
 *   (; ∂ₜn_frz, ∂ₜq_frz) = CMI_het.liquid_freezing_rate(rf, pdf, tps, q, ρ, N, T)          src/IceNucleation.jl:274-311 (rain PSD),
 *                                                                                            :355-389 (cloud PSD)
 * (KA kernel test_rain_freezing_kernel!, test/gpu_tests.jl:463-468).  ice->rain_freezing, ice->rain_pdf / ice->cloud_pdf are read;
 * flags: CMX_FREEZE_CLOUD_PSD selects the cloud method, CMX_P3_RAIN_PDF_LIMITED the limited rain PSD.  N is per m³; the rates are per
 * kg of air.  Either output may be NULL. */
int32_t cmx_liquid_freezing_rate_f32(const cmx_p3_ice_params_f32 *ice, const cmx_thermo_f32 *tps, uint32_t flags, int64_t n, const float *q,
                                     const float *rho, const float *N, const float *T, float *dn_frz, float *dq_frz, void *stream);
int32_t cmx_liquid_freezing_rate_f64(const cmx_p3_ice_params_f64 *ice, const cmx_thermo_f64 *tps, uint32_t flags, int64_t n, const double *q,
                                     const double *rho, const double *N, const double *T, double *dn_frz, double *dq_frz, void *stream);

/* (8) P3 liquid–ice collisions: replaces, per point,
 *   rates   = P3.∫liquid_ice_collisions(state, logλ, psd_c, psd_r, L_c, N_c, L_r, N_r, aps, tps, vel, ρₐ, T, m_liq; quad)
 *                                                                                         src/P3_processes.jl:527-562
 *   sources = P3.bulk_liquid_ice_collision_sources(state, logλ, psd_c, psd_r, L_c, N_c, L_r, N_r, aps, tps, vel, ρₐ, T; quad)
 *                                                                                         src/P3_processes.jl:600-655
 * (called from the 2M+P3 fused entry, BMT:962-972).  `ice` carries scheme (+ vent, ρ_rim_local), the Chen-2022 rain and
 * ice fall-speed tables, cloud_pdf and rain_pdf (flags & CMX_P3_RAIN_PDF_LIMITED selects the limited rain PSD); the
 * quadrature rule is the explicit `quad` (ice->quad is what the fused entry passes).  The rain inner integral uses the
 * reference's closed form (closed_rain_inner_NM :343-369: crossover diameter by a fixed-budget Brent solve, incomplete-gamma
 * moments), the cloud inner integral and both rime-volume integrals the quadrature rule.
 * sources[7] = device columns (∂ₜq_c, ∂ₜq_r, ∂ₜN_c, ∂ₜN_r, ∂ₜL_rim, ∂ₜL_ice, ∂ₜB_rim), rates[10] = device columns (QCFRZ, QCSHD,
 * NCCOL, QRFRZ, QRSHD, NRCOL, ∫M_col, BCCOL, BRCOL, ∫𝟙_wet M_col); either array (host array of device pointers) or any
 * entry may be NULL.  Points with ρn_ice or ρq_ice < eps(FT) give 0. */
int32_t cmx_p3_liquid_ice_collisions_f32(const cmx_p3_ice_params_f32 *ice, const cmx_air_properties_f32 *aps, const cmx_thermo_f32 *tps,
                                         const cmx_quadrature_f32 *quad, uint32_t flags, int64_t n, const float *rho_q_ice,
                                         const float *rho_n_ice, const float *x3, const float *x4, const float *L_c, const float *N_c,
                                         const float *L_r, const float *N_r, const float *rho_air, const float *T, const float *log_lambda,
                                         float *const *sources, float *const *rates, void *stream);
int32_t cmx_p3_liquid_ice_collisions_f64(const cmx_p3_ice_params_f64 *ice, const cmx_air_properties_f64 *aps, const cmx_thermo_f64 *tps,
                                         const cmx_quadrature_f64 *quad, uint32_t flags, int64_t n, const double *rho_q_ice,
                                         const double *rho_n_ice, const double *x3, const double *x4, const double *L_c, const double *N_c,
                                         const double *L_r, const double *N_r, const double *rho_air, const double *T, const double *log_lambda,
                                         double *const *sources, double *const *rates, void *stream);

/* (9) 2M + P3 fused entry: replaces, per point,
 *   bulk_microphysics_tendencies(Microphysics2Moment(), mp::Microphysics2MParams{WR, P3IceParams}, tps, ρ, T, q_tot, q_lcl, n_lcl,
 *                                q_rai, n_rai, q_ice, n_ice, q_rim, b_rim, logλ, inpc_log_shift)          BMT:898-1083
 * i.e. warm rain with the ice content in the vapour budget (:942), liquid–ice collisions, aggregation and melting where
 * q_ice > ϵ and n_ice > ϵ (:959-995), Frostenberg-2023 deposition nucleation (:1004-1010), F23-capped Bigg immersion freezing
 * of cloud drops (:1013-1034), ice sublimation / deposition (:1037-1054), ice number adjustment (:1057-1064) and Bigg freezing
 * of rain (:1067-1075).  log_lambda is an INPUT (the host model caches it, cmx_p3_shape_* produces it); inpc_log_shift may be
 * NULL (= 0); the optional w, p arguments of the reference are unused by it (aerosol activation is not wired in, :729).
 * tendencies[8] = device columns (dq_lcl_dt, dn_lcl_dt, dq_rai_dt, dn_rai_dt, dq_ice_dt, dn_ice_dt, dq_rim_dt, db_rim_dt), all
 * required; the ninth field of the reference's NamedTuple, dn_lcl_activation_dt, is identically 0.
 * flags: CMX_P3_RAIN_PDF_LIMITED (is_limited of both the SB2006 set and P3IceParams.rain_pdf), CMX_P3_SLOPE_CONSTANT,
 * CMX_P3_NO_ASPECT_RATIO.  Two launches on `stream`: a pointwise kernel that writes the columns, then the 8-lanes-per-point
 * quadrature kernel that adds the ice-process terms to them. */
int32_t cmx_microphysics_2m_p3_tendencies_f32(const cmx_warm_rain_2m_f32 *warm_rain, const cmx_p3_ice_params_f32 *ice, const cmx_thermo_f32 *tps,
                                              uint32_t flags, int64_t n, const float *rho, const float *T, const float *q_tot, const float *q_lcl,
                                              const float *n_lcl, const float *q_rai, const float *n_rai, const float *q_ice, const float *n_ice,
                                              const float *q_rim, const float *b_rim, const float *log_lambda, const float *inpc_log_shift,
                                              float *const *tendencies, void *stream);
int32_t cmx_microphysics_2m_p3_tendencies_f64(const cmx_warm_rain_2m_f64 *warm_rain, const cmx_p3_ice_params_f64 *ice, const cmx_thermo_f64 *tps,
                                              uint32_t flags, int64_t n, const double *rho, const double *T, const double *q_tot, const double *q_lcl,
                                              const double *n_lcl, const double *q_rai, const double *n_rai, const double *q_ice, const double *n_ice,
                                              const double *q_rim, const double *b_rim, const double *log_lambda, const double *inpc_log_shift,
                                              double *const *tendencies, void *stream);

/* … and on the host model's own storage, like cmx_sb2006_warm_rain_tendencies_fields_* and cmx_mp1m_*_fields_*: every column is n_seg
 * runs of seg_len contiguous elements with its own run stride (a component of a ClimaCore VIJFH field in place: seg_len = Nv·Ni·Nj, stride =
 * Nv·Ni·Nj·Nf).  in[13] = (rho, T, q_tot, q_lcl, n_lcl, q_rai, n_rai, q_ice, n_ice, q_rim, b_rim, log_lambda, inpc_log_shift — the last may be
 * NULL), in_seg_stride[13]; out[8] = the eight tendency columns in the order of cmx_microphysics_2m_p3_tendencies_*, out_seg_stride[8].  Strides
 * are in elements and >= seg_len.  Bit-identical to the SoA entry on the same states.  (No array-of-rows output form.) */
int32_t cmx_microphysics_2m_p3_tendencies_fields_f32(const cmx_warm_rain_2m_f32 *warm_rain, const cmx_p3_ice_params_f32 *ice, const cmx_thermo_f32 *tps,
                                                     uint32_t flags, int64_t n_seg, int64_t seg_len, const float *const *in,
                                                     const int64_t *in_seg_stride, float *const *out, const int64_t *out_seg_stride, void *stream);
int32_t cmx_microphysics_2m_p3_tendencies_fields_f64(const cmx_warm_rain_2m_f64 *warm_rain, const cmx_p3_ice_params_f64 *ice, const cmx_thermo_f64 *tps,
                                                     uint32_t flags, int64_t n_seg, int64_t seg_len, const double *const *in,
                                                     const int64_t *in_seg_stride, double *const *out, const int64_t *out_seg_stride, void *stream);

/* UT.gamma_inc(a, x) = (P, Q) and UT.gamma_inc_inv(a, p, q) over columns — src/Utilities.jl:54-61,93-144 and :205-252 (KA wrapper
 * test_gamma_inc_kernel!, test/gpu_tests.jl:456-461; CPU test test/gamma_inc_tests.jl): the reference's fast regularised incomplete gamma
 * functions — series for x < a + 1, Lentz continued fraction otherwise, 20 (Float32) / 30 (Float64) terms — and their Halley inverse,
 * evaluated by the SAME device routines the P3 kernels call (shape solver moments, quantile bounds, closed-form rain collisions).  a > 0.
 * Either of P, Q may be NULL (not both).  The device series / continued fraction may stop early once converged to eps(FT); the reference's
 * fixed term count is an upper bound, so results agree with it to rounding wherever the reference's truncation has converged. */
int32_t cmx_gamma_inc_f32(int64_t n, const float *a, const float *x, float *P, float *Q, void *stream);
int32_t cmx_gamma_inc_f64(int64_t n, const double *a, const double *x, double *P, double *Q, void *stream);
int32_t cmx_gamma_inc_inv_f32(int64_t n, const float *a, const float *p, const float *q, float *x, void *stream);
int32_t cmx_gamma_inc_inv_f64(int64_t n, const double *a, const double *p, const double *q, double *x, void *stream);

/* Size-distribution helpers over columns (VERDICT r03 "missing" 5) — the reference's DistributionTools and the SB2006 PSD accessors:
 *   DT.generalized_gamma_quantile(ν, μ, B, Y) = (UT.gamma_inc_inv((ν+1)/μ, Y, 1−Y)/B)^(1/μ)      src/DistributionTools.jl:44-47
 *   DT.generalized_gamma_cdf(ν, μ, B, x)      = P((ν+1)/μ, B x^μ), 0 for x ≤ 0                    :75-82
 *   DT.exponential_quantile(D_mean, Y)        = exp(log D_mean + cloglog(Y))                     :146-151
 *   DT.exponential_cdf(D_mean, D)             = exp(log1mexp(−D/D_mean)), 0 for D < 0             :124-129
 * (test/DistributionTools_tests.jl), ν and μ shared by the call, B / D_mean per point.  Where the scalar functions throw a DomainError
 * (μ ≤ 0, B ≤ 0, D_mean ≤ 0, Y outside [0, 1]) the array entries write NaN.  `quantile` needs Y, `cdf` needs x (resp. D); either output may
 * be NULL, not both. */
int32_t cmx_generalized_gamma_f32(float nu, float mu, int64_t n, const float *B, const float *Y, const float *x, float *quantile, float *cdf, void *stream);
int32_t cmx_generalized_gamma_f64(double nu, double mu, int64_t n, const double *B, const double *Y, const double *x, double *quantile, double *cdf,
                                  void *stream);
int32_t cmx_exponential_distribution_f32(int64_t n, const float *D_mean, const float *Y, const float *D, float *quantile, float *cdf, void *stream);
int32_t cmx_exponential_distribution_f64(int64_t n, const double *D_mean, const double *Y, const double *D, double *quantile, double *cdf, void *stream);
/*   CM2.size_distribution_value(pdf, q, ρₐ, N, D)            n(D) of the rain (N₀r e^(−D/D̄r)) or cloud (N₀c D^(3ν+2) e^(−λc D^(3μ))) PSD      src/Microphysics2M.jl:270-315
 *   CM2.get_size_distribution_bounds(pdf, q, ρₐ, N, p)        the p and 1 − p quantiles of that PSD (the reference's default p = eps(FT))     :336-354
 * flags: CMX_PSD_CLOUD selects pdf_c (CloudParticlePDF_SB2006; pdf_r may be NULL), otherwise pdf_r (CMX_SB2006_LIMITED: the limited rain PSD;
 * pdf_c may be NULL).  N per m³.  D is needed for n_D only; any output may be NULL (not all three).  With CMX_SB2006_LIMITED the limiter
 * pairs of pdf_r must be positive and ordered (xr_min ≤ xr_max, N0_min ≤ N0_max, lambda_min ≤ lambda_max) as for the rate entries: a struct of
 * the not-limited variant (limiters zero) passed with the flag set returns CMX_ERR_BAD_ARG instead of clamping N₀ and λ to 0. */
#define CMX_PSD_CLOUD (1u << 5)
int32_t cmx_sb2006_size_distribution_f32(const cmx_cloud_pdf_sb2006_f32 *pdf_c, const cmx_rain_pdf_sb2006_f32 *pdf_r, uint32_t flags, float p, int64_t n,
                                         const float *q, const float *rho, const float *N, const float *D, float *n_D, float *D_min, float *D_max,
                                         void *stream);
int32_t cmx_sb2006_size_distribution_f64(const cmx_cloud_pdf_sb2006_f64 *pdf_c, const cmx_rain_pdf_sb2006_f64 *pdf_r, uint32_t flags, double p, int64_t n,
                                         const double *q, const double *rho, const double *N, const double *D, double *n_D, double *D_min, double *D_max,
                                         void *stream);

/* ---------------------------------------------------------------------------
 * (0) 0-moment entry of bulk_microphysics_tendencies (src/BulkMicrophysicsTendencies.jl:658-680; KA kernels
 * test_bulk_tendencies_0m_kernel!, test_bulk_tendencies_0m_S0_kernel!, test/gpu_tests.jl:364-383, and
 * test_0_moment_micro_kernel! / test_0M_derivatives_kernel!, test/gpu_tests.jl:105-141).  Replaces the broadcasts
 *   BMT.bulk_microphysics_tendencies.(Microphysics0Moment(), mp, tps, T, q_lcl, q_icl[, q_vap_sat])
 *   CM0.remove_precipitation.(p0m, q_lcl, q_icl[, q_vap_sat])   CM0.∂remove_precipitation_∂q_tot.(…)   src/Microphysics0M.jl:35-75
 * dq_tot_dt = −max(0, q_lcl⁺ + q_icl⁺ − threshold)/τ_precip with threshold = qc_0 (q_vap_sat == NULL) or S_0·q_vap_sat;
 * ddq_dq_tot (optional) = −1/τ_precip where condensate exceeds the threshold, else 0.  Negative q_lcl / q_icl are clamped to 0 as BMT
 * does (a no-op for the non-negative inputs CM0's direct callers pass).  T is not read by the reference and is not an argument.
 * ------------------------------------------------------------------------- */
int32_t cmx_mp0m_tendencies_f32(const cmx_parameters_0m_f32 *p, int64_t n, const float *q_lcl, const float *q_icl,
                                const float *q_vap_sat, float *dq_tot_dt, float *ddq_dq_tot, void *stream);
int32_t cmx_mp0m_tendencies_f64(const cmx_parameters_0m_f64 *p, int64_t n, const double *q_lcl, const double *q_icl,
                                const double *q_vap_sat, double *dq_tot_dt, double *ddq_dq_tot, void *stream);

/* ---------------------------------------------------------------------------
 * (10) Cloud diagnostics over columns — src/CloudDiagnostics.jl (round 5; what a host model's radiation / radar diagnostics broadcast over the SAME state
 * columns the tendency entries read):
 *   Z_1m      = CMD.radar_reflectivity_1M(rain, q_rai, ρ)                                    :31-46    [dBZ, clipped at −150]
 *   Z_2m      = CMD.radar_reflectivity_2M(SB2006, q_lcl, q_rai, N_lcl, N_rai, ρ)             :64-84    [dBZ, clipped at −150]
 *   reff_2m   = CMD.effective_radius_2M(SB2006, q_lcl, q_rai, N_lcl, N_rai, ρ)               :100-125  [m]
 *   reff_lh97 = CMD.effective_radius_Liu_Hallet_97(wtr, ρ, q_lcl, N_lcl, q_rai, N_rai)       :143-163  [m]
 * (CMD.effective_radius_const is the r_eff field of cmx_cloud_liquid / cmx_cloud_ice.)  N per m³ as in the reference.  Any output may be NULL (not all
 * four); the parameter structs and columns only a NULL output needs may be NULL: `rain` and q_rai for Z_1m; pdf_c, pdf_r (flags: CMX_SB2006_LIMITED as
 * for the rate entries, same limiter-pair rule) and all four of q_lcl, q_rai, N_lcl, N_rai for Z_2m / reff_2m; rho_w (WaterProperties.ρw) and q_lcl for
 * reff_lh97 — whose three-argument method (N_lcl = 100, no rain: :165-180) is selected by passing N_lcl, q_rai and N_rai all NULL.  The reference's
 * gates are kept (absent species, notvalid(B) = B is 0 or not finite in the float type, M² ≤ ϵ); a NaN input gives NaN.
 * ------------------------------------------------------------------------- */
int32_t cmx_cloud_diagnostics_f32(const cmx_rain_f32 *rain, const cmx_cloud_pdf_sb2006_f32 *pdf_c, const cmx_rain_pdf_sb2006_f32 *pdf_r, float rho_w,
                                  uint32_t flags, int64_t n, const float *rho, const float *q_lcl, const float *q_rai, const float *N_lcl,
                                  const float *N_rai, float *Z_1m, float *Z_2m, float *reff_2m, float *reff_lh97, void *stream);
int32_t cmx_cloud_diagnostics_f64(const cmx_rain_f64 *rain, const cmx_cloud_pdf_sb2006_f64 *pdf_c, const cmx_rain_pdf_sb2006_f64 *pdf_r, double rho_w,
                                  uint32_t flags, int64_t n, const double *rho, const double *q_lcl, const double *q_rai, const double *N_lcl,
                                  const double *N_rai, double *Z_1m, double *Z_2m, double *reff_2m, double *reff_lh97, void *stream);

/* ---------------------------------------------------------------------------
 * (3) Optional diagnostic sums over one rank's shard: Σx of `ncols` (≤ CMX_COLUMN_SUMS_MAX_COLS) device columns of length
 * n, accumulated in double, into `sums[ncols]` (device, double).  ONE launch reduces all columns to CMX_COLUMN_SUMS_PARTIALS partial
 * sums each (`workspace`: ncols · CMX_COLUMN_SUMS_PARTIALS doubles of device memory owned by the caller — the library allocates nothing),
 * a second, one-workgroup-per-column launch adds those in a fixed tree.  No floating-point atomics: the result is a pure function of
 * the column contents and n — bit-identical from run to run and from device to device (the decomposition does not depend on the CU
 * count).  It is NOT invariant under re-sharding: a rank count changes the order of additions, so the all-reduced total of 8 shards and
 * the sum of the unsharded column agree to rounding (a few ulp of Σ|x|·2⁻⁵³·log₂ n), not bit for bit.  The caller all-reduces the ≤ 16
 * doubles over RCCL / MPI; the library itself performs no communication.
 * ------------------------------------------------------------------------- */
#define CMX_COLUMN_SUMS_MAX_COLS 16
#define CMX_COLUMN_SUMS_PARTIALS 1024
int32_t cmx_column_sums_f32(int32_t ncols, const float *const *cols, int64_t n, double *sums, double *workspace, void *stream);
int32_t cmx_column_sums_f64(int32_t ncols, const double *const *cols, int64_t n, double *sums, double *workspace, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CMX_H */
