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
 *  - `_f32` entry points compute in float with the reference's Float32
 *    thresholds (eps(Float32), cbrt(floatmin(Float32)) — src/Utilities.jl:318-340),
 *    `_f64` ones in double with the Float64 thresholds.
 */
#ifndef CMX_H
#define CMX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CMX_VERSION_MAJOR 0
#define CMX_VERSION_MINOR 1

typedef enum cmx_status {
    CMX_OK = 0,
    CMX_ERR_BAD_ARG = -1,     /* null pointer, n < 0, unknown flag combination */
    CMX_ERR_HIP = -2,         /* a HIP runtime call failed: see cmx_last_hip_error() */
    CMX_ERR_UNSUPPORTED = -3  /* valid request this build has no kernel for */
} cmx_status;

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
            T_freeze;                                                                          \
    } cmx_thermo_##SFX;                                                                        \
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
    typedef struct cmx_abifm_dust_##SFX { FT ABIFM_m, ABIFM_c; } cmx_abifm_dust_##SFX;

CMX_DECLARE_PARAM_STRUCTS(float, f32)
CMX_DECLARE_PARAM_STRUCTS(double, f64)

/* ---------------------------------------------------------------------------
 * Library / device queries
 * ------------------------------------------------------------------------- */
/* (major << 16) | minor */
int32_t cmx_version(void);
/* text of the last HIP error seen by this thread ("" if none); static storage */
const char *cmx_last_hip_error(void);

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

/* CO.a_w_ice(tps, T) and CO.a_w_eT(tps, e, T) over columns — src/Common.jl:250-253,267-271
 * (KA wrappers Common_a_w_ice_kernel!, Common_a_w_eT_kernel!, test/gpu_tests.jl:340-362).
 * `e` may be NULL iff `a_w_eT` is NULL. */
int32_t cmx_water_activity_f32(const cmx_thermo_f32 *tps, int64_t n, const float *T, const float *e,
                               float *a_w_ice, float *a_w_eT, void *stream);
int32_t cmx_water_activity_f64(const cmx_thermo_f64 *tps, int64_t n, const double *T, const double *e,
                               double *a_w_ice, double *a_w_eT, void *stream);

/* ---------------------------------------------------------------------------
 * (3) Optional diagnostic sums over one rank's shard: per-column
 * Σx (double accumulation) of `ncols` device columns of length n into
 * `sums[ncols]` (device, double).  The caller all-reduces the ≤16 doubles over
 * RCCL; the library itself performs no communication.
 * ------------------------------------------------------------------------- */
int32_t cmx_column_sums_f32(int32_t ncols, const float *const *cols, int64_t n,
                            double *sums, void *stream);
int32_t cmx_column_sums_f64(int32_t ncols, const double *const *cols, int64_t n,
                            double *sums, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CMX_H */
