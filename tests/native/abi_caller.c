/*
 * abi_caller.c — a plain-C (C11, gcc) caller of libcmx.so through include/cmx.h.
 *
 * The reference-side binding is a Julia `ccall` shim that cannot be executed in this image (no
 * julia); this program is the next best witness that the header is a usable *caller* contract: it includes cmx.h as C,
 * fills every parameter struct from literals (abi_caller_params.h — no Python host mirror involved), allocates device
 * columns with the HIP C API, calls the entry points and prints the outputs as one JSON object.  tests/test_abi_caller.py
 * compiles it (CPU: compile + link only), runs it on the GPU box and compares the output with the reference's own
 * known-answer values (tests/golden/, test/gpu_tests.jl:608-630,821-872), the oracle, and the ctypes path.
 *
 *   gcc -std=c11 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include tests/native/abi_caller.c \
 *       -L cloudmicrophysics.jl_amd/csrc -lcmx -L /opt/rocm/lib -lamdhip64 -o abi_caller
 */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "abi_caller_params.h"

/* the layout contract, asserted again from the caller's side (C11 _Static_assert; cmx.h carries the full list) */
_Static_assert(sizeof(cmx_thermo_f32) == 13 * 4 && sizeof(cmx_thermo_f64) == 13 * 8, "cmx_thermo_*: 13 fields (cv_l is the 13th)");
_Static_assert(sizeof(cmx_warm_rain_2m_f32) == 50 * 4 && sizeof(cmx_warm_rain_2m_f64) == 50 * 8, "cmx_warm_rain_2m_*: 50 fields");
_Static_assert(sizeof(cmx_microphysics_1m_f64) == 90 * 8, "cmx_microphysics_1m_*: 90 fields");
_Static_assert(sizeof(cmx_rain_vel_f64) == 19 * 8, "cmx_rain_vel_*: 7 + 12 fields");

#define N 8   /* identical points, as the reference's KA tests launch ndrange = 10 copies (test/gpu_tests.jl:825-833) */

#define HIP_OK(call)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #call, hipGetErrorString(e_)); exit(2); } \
    } while (0)
#define CMX_CALL(call)                                                                                       \
    do {                                                                                                     \
        int32_t s_ = (call);                                                                                 \
        if (s_ != CMX_OK) { fprintf(stderr, "%s -> status %d (%s)\n", #call, (int)s_, cmx_last_hip_error()); exit(3); } \
    } while (0)

static double *dev_const(double v) {
    double h[N], *d = NULL;
    for (int i = 0; i < N; ++i) h[i] = v;
    HIP_OK(hipMalloc((void **)&d, sizeof h));
    HIP_OK(hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice));
    return d;
}
static double *dev_out(void) {
    double *d = NULL;
    HIP_OK(hipMalloc((void **)&d, N * sizeof(double)));
    HIP_OK(hipMemset(d, 0xff, N * sizeof(double)));   /* NaN pattern: an unwritten column cannot pass */
    return d;
}
/* copies a column back, checks all N copies are bit-identical, returns element 0 */
static double fetch(const double *d) {
    double h[N];
    HIP_OK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
    for (int i = 1; i < N; ++i)
        if (memcmp(&h[i], &h[0], sizeof(double)) != 0) { fprintf(stderr, "copies of one state differ\n"); exit(4); }
    return h[0];
}
static void emit(const char *name, double v, int last) { printf("    \"%s\": %.17g%s\n", name, v, last ? "" : ","); }

int main(void) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { fprintf(stderr, "no HIP device\n"); return 77; }
    printf("{\n  \"cmx_version\": %d,\n", (int)cmx_version());

    /* ---- (1) per-process SB2006 rates on the reference's KAT state, test/gpu_tests.jl:827-833 ------------------ */
    double *T = dev_const(290.0), *q_tot = dev_const(7e-3), *q_lcl = dev_const(2e-3), *q_rai = dev_const(5e-4);
    double *rho = dev_const(1.2), *N_lcl = dev_const(1e8), *N_rai = dev_const(1e7);
    static const char *const proc_names[CMX_SB2006_NPROC] = {
        "acnv_dq_lcl_dt", "acnv_dN_lcl_dt", "acnv_dq_rai_dt", "acnv_dN_rai_dt", "lcl_self_collection", "accr_dq_lcl_dt", "accr_dN_lcl_dt",
        "accr_dq_rai_dt", "rain_self_collection", "rain_breakup", "rain_vel_n", "rain_vel_m", "evap_dN_rai_dt", "evap_dq_rai_dt",
        "numadj_rai", "numadj_lcl", "condevap"};
    double *proc[CMX_SB2006_NPROC];
    for (int k = 0; k < CMX_SB2006_NPROC; ++k) proc[k] = dev_out();
    for (int limited = 1; limited >= 0; --limited) {
        uint32_t flags = (limited ? CMX_SB2006_LIMITED : 0u) | CMX_VEL_SB2006;
        CMX_CALL(cmx_sb2006_process_rates_f64(&WARM_RAIN_2M, &THERMO, &RAIN_VEL, flags, N, q_tot, q_lcl, q_rai, N_lcl, N_rai, rho, T,
                                              proc, NULL));
        HIP_OK(hipDeviceSynchronize());
        printf("  \"process_rates_%s\": {\n", limited ? "limited" : "notlimited");
        for (int k = 0; k < CMX_SB2006_NPROC; ++k) emit(proc_names[k], fetch(proc[k]), k == CMX_SB2006_NPROC - 1);
        printf("  },\n");
    }

    /* ---- (2) the north-star fused entry on the same state (n per kg of air = N / ρ, as in BMT:828-837) ---------- */
    double *n_lcl = dev_const(1e8 / 1.2), *n_rai = dev_const(1e7 / 1.2);
    double *o[6];
    for (int k = 0; k < 6; ++k) o[k] = dev_out();
    CMX_CALL(cmx_sb2006_warm_rain_tendencies_f64(&WARM_RAIN_2M, &THERMO, &RAIN_VEL, CMX_SB2006_LIMITED | CMX_VEL_SB2006, N, rho, T, q_tot,
                                                 q_lcl, n_lcl, q_rai, n_rai, o[0], o[1], o[2], o[3], o[4], o[5], NULL));
    HIP_OK(hipDeviceSynchronize());
    static const char *const fused_names[6] = {"dq_lcl_dt", "dn_lcl_dt", "dq_rai_dt", "dn_rai_dt", "vt_rai_n", "vt_rai_m"};
    printf("  \"warm_rain_tendencies\": {\n");
    for (int k = 0; k < 6; ++k) emit(fused_names[k], fetch(o[k]), k == 5);
    printf("  },\n");

    /* ---- (3) 1-moment Instantaneous tendencies: a mixed-phase state below freezing (all 13 processes live) ------ */
    double *T1 = dev_const(268.0), *qt1 = dev_const(6e-3), *q5 = dev_const(5e-4);
    double *m[4];
    for (int k = 0; k < 4; ++k) m[k] = dev_out();
    CMX_CALL(cmx_mp1m_tendencies_f64(&MICROPHYSICS_1M, &THERMO, MICROPHYSICS_1M_FLAGS, N, rho, T1, qt1, q5, q5, q5, q5, m[0], m[1], m[2],
                                     m[3], NULL));
    HIP_OK(hipDeviceSynchronize());
    static const char *const mp1m_names[4] = {"dq_lcl_dt", "dq_icl_dt", "dq_rai_dt", "dq_sno_dt"};
    printf("  \"mp1m_tendencies\": {\n");
    for (int k = 0; k < 4; ++k) emit(mp1m_names[k], fetch(m[k]), k == 3);
    printf("  },\n");

    /* ---- (4) bulk sedimentation velocities on the reference's KAT state, test/gpu_tests.jl:608-630 -------------- */
    double *rho_s = dev_const(0.95), *ql = dev_const(4e-3), *qi = dev_const(3e-3), *qr = dev_const(2e-3), *qs = dev_const(1e-3);
    double *w[4];
    for (int k = 0; k < 4; ++k) w[k] = dev_out();
    CMX_CALL(cmx_sedimentation_velocities_f64(&MICROPHYSICS_1M, &STOKES_VEL, &RAIN_VEL.chen2022, &CHEN_ICE_VEL, N, rho_s, ql, qi, qr, qs,
                                              w[0], w[1], w[2], w[3], NULL));
    HIP_OK(hipDeviceSynchronize());
    static const char *const sed_names[4] = {"w_lcl", "w_icl", "w_rai", "w_sno"};
    printf("  \"sedimentation_velocities\": {\n");
    for (int k = 0; k < 4; ++k) emit(sed_names[k], fetch(w[k]), k == 3);
    printf("  },\n");

    /* ---- (5) error behaviour seen by a C caller: status codes, never a crash ----------------------------------- */
    int32_t s_null = cmx_sb2006_warm_rain_tendencies_f64(NULL, &THERMO, NULL, CMX_SB2006_LIMITED, N, rho, T, q_tot, q_lcl, n_lcl, q_rai,
                                                         n_rai, o[0], o[1], o[2], o[3], NULL, NULL, NULL);
    int32_t s_neg = cmx_sb2006_warm_rain_tendencies_f64(&WARM_RAIN_2M, &THERMO, NULL, CMX_SB2006_LIMITED, -1, rho, T, q_tot, q_lcl, n_lcl,
                                                        q_rai, n_rai, o[0], o[1], o[2], o[3], NULL, NULL, NULL);
    int32_t s_zero = cmx_sb2006_warm_rain_tendencies_f64(&WARM_RAIN_2M, &THERMO, NULL, CMX_SB2006_LIMITED, 0, rho, T, q_tot, q_lcl, n_lcl,
                                                         q_rai, n_rai, o[0], o[1], o[2], o[3], NULL, NULL, NULL);
    printf("  \"status\": {\"null_params\": %d, \"negative_n\": %d, \"empty\": %d}\n}\n", (int)s_null, (int)s_neg, (int)s_zero);

    double *all[] = {T, q_tot, q_lcl, q_rai, rho, N_lcl, N_rai, n_lcl, n_rai, T1, qt1, q5, rho_s, ql, qi, qr, qs};
    for (size_t k = 0; k < sizeof all / sizeof *all; ++k) HIP_OK(hipFree(all[k]));
    for (int k = 0; k < CMX_SB2006_NPROC; ++k) HIP_OK(hipFree(proc[k]));
    for (int k = 0; k < 6; ++k) HIP_OK(hipFree(o[k]));
    for (int k = 0; k < 4; ++k) { HIP_OK(hipFree(m[k])); HIP_OK(hipFree(w[k])); }
    return 0;
}
