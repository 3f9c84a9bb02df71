// cmx_sb2006_kernels.hip — fused SB2006 warm-rain kernels for gfx950 and their C-ABI entry points.
//
// Kernel shape: pointwise map over structure-of-arrays state columns, HBM-bound.
// One lane owns VEC consecutive points (VEC·sizeof(FT) = 16 B), so each of the 7 input columns is
// read with one global_load_dwordx4 per lane (1 KiB per wave-instruction, fully coalesced) and each
// of the 4–6 output columns written with one global_store_dwordx4; loads and stores carry the
// non-temporal hint (every byte is touched exactly once).  256-thread workgroups, grid-stride over
// CUs × k workgroups.  No LDS, no MFMA: there is no data reuse and no contraction on this path.
#include <hip/hip_runtime.h>

#include "cmx_launch.hpp"
#include "cmx_sb2006.hpp"

namespace cmx {

template <typename FT> struct SbIn { const FT *rho, *T, *q_tot, *q_lcl, *n_lcl, *q_rai, *n_rai; };
template <typename FT> struct SbOut { FT *dq_lcl, *dn_lcl, *dq_rai, *dn_rai, *vt_n, *vt_m; };
template <typename FT> struct SbProcOut { FT *col[CMX_SB2006_NPROC]; };

// bulk_microphysics_tendencies(::Microphysics2Moment, …) over columns — BMT:820-854 + :707-782
template <typename FT, bool LIMITED, int VEL, int VEC>
__global__ __launch_bounds__(kBlock) void sb2006_tendencies_kernel(const SbConsts<FT> c, const SbIn<FT> in,
                                                                   const SbOut<FT> out, const int64_t nvec) {
    using M = Math<FT>;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < nvec; i += stride) {
        FT rho[VEC], T[VEC], q_tot[VEC], q_lcl[VEC], n_lcl[VEC], q_rai[VEC], n_rai[VEC];
        load_col<FT, VEC>(in.rho, i, rho);
        load_col<FT, VEC>(in.T, i, T);
        load_col<FT, VEC>(in.q_tot, i, q_tot);
        load_col<FT, VEC>(in.q_lcl, i, q_lcl);
        load_col<FT, VEC>(in.n_lcl, i, n_lcl);
        load_col<FT, VEC>(in.q_rai, i, q_rai);
        load_col<FT, VEC>(in.n_rai, i, n_rai);
        FT dq_lcl[VEC], dn_lcl[VEC], dq_rai[VEC], dn_rai[VEC], vt_n[VEC], vt_m[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            // clamp_to_nonneg — BMT:828-837 (T is not clamped)
            const FT r_ = M::max(FT(0), rho[k]);
            const FT qt = M::max(FT(0), q_tot[k]);
            const FT ql = M::max(FT(0), q_lcl[k]);
            const FT qr = M::max(FT(0), q_rai[k]);
            const FT nl = M::max(FT(0), n_lcl[k]);
            const FT nr = M::max(FT(0), n_rai[k]);
            // N = ρ n — BMT:718-719
            const SbRates<FT> p = sb2006_point<FT, LIMITED, VEL>(c, r_, T[k], qt, ql, qr, r_ * nl, r_ * nr, nl, nr);
            // accumulation order of warm_rain_tendencies_2m — BMT:738-779
            dq_lcl[k] = (p.cond + p.au_dq_lcl) + p.ac_dq_lcl;
            dn_lcl[k] = ((p.au_dN_lcl * p.inv_rho + p.lsc * p.inv_rho) + p.ac_dN_lcl * p.inv_rho) + p.na_lcl;
            dq_rai[k] = (p.evq + p.au_dq_rai) + p.ac_dq_rai;
            dn_rai[k] = (((p.evN * p.inv_rho + p.au_dN_rai * p.inv_rho) + p.rsc * p.inv_rho) + p.rbr * p.inv_rho) + p.na_rai;
            vt_n[k] = p.vt_n;
            vt_m[k] = p.vt_m;
        }
        store_col<FT, VEC>(out.dq_lcl, i, dq_lcl);
        store_col<FT, VEC>(out.dn_lcl, i, dn_lcl);
        store_col<FT, VEC>(out.dq_rai, i, dq_rai);
        store_col<FT, VEC>(out.dn_rai, i, dn_rai);
        if constexpr (VEL != VEL_NONE) {
            if (out.vt_n) store_col<FT, VEC>(out.vt_n, i, vt_n);
            if (out.vt_m) store_col<FT, VEC>(out.vt_m, i, vt_m);
        }
    }
}

// SB2006_2M_kernel (test/gpu_tests.jl:220-235): the individual process rates, N per m³, no clamping
template <typename FT, bool LIMITED, int VEL>
__global__ __launch_bounds__(kBlock) void sb2006_process_kernel(const SbConsts<FT> c, const FT *__restrict__ q_tot,
                                                                const FT *__restrict__ q_lcl, const FT *__restrict__ q_rai,
                                                                const FT *__restrict__ N_lcl, const FT *__restrict__ N_rai,
                                                                const FT *__restrict__ rho, const FT *__restrict__ T,
                                                                const SbProcOut<FT> out, const int64_t n) {
    using M = Math<FT>;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const FT r_ = rho[i];
        const FT inv_r = M::rcp(r_);
        const SbRates<FT> p = sb2006_point<FT, LIMITED, VEL>(c, r_, T[i], q_tot[i], q_lcl[i], q_rai[i], N_lcl[i],
                                                             N_rai[i], N_lcl[i] * inv_r, N_rai[i] * inv_r);
#define CMX_PUT(colid, v) if (out.col[colid]) out.col[colid][i] = (v)
        CMX_PUT(CMX_SB_ACNV_DQ_LCL, p.au_dq_lcl);
        CMX_PUT(CMX_SB_ACNV_DN_LCL, p.au_dN_lcl);
        CMX_PUT(CMX_SB_ACNV_DQ_RAI, p.au_dq_rai);
        CMX_PUT(CMX_SB_ACNV_DN_RAI, p.au_dN_rai);
        CMX_PUT(CMX_SB_LCL_SELFCOL, p.lsc);
        CMX_PUT(CMX_SB_ACCR_DQ_LCL, p.ac_dq_lcl);
        CMX_PUT(CMX_SB_ACCR_DN_LCL, p.ac_dN_lcl);
        CMX_PUT(CMX_SB_ACCR_DQ_RAI, p.ac_dq_rai);
        CMX_PUT(CMX_SB_RAI_SELFCOL, p.rsc);
        CMX_PUT(CMX_SB_RAI_BREAKUP, p.rbr);
        CMX_PUT(CMX_SB_RAI_VEL_N, p.vt_n);
        CMX_PUT(CMX_SB_RAI_VEL_M, p.vt_m);
        CMX_PUT(CMX_SB_EVAP_DN_RAI, p.evN);
        CMX_PUT(CMX_SB_EVAP_DQ_RAI, p.evq);
        CMX_PUT(CMX_SB_NUMADJ_RAI, p.na_rai);
        CMX_PUT(CMX_SB_NUMADJ_LCL, p.na_lcl);
        CMX_PUT(CMX_SB_CONDEVAP, p.cond);
#undef CMX_PUT
    }
}

// ---- host side -----------------------------------------------------------------------------
static int decode_vel(uint32_t flags, bool want) {
    if (!want) return VEL_NONE;
    const bool sb = flags & CMX_VEL_SB2006, ch = flags & CMX_VEL_CHEN2022;
    if (sb == ch) return -1;   // none or both
    return sb ? VEL_SB : VEL_CHEN;
}

template <typename FT, int VEC>
static void launch_tendencies(bool limited, int vel, const SbConsts<FT> &c, const SbIn<FT> &in, const SbOut<FT> &out,
                              int64_t nvec, hipStream_t s) {
    const int grid = grid_for(nvec);
#define CMX_LAUNCH(L, V) \
    hipLaunchKernelGGL((sb2006_tendencies_kernel<FT, L, V, VEC>), dim3(grid), dim3(kBlock), 0, s, c, in, out, nvec)
    if (limited) {
        if (vel == VEL_NONE) CMX_LAUNCH(true, VEL_NONE);
        else if (vel == VEL_SB) CMX_LAUNCH(true, VEL_SB);
        else CMX_LAUNCH(true, VEL_CHEN);
    } else {
        if (vel == VEL_NONE) CMX_LAUNCH(false, VEL_NONE);
        else if (vel == VEL_SB) CMX_LAUNCH(false, VEL_SB);
        else CMX_LAUNCH(false, VEL_CHEN);
    }
#undef CMX_LAUNCH
}

template <typename FT, typename WR, typename TH, typename VL>
static int32_t tendencies_entry(const WR *wr, const TH *tps, const VL *vel, uint32_t flags, int64_t n, const FT *rho,
                                const FT *T, const FT *q_tot, const FT *q_lcl, const FT *n_lcl, const FT *q_rai,
                                const FT *n_rai, FT *dq_lcl, FT *dn_lcl, FT *dq_rai, FT *dn_rai, FT *vt_n, FT *vt_m,
                                void *stream) {
    if (!wr || !tps || n < 0) return CMX_ERR_BAD_ARG;
    if (n == 0) return CMX_OK;
    if (!rho || !T || !q_tot || !q_lcl || !n_lcl || !q_rai || !n_rai || !dq_lcl || !dn_lcl || !dq_rai || !dn_rai)
        return CMX_ERR_BAD_ARG;
    const bool want_vel = vt_n || vt_m;
    const int velk = decode_vel(flags, want_vel);
    if (velk < 0 || (want_vel && !vel)) return CMX_ERR_BAD_ARG;
    const bool limited = flags & CMX_SB2006_LIMITED;
    const SbConsts<FT> c = make_sb_consts<FT>(*wr, *tps, vel, (double)Math<FT>::eps_1m());
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    constexpr int VEC = Math<FT>::VEC;
    const void *ptrs[] = {rho, T, q_tot, q_lcl, n_lcl, q_rai, n_rai, dq_lcl, dn_lcl, dq_rai, dn_rai, vt_n, vt_m};
    bool vec_ok = true;
    for (const void *p : ptrs) vec_ok = vec_ok && aligned16(p);
    const int64_t nvec = vec_ok ? n / VEC : 0;
    if (nvec > 0) {
        SbIn<FT> in{rho, T, q_tot, q_lcl, n_lcl, q_rai, n_rai};
        SbOut<FT> out{dq_lcl, dn_lcl, dq_rai, dn_rai, vt_n, vt_m};
        launch_tendencies<FT, VEC>(limited, velk, c, in, out, nvec, s);
    }
    const int64_t done = nvec * VEC;
    if (done < n) {   // unaligned columns, or the n % VEC tail: one point per lane
        SbIn<FT> in{rho + done, T + done, q_tot + done, q_lcl + done, n_lcl + done, q_rai + done, n_rai + done};
        SbOut<FT> out{dq_lcl + done, dn_lcl + done, dq_rai + done, dn_rai + done, vt_n ? vt_n + done : nullptr,
                      vt_m ? vt_m + done : nullptr};
        launch_tendencies<FT, 1>(limited, velk, c, in, out, n - done, s);
    }
    CMX_HIP_TRY(hipGetLastError());
    return CMX_OK;
}

template <typename FT, typename WR, typename TH, typename VL>
static int32_t process_entry(const WR *wr, const TH *tps, const VL *vel, uint32_t flags, int64_t n, const FT *q_tot,
                             const FT *q_lcl, const FT *q_rai, const FT *N_lcl, const FT *N_rai, const FT *rho,
                             const FT *T, FT *const out[CMX_SB2006_NPROC], void *stream) {
    if (!wr || !tps || !out || n < 0) return CMX_ERR_BAD_ARG;
    if (n == 0) return CMX_OK;
    if (!q_tot || !q_lcl || !q_rai || !N_lcl || !N_rai || !rho || !T) return CMX_ERR_BAD_ARG;
    const bool want_vel = out[CMX_SB_RAI_VEL_N] || out[CMX_SB_RAI_VEL_M];
    const int velk = decode_vel(flags, want_vel);
    if (velk < 0 || (want_vel && !vel)) return CMX_ERR_BAD_ARG;
    const bool limited = flags & CMX_SB2006_LIMITED;
    const SbConsts<FT> c = make_sb_consts<FT>(*wr, *tps, vel, (double)Math<FT>::eps_1m());
    SbProcOut<FT> o;
    for (int k = 0; k < CMX_SB2006_NPROC; ++k) o.col[k] = out[k];
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int grid = grid_for(n);
#define CMX_LAUNCH(L, V)                                                                                        \
    hipLaunchKernelGGL((sb2006_process_kernel<FT, L, V>), dim3(grid), dim3(kBlock), 0, s, c, q_tot, q_lcl, q_rai, \
                       N_lcl, N_rai, rho, T, o, n)
    if (limited) {
        if (velk == VEL_NONE) CMX_LAUNCH(true, VEL_NONE);
        else if (velk == VEL_SB) CMX_LAUNCH(true, VEL_SB);
        else CMX_LAUNCH(true, VEL_CHEN);
    } else {
        if (velk == VEL_NONE) CMX_LAUNCH(false, VEL_NONE);
        else if (velk == VEL_SB) CMX_LAUNCH(false, VEL_SB);
        else CMX_LAUNCH(false, VEL_CHEN);
    }
#undef CMX_LAUNCH
    CMX_HIP_TRY(hipGetLastError());
    return CMX_OK;
}

}  // namespace cmx

extern "C" {

int32_t cmx_sb2006_warm_rain_tendencies_f32(const cmx_warm_rain_2m_f32 *warm_rain, const cmx_thermo_f32 *tps,
                                            const cmx_rain_vel_f32 *vel, uint32_t flags, int64_t n, const float *rho,
                                            const float *T, const float *q_tot, const float *q_lcl, const float *n_lcl,
                                            const float *q_rai, const float *n_rai, float *dq_lcl_dt, float *dn_lcl_dt,
                                            float *dq_rai_dt, float *dn_rai_dt, float *vt_rai_n, float *vt_rai_m,
                                            void *stream) {
    return cmx::tendencies_entry<float>(warm_rain, tps, vel, flags, n, rho, T, q_tot, q_lcl, n_lcl, q_rai, n_rai,
                                        dq_lcl_dt, dn_lcl_dt, dq_rai_dt, dn_rai_dt, vt_rai_n, vt_rai_m, stream);
}

int32_t cmx_sb2006_warm_rain_tendencies_f64(const cmx_warm_rain_2m_f64 *warm_rain, const cmx_thermo_f64 *tps,
                                            const cmx_rain_vel_f64 *vel, uint32_t flags, int64_t n, const double *rho,
                                            const double *T, const double *q_tot, const double *q_lcl,
                                            const double *n_lcl, const double *q_rai, const double *n_rai,
                                            double *dq_lcl_dt, double *dn_lcl_dt, double *dq_rai_dt, double *dn_rai_dt,
                                            double *vt_rai_n, double *vt_rai_m, void *stream) {
    return cmx::tendencies_entry<double>(warm_rain, tps, vel, flags, n, rho, T, q_tot, q_lcl, n_lcl, q_rai, n_rai,
                                         dq_lcl_dt, dn_lcl_dt, dq_rai_dt, dn_rai_dt, vt_rai_n, vt_rai_m, stream);
}

int32_t cmx_sb2006_process_rates_f32(const cmx_warm_rain_2m_f32 *warm_rain, const cmx_thermo_f32 *tps,
                                     const cmx_rain_vel_f32 *vel, uint32_t flags, int64_t n, const float *q_tot,
                                     const float *q_lcl, const float *q_rai, const float *N_lcl, const float *N_rai,
                                     const float *rho, const float *T, float *const out[CMX_SB2006_NPROC],
                                     void *stream) {
    return cmx::process_entry<float>(warm_rain, tps, vel, flags, n, q_tot, q_lcl, q_rai, N_lcl, N_rai, rho, T, out,
                                     stream);
}

int32_t cmx_sb2006_process_rates_f64(const cmx_warm_rain_2m_f64 *warm_rain, const cmx_thermo_f64 *tps,
                                     const cmx_rain_vel_f64 *vel, uint32_t flags, int64_t n, const double *q_tot,
                                     const double *q_lcl, const double *q_rai, const double *N_lcl,
                                     const double *N_rai, const double *rho, const double *T,
                                     double *const out[CMX_SB2006_NPROC], void *stream) {
    return cmx::process_entry<double>(warm_rain, tps, vel, flags, n, q_tot, q_lcl, q_rai, N_lcl, N_rai, rho, T, out,
                                      stream);
}

}  // extern "C"
